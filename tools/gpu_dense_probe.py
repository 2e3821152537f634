"""GPU box: the dense sample path stage by stage on one C4-shaped tape (tools only)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from readtape_amd import frontend
rows_target = float(sys.argv[1]) if len(sys.argv) > 1 else 5e6
nparm = int(sys.argv[2]) if len(sys.argv) > 2 else 8
tape = bench.make_base_tape(seed=1000, target_rows=int(rows_target), kind=sys.argv[3] if len(sys.argv) > 3 else "gcr")
hdr = tape.spec.header()
extra = [(1.4, 0.20, 0.2, 0.5, 0, 0.0), (1.6, 0.14, 0.0, 0.5, 0, 0.0), (1.5, 0.10, 0.1, 0.5, 0, 0.0), (1.3, 0.25, 0.2, 0.5, 0, 0.0)]
parmsets = (list(frontend.DEFAULT_PARMSETS[hdr.mode]) + extra)[:nparm] if nparm > len(frontend.DEFAULT_PARMSETS[hdr.mode]) else None
cfg = frontend.FrontEndConfig.from_header(hdr, nparmsets=nparm, parmsets=parmsets)
fe = frontend.FrontEnd(cfg)
rows = torch.from_numpy(tape.rows).cuda()
copies = int(os.environ.get("PROBE_COPIES", "1"))
if copies > 1: rows = rows.repeat(copies, 1).contiguous()
print("rows", tuple(rows.shape), "stop", os.environ.get("RTFE_DENSE_STOP"), flush=True)
fe.set_timing(True)
for i in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    r = fe.scan(rows)
    torch.cuda.synchronize(); dt = time.time() - t0
    print("scan", i, round(dt * 1e3, 2), "ms", flush=True)
ms, n = fe.kernel_ms()
print({k: round(v / max(n, 1), 3) for k, v in ms.items() if v > 0.02 * max(n, 1)}, flush=True)
r.fetch(events=False)
print("bursts", r.nbursts, "events", int(r.counts.sum()), fe.scan_stats(r), flush=True)
