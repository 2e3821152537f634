"""GPU box: where a chunk of k_zeros spends its cycles (RTFE_DEBUG=1 counters of rtfe_zeros.hip)."""
import os, sys
os.environ["RTFE_DEBUG"] = "1"
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import bench
from readtape_amd import frontend
base = bench.make_base_tape(seed=1002, target_rows=5e6, kind="pe")
hdr = base.spec.header()
k = max(1, int(1e8 // base.rows.shape[0]))
rows = torch.from_numpy(base.rows).cuda().repeat(k, 1).contiguous()
fe = frontend.FrontEnd(frontend.FrontEndConfig.from_header(hdr, nparmsets=1, find_zeros=True))
fe.set_timing(True)
for _ in range(2):
    r = fe.scan(rows); ms, n = fe.kernel_ms(); r.fetch(events=False)
ws = r.bufs["ws"].cpu().numpy()
d2 = ws[136:200].view(np.uint64)
nt = max(int(d2[6]), 1)
print("rows", rows.shape[0], "ms", {k2: round(v / max(n, 1), 2) for k2, v in ms.items() if v > 0.01}, "chunks", nt, "repair rounds", int(d2[4]), "lanes run again", int(d2[3]),
      "cycles/chunk (pass, joins + repairs, events + walkers)", [int(d2[i] / nt) for i in range(3)])
