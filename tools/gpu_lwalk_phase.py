"""GPU box helper: per-phase cycle counters of k_lwalk (RTFE_DEBUG=1) on the bench tape: per (wave, round, pass) the cycles of
the directory scan + copy (up to the barrier), of the walk as lane 0 sees it, and of the wait for the slowest lane."""
import os, sys, json
os.environ.setdefault("RTFE_DEBUG", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from readtape_amd import frontend
tape = bench.make_base_tape(1000, 5_000_000)
base = torch.from_numpy(tape.rows).cuda()
rows = base.repeat(4, 1).contiguous()
fe = frontend.FrontEnd(frontend.FrontEndConfig.from_header(tape.spec.header(), nparmsets=1))
fe.set_timing(True)
r = fe.scan(rows); ms = fe.kernel_ms(); r.fetch()
ws = r.bufs["ws"].cpu().numpy()
dbg = ws[64:128].view(np.uint64); d2 = ws[136:200].view(np.uint64); why = ws[200:264].view(np.uint64)
n = max(int(dbg[7]), 1)
print(json.dumps({"rows": int(rows.shape[0]), "k_walk_ms": ms["k_walk"], "passes": n, "cycles_per_pass": {"dir+copy": float(d2[0]) / n, "walk(lane0)": float(d2[1]) / n, "wait_for_slowest": float(d2[2]) / n},
                  "why": [int(x) for x in why]}))
