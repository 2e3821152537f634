"""GPU box helper: per-phase cycle counters of k_decode (RTFE_DEBUG=1) on the bench tape."""
import os, sys, json
os.environ.setdefault("RTFE_DEBUG", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from readtape_amd import frontend
rows_target = float(sys.argv[1]) if len(sys.argv) > 1 else 1e8
tape = bench.make_base_tape(1000, 5_000_000)
base = torch.from_numpy(tape.rows).cuda()
copies = max(1, int(round(rows_target / base.shape[0])))
rows = base.repeat(copies, 1).contiguous()
fe = frontend.FrontEnd(frontend.FrontEndConfig.from_header(tape.spec.header(), nparmsets=1))
fe.set_timing(True)
for _ in range(2):
    r = fe.scan(rows)
ms = fe.kernel_ms()[0]
r.fetch()
ws = r.bufs["ws"].cpu().numpy()
dbg = ws[64:128].view(np.uint64)
d2 = ws[136:200].view(np.uint64)
why = ws[200:264].view(np.uint64)
scr = ws[264:328].view(np.uint64)
no = max(int(dbg[7]), 1)
lens = (r.bursts["end_sample"] - r.bursts["reset_sample"])
print(json.dumps({"rows": int(rows.shape[0]), "kernel_ms": ms, "bursts": int(r.nbursts), "burst_len_max": int(lens.max()), "burst_len_mean": float(lens.mean()),
                  "tiles": int(dbg[3]), "cyc_per_tile": {"load": float(dbg[0]) / max(int(dbg[3]), 1), "screen": float(dbg[1]) / max(int(dbg[3]), 1), "walk": float(dbg[2]) / max(int(dbg[3]), 1), "walk_build": float(dbg[4]) / max(int(dbg[3]), 1), "walk_chain": float(dbg[5]) / max(int(dbg[3]), 1), "walk_final": float(dbg[6]) / max(int(dbg[3]), 1)},
                  "events": int(r.counts.sum()), "optimistic_tiles": int(dbg[7]), "opt_cyc": {"dir": float(d2[0]) / no, "unpack": float(d2[1]) / no, "walk": float(d2[2]) / no, "final": float(d2[3]) / no}, "why(0?,1 notfast,2 k>=4,3 top!fast,4 bot!fast,5 guard,6 misc,7 chain)": [int(x) for x in why], "screen_cyc_per_tile(load,screen,build,write)": [float(scr[i]) / max(int(scr[4]), 1) for i in range(4)], "screen_chain_starts": [float(scr[5]) / max(int(scr[4]), 1), float(scr[6]) / max(int(scr[4]), 1)], "screen_tiles": int(scr[4])}))
