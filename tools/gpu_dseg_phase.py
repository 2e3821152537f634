"""GPU box helper: per-phase cycle counters of k_dseg (RTFE_DEBUG=7) on the bench's GCR / PE tape.  usage: gpu_dseg_phase.py rows nparm kind"""
import os, sys, json
os.environ["RTFE_DEBUG"] = "7"
os.environ.setdefault("RTFE_DENSE_STOP", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from readtape_amd import frontend
rows_target = float(sys.argv[1]) if len(sys.argv) > 1 else 1e8
nparm = int(sys.argv[2]) if len(sys.argv) > 2 else 1
kind = sys.argv[3] if len(sys.argv) > 3 else "gcr"
tape = bench.make_base_tape(seed=1000, target_rows=5_000_000, kind=kind)
hdr = tape.spec.header()
extra = [(1.4, 0.20, 0.2, 0.5, 0, 0.0), (1.6, 0.14, 0.0, 0.5, 0, 0.0), (1.5, 0.10, 0.1, 0.5, 0, 0.0), (1.3, 0.25, 0.2, 0.5, 0, 0.0)]
parmsets = (list(frontend.DEFAULT_PARMSETS[hdr.mode]) + extra)[:nparm] if nparm > len(frontend.DEFAULT_PARMSETS[hdr.mode]) else None
cfg = frontend.FrontEndConfig.from_header(hdr, nparmsets=nparm, parmsets=parmsets)
base = torch.from_numpy(tape.rows).cuda()
rows = base.repeat(max(1, int(round(rows_target / base.shape[0]))), 1).contiguous()
fe = frontend.FrontEnd(cfg)
fe.set_timing(True)
for _ in range(2):
    r = fe.scan(rows)
torch.cuda.synchronize()
ms = fe.kernel_ms()[0]
ws = r.bufs["ws"][:512].cpu().numpy()
scr = ws[264:328].view(np.uint64).astype(np.float64)
tot = scr[0] + scr[1] + scr[2] + scr[7]
ntiles = rows.shape[0] / 1024
print(json.dumps({"env": {k: v for k, v in os.environ.items() if k.startswith("RTFE_D")}, "rows": int(rows.shape[0]), "nparm": nparm, "k_dseg_ms": round(ms.get("k_dseg", 0) / 2, 2),
                  "share": {"load+bands": round(scr[0] / tot, 3), "screen": round(scr[1] / tot, 3), "classify": round(scr[7] / tot, 3), "walk": round(scr[2] / tot, 3)},
                  "kcycles_per_tile_pass": round(tot / ntiles / 1e3, 1), "walk_steps_per_lane": round(scr[3] / max(scr[5], 1), 2), "fires_per_lane": round(scr[4] / max(scr[5], 1), 2),
                  "lane_walk_cycles": round(scr[6] / max(scr[5], 1))}))
