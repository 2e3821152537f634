"""GPU box helper: distribution of candidate units / runs per tile (from the tile directory k_screen leaves in the workspace)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from readtape_amd import frontend
tape = bench.make_base_tape(1000, 5_000_000)
rows = torch.from_numpy(tape.rows).cuda()
fe = frontend.FrontEnd(frontend.FrontEndConfig.from_header(tape.spec.header(), nparmsets=1))
r = fe.scan(rows)
r.fetch()
ws = r.bufs["ws"].cpu().numpy()
nrows = rows.shape[0]; ntrks = 9; T = 512
nchunks = nrows * ntrks // 512 + 1
nwords = (nchunks + 63) // 64
dir_off = (512 + nwords * 8 + 255) & ~255
ntiles = (nrows + T - 1) // T
d = ws[dir_off:dir_off + ntiles * ntrks * 8].view(np.uint16).reshape(ntiles, ntrks, 4)
units = d[:, :, 0].astype(np.int64).sum(axis=1)
runs = d[:, :, 1].astype(np.int64).sum(axis=1)
print(json.dumps({"tiles": int(ntiles), "units_mean": float(units.mean()), "units_p50": float(np.percentile(units, 50)), "units_p99": float(np.percentile(units, 99)),
                  "units_max": int(units.max()), "runs_mean": float(runs.mean()), "runs_p99": float(np.percentile(runs, 99)), "runs_max": int(runs.max()),
                  "list_units_max": int(d[:, :, 0].max()), "events": int(r.counts.sum())}))
