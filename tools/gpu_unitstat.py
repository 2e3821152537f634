"""GPU box helper: distribution of candidate units / runs per tile (from the tile directory k_screen leaves in the workspace)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from readtape_amd import frontend, synth

def stat(name, tape, **kw):
    rows = torch.from_numpy(tape.rows).cuda()
    fe = frontend.FrontEnd(frontend.FrontEndConfig.from_header(tape.spec.header(), **kw))
    r = fe.scan(rows)
    r.fetch()
    ws = r.bufs["ws"].cpu().numpy()
    nrows = rows.shape[0]; ntrks = tape.rows.shape[1]; T = 512
    nst = ntrks * len(set(fe.widths))
    nchunks = nrows * ntrks // 512 + 1
    nwords = (nchunks + 63) // 64
    dir_off = (512 + nwords * 8 + 255) & ~255
    ntiles = (nrows + T - 1) // T
    d = ws[dir_off:dir_off + ntiles * nst * 8].view(np.uint16).reshape(ntiles, nst, 4)
    bad = (d[:, :, 0] == 0xFFFF)
    units = np.where(bad, 0, d[:, :, 0]).astype(np.int64).sum(axis=1)
    runs = d[:, :, 1].astype(np.int64).sum(axis=1)
    print(json.dumps({"config": name, "tiles": int(ntiles), "lists_incomplete": int(bad.sum()), "tiles_with_incomplete": int(bad.any(axis=1).sum()),
                      "units_mean": float(units.mean()), "units_p99": float(np.percentile(units, 99)), "units_max": int(units.max()),
                      "runs_mean": float(runs.mean()), "runs_p99": float(np.percentile(runs, 99)), "runs_max": int(runs.max()),
                      "list_units_max": int(np.where(bad, 0, d[:, :, 0]).max()), "list_runs_max": int(d[:, :, 1].max()), "events": int(r.counts.sum()), "flagged": int((r.bursts["flags"] & ~np.uint32(frontend.F_EXACT_START)).astype(bool).sum()), "widths": fe.widths}))

if __name__ == "__main__":
    floors = [float(x) for x in sys.argv[1:]] or [0.0]
    for fl in floors:
        stat(f"C2 NRZI floor {fl}", bench.make_base_tape(1000, 5_000_000), nparmsets=1, screen_floor_height=fl)
        stat(f"PE peak floor {fl}", synth.pe_tape(seed=71, nblocks=40, minlen=500, maxlen=4000, gap_samples=6000), nparmsets=1, screen_floor_height=fl)
        stat(f"GCR floor {fl}", synth.gcr_tape(seed=81, nblocks=20, minlen=1000, maxlen=4000, gap_samples=8000), nparmsets=1, screen_floor_height=fl)
