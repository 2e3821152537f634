"""GPU box helper: where a window's fetch (ScanResult.fetch) spends its host time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from readtape_amd import frontend, pipeline
tape = bench.make_base_tape(1000, 5_000_000)
hdr = tape.spec.header()
rows = torch.from_numpy(np.tile(tape.rows, (2, 1))[: (1 << 23) + (1 << 18)]).cuda()
full = pipeline.default_parmsets(hdr.mode, 1)
cfg = frontend.FrontEndConfig.from_header(hdr, parmsets=pipeline.frontend_parmsets(full))
fe = frontend.FrontEnd(cfg)
import cProfile, pstats
for rep in range(3):
    res = fe.scan(rows, own_rows=1 << 23)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if rep == 2:
        pr = cProfile.Profile(); pr.enable()
    res.fetch()
    if rep == 2:
        pr.disable()
    print("fetch %.2f ms, bursts %d, events %d (%.1f MB)" % ((time.perf_counter() - t0) * 1e3, res.nbursts, res._events.shape[0], res._events.nbytes / 1e6))
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
