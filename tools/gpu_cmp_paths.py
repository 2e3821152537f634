"""GPU check: the peak-record path (k_peaks -> k_chain) against the sample path (RTFE_PEAK_PATH=0) on golden tapes, event for event."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]
import numpy as np
import torch
from golden_util import load_case
from parity_util import config_for
from readtape_amd import frontend

for name in sys.argv[1:]:
    g = load_case(name)
    cfg = config_for(g["hdr"], g["oracle_opts"])
    res = []
    for pp in ("0", "1"):
        os.environ["RTFE_PEAK_PATH"] = pp
        fe = frontend.FrontEnd(cfg)
        t0 = time.time()
        r = fe.scan(g["rows"]).fetch()
        res.append((fe, r, time.time() - t0))
    (f0, r0, t0), (f1, r1, t1) = res
    bad = []
    if r0.nbursts != r1.nbursts:
        bad.append("nbursts")
    else:
        for k in ("zone_first", "zone_end", "reset_sample", "safe_last", "end_sample", "flags"):
            if not np.array_equal(r0.bursts[k], r1.bursts[k]):
                bad.append(k)
        nd = 0
        for b in range(r0.nbursts):
            for p in range(len(cfg.parmsets)):
                for t in range(cfg.ntrks):
                    if r0.track_events(b, p, t).tobytes() != r1.track_events(b, p, t).tobytes():
                        nd += 1
        if nd:
            bad.append(f"{nd} event lists")
    print(name, "OK" if not bad else "DIFF " + str(bad), f1.scan_stats(r1), "events", int(r1.counts.sum()), "%.3f / %.3f s" % (t0, t1), flush=True)
