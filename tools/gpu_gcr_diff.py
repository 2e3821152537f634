"""GCR golden case: default path vs peak path, event lists side by side where they differ (debugging aid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]
import numpy as np
from golden_util import load_case
from parity_util import config_for
from readtape_amd import frontend
g = load_case("gcr")
cfg = config_for(g["hdr"], g["oracle_opts"])
for rep in range(6):
    res = []
    for pp in ("0", "1"):
        os.environ["RTFE_PEAK_PATH"] = pp
        fe = frontend.FrontEnd(cfg)
        r = fe.scan(g["rows"]).fetch()
        res.append((fe, r))
    (f0, r0), (f1, r1) = res
    st = f1.scan_stats(r1)
    nd = 0
    for b in range(r0.nbursts):
        for t in range(cfg.ntrks):
            a, c = r0.track_events(b, 0, t), r1.track_events(b, 0, t)
            if a.tobytes() != c.tobytes():
                nd += 1
                n = min(len(a), len(c))
                k = next((i for i in range(n) if a[i].tobytes() != c[i].tobytes()), n)
                if nd <= 2:
                    print("rep", rep, "burst", b, "trk", t, "len", len(a), len(c), "first diff at", k)
                    for i in range(max(0, k - 2), min(n, k + 3)):
                        print("   ", i, a[i], c[i])
    print("rep", rep, "differing lists", nd, "stats", {k: st[k] for k in ("redone", "parallel", "sequential", "gave_up")}, "reset", r0.bursts["reset_sample"].tolist(), r1.bursts["reset_sample"].tolist(), "end", r0.bursts["end_sample"].tolist(), r1.bursts["end_sample"].tolist())
