"""One golden case through the peak path on the GPU: burst table, counts, chain statistics (debugging aid)."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]
os.environ.setdefault("RTFE_PEAK_PATH", "1")
import numpy as np
from golden_util import load_case
from parity_util import check_tape, config_for, oracle_attempts
from readtape_amd import frontend
for name in sys.argv[1:]:
    g = load_case(name)
    cfg = config_for(g["hdr"], g["oracle_opts"])
    fe = frontend.FrontEnd(cfg)
    r = fe.scan(g["rows"]).fetch()
    print(name, "nbursts", r.nbursts, "counts", r.counts.reshape(r.nbursts, -1)[:, :9].tolist(), "flags", r.bursts["flags"].tolist())
    print(fe.scan_stats(r))
    with tempfile.TemporaryDirectory() as wd:
        att = oracle_attempts(g["hdr"], g["rows"], g["oracle_opts"], wd)
        msgs, stats = check_tape(fe, g["hdr"], g["rows"], att)
    print(stats, msgs[:4])
