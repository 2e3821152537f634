"""GPU box helper: one -zeros tape of the stress sweep, with and without the sub-segment path."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, subprocess, refdump
from readtape_amd import synth, pipeline, tbin
from parity_util import ORACLE, build_oracle
build_oracle()
tape = synth.pe_tape(seed=833875458, nblocks=int(sys.argv[1]) if len(sys.argv) > 1 else 4, minlen=30, maxlen=int(sys.argv[2]) if len(sys.argv) > 2 else 900, gap_samples=3000, amplitude=2.5, noise_mv=50.0, jitter=0.08)
hdr = tape.spec.header()
for par in ("1", "0"):
    os.environ["RTFE_ZC_PARALLEL"] = par
    with tempfile.TemporaryDirectory() as wd:
        tbin.write_tbin(os.path.join(wd, "t.tbin"), hdr, tape.rows)
        p = subprocess.run([ORACLE, "-v", f"-out={wd}/o", f"-evt={wd}/o.evt", "-zeros", os.path.join(wd, "t.tbin")], capture_output=True, text=True)
        st, res = pipeline.decode_tape(hdr, tape.rows, os.path.join(wd, "g.tap"), evt_path=os.path.join(wd, "g.evt"), find_zeros=True)
        a, b = refdump.load(os.path.join(wd, "g.evt")), refdump.load(os.path.join(wd, "o.evt"))
        d = refdump.compare(a, b)
        print("parallel", par, "rows", tape.rows.shape[0], "bursts", res.nbursts, "tap same", open(os.path.join(wd, "g.tap"), "rb").read() == open(os.path.join(wd, "o.tap"), "rb").read(), "diffs", d[:2], "exact", st["exact_scans"])
