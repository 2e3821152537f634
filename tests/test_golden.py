"""The CPU oracle against the committed golden vectors (outputs of the unmodified reference, made by
tests/make_goldens.py): identical .tap bytes and identical event streams.  Needs no reference, so it
also runs on the GPU box."""
import os
import subprocess

import pytest

import refdump
from golden_util import case_names, load_case
from readtape_amd import tbin


@pytest.mark.parametrize("name", case_names())
def test_oracle_matches_golden(name, tmp_path, oracle_bin):
    g = load_case(name)
    wd = str(tmp_path)
    tbin.write_tbin(os.path.join(wd, "t.tbin"), g["hdr"], g["rows"])
    p = subprocess.run([oracle_bin, "-v", f"-out={wd}/o", f"-evt={wd}/o.evt"] + g["oracle_opts"] + [f"{wd}/t.tbin"],
                       capture_output=True, text=True)
    assert p.returncode in (0, 99), p.stderr
    a = refdump.load(f"{wd}/o.evt")
    b = g["events"]
    if g["returncode"] != 0:
        n = min(a.size, b.size); a, b = a[:n], b[:n]
    else:
        ot = open(f"{wd}/o.tap", "rb").read()
        assert ot == g["tap"]
        # the reference's per-block result lines (error / parity / ECC / corrected-bit counts, AGC range, speed, offsets)
        mine = [l.strip() for l in open(f"{wd}/o.log").read().splitlines() if l.startswith("wrote block") or "tapemark at" in l or "observed flux transitions" in l or "density was set to" in l]
        assert mine == list(g["blocklog"])
    diffs = refdump.compare(a, b)
    assert not diffs, "; ".join(diffs)
    assert len(case_names()) >= 10
