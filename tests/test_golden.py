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
    popt = []
    if g.get("parms_text"):
        open(os.path.join(wd, "t.parms"), "w").write(g["parms_text"])
        popt = [f"-parms={wd}/t.parms"]
    p = subprocess.run([oracle_bin, "-v", f"-out={wd}/o", f"-evt={wd}/o.evt"] + g["oracle_opts"] + popt + [f"{wd}/t.tbin"],
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
        mine = [l.strip() for l in open(f"{wd}/o.log").read().splitlines() if l.startswith("wrote block") or "tapemark at" in l or "observed flux transitions" in l or "density was set to" in l or "average peak height is" in l]
        assert mine == list(g["blocklog"])
    diffs = refdump.compare(a, b)
    assert not diffs, "; ".join(diffs)
    assert len(case_names()) >= 10


def test_parms_text_value_ranges_are_enforced():
    """A value outside its range is fatal in the reference's .parms reader - also for a parameter the mode ignores
    (src/parmsets.c:61-73, 286-297): the host library refuses the file the same way."""
    import ctypes as C
    from readtape_amd import pipeline
    lib = pipeline._load_decode_lib()
    arr = (pipeline._Parms * 15)()
    hdr = "parms active, clk_window, clk_alpha, agc_window, agc_alpha, min_peak, clk_factor, pulse_adj, pkww_bitfrac, pkww_rise, midbit, z1pt, z2pt, id\n"
    good = hdr + "{1, 0, 0.2, 0, 0.3, 1.0, 0, 0.3, 0.7, 0.2, 0.5, 1.45, 2.35, PRM}\n"
    assert lib.rt_parse_parms_text(tbin.MODE_NRZI, good.encode(), arr) == 1
    for bad in ("{1, 0, 0.2, 0, 0.3, 1.0, 0, 0.3, 0.7, 0.2, 0.5, 0, 2.35, PRM}",       # z1pt below 1 (ignored for NRZI, still checked)
                "{1, 0, 0.2, 0, 0.3, 1.0, 0, 0.3, 2.4, 0.2, 0.5, 1.45, 2.35, PRM}",    # pkww_bitfrac above 2
                "{1, 0, 0.2, 11, 0.0, 1.0, 0, 0.3, 0.7, 0.2, 0.5, 1.45, 2.35, PRM}",   # agc_window above 10
                "{1, 0, 1.5, 0, 0.3, 1.0, 0, 0.3, 0.7, 0.2, 0.5, 1.45, 2.35, PRM}"):   # clk_alpha above 1
        assert lib.rt_parse_parms_text(tbin.MODE_NRZI, (hdr + bad + "\n").encode(), arr) < 0, bad
