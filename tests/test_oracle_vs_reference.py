"""Pins the CPU oracle (oracle/) against the UNMODIFIED reference compiled by oracle/Makefile:
identical SIMH .tap bytes and identical front-end event streams (track, polarity, peak time bits,
voltage bits, detection sample, AGC gain bits, baseline bits, parmset) — on fresh synthetic tapes.
Runs only where /root/reference is mounted; the committed vectors in tests/golden/ carry the same
check to machines without it (tests/test_golden.py)."""
import os
import subprocess

import pytest

import refdump
from cases import CASES


def run_reference(ref_bin, workdir, base, ref_opts):
    opts = ["-v", "-tap", "-nolabels"] + list(ref_opts)
    if "-m" not in opts:
        opts.append("-nm")
    env = dict(os.environ, RT_EVENT_DUMP=os.path.join(workdir, base + ".ref.evt"))
    p = subprocess.run([ref_bin] + opts + [base], cwd=workdir, env=env, capture_output=True, text=True)
    return p


def run_oracle(oracle_bin, workdir, base, or_opts):
    p = subprocess.run([oracle_bin, "-v", f"-out={workdir}/{base}.or", f"-evt={workdir}/{base}.or.evt"] + list(or_opts)
                       + [f"{workdir}/{base}.tbin"], capture_output=True, text=True)
    return p


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_matches_reference(name, tmp_path, oracle_bin, ref_bin):
    build, ref_opts, or_opts = CASES[name]
    tape = build()
    wd = str(tmp_path)
    tape.write(os.path.join(wd, "t.tbin"))
    if getattr(build, "parms_text", ""):
        open(os.path.join(wd, "t.parms"), "w").write(build.parms_text)
        or_opts = list(or_opts) + [f"-parms={wd}/t.parms"]
    pr = run_reference(ref_bin, wd, "t", ref_opts)
    po = run_oracle(oracle_bin, wd, "t", or_opts)
    assert po.returncode in (0, 99), po.stderr
    ref_tap = os.path.join(wd, "t.tap")
    or_tap = os.path.join(wd, "t.or.tap")
    if pr.returncode == 0:
        rt = open(ref_tap, "rb").read() if os.path.exists(ref_tap) else b""
        ot = open(or_tap, "rb").read()
        assert rt == ot, f"{name}: .tap differs"
    a = refdump.load(os.path.join(wd, "t.or.evt"))
    b = refdump.load(os.path.join(wd, "t.ref.evt"))
    if pr.returncode != 0:
        # the reference hit one of its own fatal asserts (exit 99): compare up to where it stopped
        n = min(a.size, b.size)
        a, b = a[:n], b[:n]
    diffs = refdump.compare(a, b)
    assert not diffs, f"{name}: " + "; ".join(diffs)
    assert a.size > 0
