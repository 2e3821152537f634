"""Generates tests/golden/csv_*.npz: a CSV sample file (text), the options, and the .tbin the UNMODIFIED reference converter
(oracle/_ref/csvtbin_ref, compiled by oracle/Makefile from /root/reference/src/csvtbin.c) makes of it.  Build container only."""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from readtape_amd import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "csvtbin_ref")
READTAPE = os.path.join(ROOT, "oracle", "_ref", "readtape_ref")
OUT = os.path.join(ROOT, "tests", "golden")


def csv_text(tape, digits=6, tdigits=7):
    """What a logic analyser exports: two title lines, then time and the volts of every head, a few decimals each."""
    s = tape.spec
    v = tape.rows.astype(np.float64) / 32767.0 * s.maxvolts
    t = (s.tstart_ns + np.arange(v.shape[0]) * s.tdelta_ns) / 1e9
    lines = ["Time [s], " + ", ".join(f"Channel {k}" for k in range(s.ntrks)), "Time [s], " + ", ".join(f"Channel {k}" for k in range(s.ntrks))]
    for i in range(v.shape[0]):
        lines.append(f"{t[i]:.{tdigits}f}, " + ", ".join(f"{x:.{digits}f}" for x in v[i]))
    return "\n".join(lines) + "\n"


def _rewired(t, order):
    """The same recording with its heads wired in another order: column i carries the track order[i] names (what -order= undoes)."""
    import dataclasses
    n = t.rows.shape[1]
    h2t = [n - 1 if ch in "pP" else int(ch) for ch in order]
    return dataclasses.replace(t, rows=np.ascontiguousarray(t.rows[:, h2t]))


# name -> (tape, the converter's options, the decoder's options for the .tap the reference makes of the converter's .tbin)
CASES = {
    "csv_nrzi9": (lambda: synth.nrzi_tape(seed=81, nblocks=2, minlen=30, maxlen=50, gap_samples=1200), ["-nrzi", "-bpi=800", "-ips=50"], []),
    "csv_nrzi7_order_sub2": (lambda: synth.nrzi_tape(seed=82, nblocks=2, minlen=30, maxlen=50, gap_samples=1200, ntrks=7),
                             ["-ntrks=7", "-order=543210p", "-subsample=2", "-invert", "-maxvolts=6.0", "-nrzi", "-bpi=800", "-ips=50"], ["-ntrks=7"]),
    "csv_pe_scale": (lambda: synth.pe_tape(seed=83, nblocks=1, minlen=40, maxlen=60, gap_samples=1200), ["-pe", "-scale=0.5", "-bpi=1600", "-ips=50"], []),
    # no -order at the converter (the .tbin says TBIN_NO_REORDER), the head order given to the decoder instead
    "csv_nrzi7_order_late": (lambda: _rewired(synth.nrzi_tape(seed=84, nblocks=2, minlen=30, maxlen=50, gap_samples=1200, ntrks=7), "543210p"),
                             ["-ntrks=7", "-nrzi", "-bpi=800", "-ips=50"], ["-ntrks=7", "-order=543210p"]),
}


def main():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"], check=True)
    for name, (build, opts, dopts) in CASES.items():
        tape = build()
        text = csv_text(tape)
        with tempfile.TemporaryDirectory() as wd:
            open(os.path.join(wd, "c.csv"), "w").write(text)
            p = subprocess.run([REF] + opts + ["c"], cwd=wd, capture_output=True, text=True)
            assert p.returncode == 0, p.stdout + p.stderr
            out = open(os.path.join(wd, "c.tbin"), "rb").read()
            # ... and what the unmodified decoder makes of that .tbin (csv -> .tap end to end)
            os.rename(os.path.join(wd, "c.tbin"), os.path.join(wd, "d.tbin"))      # (given "c" the decoder would open c.csv)
            q = subprocess.run([READTAPE, "-tap", "-nolabels", "-nm"] + dopts + ["d"], cwd=wd, capture_output=True, text=True)
            assert q.returncode == 0 and os.path.exists(os.path.join(wd, "d.tap")), name + "\n" + q.stdout + q.stderr
            tap = open(os.path.join(wd, "d.tap"), "rb").read()
        np.savez_compressed(os.path.join(OUT, f"{name}.npz"), csv=np.frombuffer(text.encode(), dtype=np.uint8), opts=np.array(opts), tbin=np.frombuffer(out, dtype=np.uint8),
                            decode_opts=np.array(dopts), tap=np.frombuffer(tap, dtype=np.uint8))
        print(name, len(text), "chars ->", len(out), "bytes of .tbin ->", len(tap), "bytes of .tap")


if __name__ == "__main__":
    main()
