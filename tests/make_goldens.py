"""Generates tests/golden/*.npz from the UNMODIFIED reference (oracle/_ref/readtape_evt, built by
oracle/Makefile from /root/reference/src).  Run in the build container only:

    python tests/make_goldens.py [case ...]        (no names: all cases)

Each vector holds: the synthetic tape (int16 rows + header fields), the reference command line, the
reference's SIMH .tap bytes, its exit code and its front-end event dump (oracle/ref_event_shim.c).
Only data is stored — no reference source or text."""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import refdump  # noqa: E402
from cases import CASES  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "readtape_evt")
OUT = os.path.join(ROOT, "tests", "golden")


def main():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"], check=True)
    os.makedirs(OUT, exist_ok=True)
    tapes = {}
    only = set(sys.argv[1:])
    for name in sorted(CASES):
        if only and name not in only:
            continue
        build, ref_opts, or_opts = CASES[name]
        tape = build()
        tkey = build.__name__
        if tkey not in tapes and not (only and os.path.exists(os.path.join(OUT, f"tape_{tkey}.npz")) and tkey in ("case_nrzi7", "case_pe", "case_gcr")):
            tapes[tkey] = tape
            s = tape.spec
            np.savez_compressed(os.path.join(OUT, f"tape_{tkey}.npz"), rows=tape.rows,
                                hdr=np.array([s.ntrks, s.tdelta_ns, s.mode, s.tstart_ns, s.flags], dtype=np.int64), trkorder=np.array(s.trkorder),
                                hdrf=np.array([s.maxvolts, s.bpi, s.ips], dtype=np.float32))
        with tempfile.TemporaryDirectory() as wd:
            tape.write(os.path.join(wd, "t.tbin"))
            parms_text = getattr(build, "parms_text", "")
            if parms_text:
                open(os.path.join(wd, "t.parms"), "w").write(parms_text)       # the reference looks for <basename>.parms first (src/parmsets.c:337-377)
            opts = ["-v", "-tap", "-nolabels"] + list(ref_opts)
            if "-m" not in opts:
                opts.append("-nm")
            env = dict(os.environ, RT_EVENT_DUMP=os.path.join(wd, "t.evt"))
            p = subprocess.run([REF] + opts + ["t"], cwd=wd, env=env, capture_output=True, text=True)
            tap = open(os.path.join(wd, "t.tap"), "rb").read() if os.path.exists(os.path.join(wd, "t.tap")) else b""
            evt = refdump.load(os.path.join(wd, "t.evt"))
            blocks = [l.strip() for l in p.stdout.splitlines() if l.startswith("wrote block") or "tapemark at" in l or (l.startswith("  track ") and "observed flux transitions" in l) or "density was set to" in l]
        np.savez_compressed(os.path.join(OUT, f"case_{name}.npz"), tape=tkey, ref_opts=np.array(opts),
                            oracle_opts=np.array(list(or_opts), dtype="U64"), tap=np.frombuffer(tap, dtype=np.uint8),
                            events=evt, returncode=p.returncode, blocklog=np.array(blocks), parms_text=np.array(parms_text))
        print(f"{name}: {tape.rows.shape[0]} rows, {evt.size} records, tap {len(tap)} bytes, rc {p.returncode}")


if __name__ == "__main__":
    main()
