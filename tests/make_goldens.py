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


# Output files and the end-of-run report (src/readtape.c:1084-1111, 2021-2044): name -> (tape of this case, reference options).
# Without -tap the reference writes numbered .bin files, one per tape file.
def _three_files():
    from readtape_amd import synth
    return synth.nrzi_tape(seed=21, nblocks=5, minlen=40, maxlen=90, marks_every=2, gap_samples=1500)


FILE_CASES = {
    "files_nrzi9_bin": (_three_files, ["-v", "-nolabels", "-nm"]),           # blocks, mark, blocks, mark, block: three .bin files
    "files_nrzi9_tap": (_three_files, ["-nolabels", "-nm", "-tap"]),        # (no -v: only the first block is logged)
    "files_nrzi7_bin": (CASES["nrzi7"][0], ["-v", "-nolabels", "-nm"]),
    "files_pe_m_tap": (CASES["pe_m"][0], ["-v", "-nolabels", "-tap", "-m"]),
    "files_gcr_bin": (CASES["gcr"][0], ["-nolabels", "-nm"]),
}


def report_lines(text):
    """The lines of the reference's log that are about its output files and its summary."""
    keep = []
    for l in text.splitlines():
        if (l.startswith('creating file "') or " was closed at time " in l or l.startswith("summary for file") or " samples were processed in " in l
                or (l.startswith("  created ") and "output file" in l) or l.startswith("  decoded ") or l.startswith("  the last block written")
                or " had errors, " in l or "blocks were unusable" in l or "good blocks had to try" in l or (l.startswith("  parmset ") and "was tried" in l)):
            keep.append(l.rstrip())
    return keep


def make_file_cases(only):
    for name, (build, opts) in sorted(FILE_CASES.items()):
        if only and name not in only:
            continue
        tape = build()
        sp = tape.spec
        with tempfile.TemporaryDirectory() as wd:
            tape.write(os.path.join(wd, "t.tbin"))
            p = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "readtape_ref")] + opts + ["t"], cwd=wd, capture_output=True, text=True)
            outs = sorted(f for f in os.listdir(wd) if f.endswith(".bin") or f.endswith(".tap"))
            blobs = {f"file{i}": np.frombuffer(open(os.path.join(wd, f), "rb").read(), dtype=np.uint8) for i, f in enumerate(outs)}
        np.savez_compressed(os.path.join(OUT, f"{name}.npz"), rows=tape.rows, hdr=np.array([sp.ntrks, sp.tdelta_ns, sp.mode, sp.tstart_ns, sp.flags], dtype=np.int64),
                            hdrf=np.array([sp.maxvolts, sp.bpi, sp.ips], dtype=np.float32), trkorder=np.array(sp.trkorder), ref_opts=np.array(opts), names=np.array(outs), report=np.array(report_lines(p.stdout)),
                            returncode=p.returncode, **blobs)
        print(f"{name}: {outs}, {len(report_lines(p.stdout))} report lines, rc {p.returncode}")


def main():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"], check=True)
    os.makedirs(OUT, exist_ok=True)
    make_file_cases(set(a for a in sys.argv[1:] if a.startswith("files_")) if sys.argv[1:] else set())
    if sys.argv[1:] and all(a.startswith("files_") for a in sys.argv[1:]):
        return
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
            blocks = [l.strip() for l in p.stdout.splitlines() if l.startswith("wrote block") or "tapemark at" in l or (l.startswith("  track ") and "observed flux transitions" in l) or "density was set to" in l or "average peak height is" in l]
        np.savez_compressed(os.path.join(OUT, f"case_{name}.npz"), tape=tkey, ref_opts=np.array(opts),
                            oracle_opts=np.array(list(or_opts), dtype="U64"), tap=np.frombuffer(tap, dtype=np.uint8),
                            events=evt, returncode=p.returncode, blocklog=np.array(blocks), parms_text=np.array(parms_text))
        print(f"{name}: {tape.rows.shape[0]} rows, {evt.size} records, tap {len(tap)} bytes, rc {p.returncode}")


if __name__ == "__main__":
    main()
