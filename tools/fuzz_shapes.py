"""Shape fuzzer for the peak path's record logic (k_sift / k_prep / k_gain), on the CPU emulator against the oracle.

Gaussian noise (tools/gpu_stress.sh, the N1 tape) almost never draws the shapes that decide whether a record may fire on the lean step:
two extremes of nearly the same height inside one window, a narrow valley right behind a flat top, a notch in a shoulder.  This tool
writes such shapes over peaks of a clean NRZI tape - every sample of the window either side of a chosen peak drawn from a mixture of
"a hair below the peak", "a little below", "well below" - and checks every event against the oracle.

  python tools/fuzz_shapes.py [--gpu] [--e2e] [seed0 [ntapes [kind]]]     (test infrastructure: the oracle through tests/parity_util; without --gpu the kernels run on tests/cpu_emul)
"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

from fuzz_util import KINDS, draw, shape_tape  # noqa: E402,F401


def main():
    from parity_util import check_tape, config_for, oracle_attempts
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    gpu = "--gpu" in sys.argv
    seed0 = int(args[0]) if len(args) > 0 else 1
    ntapes = int(args[1]) if len(args) > 1 else 4
    only = args[2] if len(args) > 2 else None          # one of KINDS: every tape of that format
    if gpu:
        from readtape_amd import frontend
        make = frontend.FrontEnd
    else:
        from emul_util import emul_frontend
        make = emul_frontend
    bad = 0
    for seed in range(seed0, seed0 + ntapes):
        d = draw(seed)
        if only:
            d["kind"] = only
        tape, rows, nsites, opts = shape_tape(seed, **d)
        hdr = tape.spec.header()
        with tempfile.TemporaryDirectory() as td:
            att = oracle_attempts(hdr, rows, opts, td)
        fe = make(config_for(hdr, opts))
        if "--e2e" in sys.argv:                                  # the whole pipeline on the shaped tape - front end, host decoders, .tap writer - against the oracle's .tap and its transitions
            import subprocess
            import refdump
            from parity_util import ORACLE
            from readtape_amd import pipeline, tbin
            with tempfile.TemporaryDirectory() as wd:
                tbin.write_tbin(os.path.join(wd, "t.tbin"), hdr, rows)
                p = subprocess.run([ORACLE, "-v", f"-out={wd}/o", f"-evt={wd}/o.evt"] + opts + [os.path.join(wd, "t.tbin")], capture_output=True, text=True)
                st, _ = pipeline.decode_tape(hdr, rows, os.path.join(wd, "g.tap"), evt_path=os.path.join(wd, "g.evt"), opts=pipeline.DecodeOptions(multiple_tries="-m" in opts), fe_factory=None if gpu else make)
                a, b = refdump.load(os.path.join(wd, "g.evt")), refdump.load(os.path.join(wd, "o.evt"))
                m2 = refdump.compare(a, b)
                if p.returncode == 0 and open(os.path.join(wd, "g.tap"), "rb").read() != open(os.path.join(wd, "o.tap"), "rb").read():
                    m2.append(".tap differs")
                print(f"{'ok' if not m2 else 'FAIL'} seed {seed} {d} e2e: {a.size} transitions, oracle rc {p.returncode}", flush=True)
                if m2:
                    bad += 1
                    print("\n".join(str(x) for x in m2[:6]), flush=True)
        for rep in range(2):                                    # (the second scan runs under the floor the first one learned)
            msgs, stats = check_tape(fe, hdr, rows, att)
            st = fe.scan_stats(fe.scan(rows).fetch())
            print(f"{'ok' if not msgs else 'FAIL'} seed {seed} {d} rep {rep} sites {nsites} events {stats['events']} exact {stats['exact']} parallel {st['parallel']} sequential {st['sequential']} redone {st['redone']} gave_up {st['gave_up']}", flush=True)
            if msgs:
                bad += 1
                print("\n".join(msgs[:6]), flush=True)
    print("FAILURES", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
