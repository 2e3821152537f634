"""GPU box helper: randomized parity sweep (device events vs the CPU oracle) over tape parameters the committed tests do
not pin: amplitudes, noise, jitter, track counts, skews, parameter-set sweeps.  Prints one line per tape; exits non-zero
on the first mismatch."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from parity_util import check_tape, config_for, oracle_attempts
from readtape_amd import frontend, synth

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
ntapes = int(sys.argv[2]) if len(sys.argv) > 2 else 16
bad = 0
for i in range(ntapes):
    kind = ["nrzi", "nrzi", "nrzi", "pe", "gcr"][int(rng.integers(0, 5))]
    amp = float(rng.choice([0.6, 1.0, 1.8, 2.5, 3.2]))
    noise = float(rng.choice([5.0, 10.0, 25.0, 50.0, 80.0]))
    jit = float(rng.choice([0.0, 0.02, 0.05, 0.08]))
    opts = []
    kw = dict(amplitude=amp, noise_mv=noise, jitter=jit)
    seed = int(rng.integers(1, 1 << 30))
    if kind == "nrzi":
        ntrks = int(rng.choice([9, 9, 7]))
        tape = synth.nrzi_tape(seed=seed, nblocks=int(rng.integers(2, 9)), minlen=16, maxlen=int(rng.choice([200, 1200, 3000])),
                               marks_every=int(rng.choice([0, 3])), ntrks=ntrks, gap_samples=int(rng.choice([1500, 4000])), **kw)
        if ntrks == 7: opts.append("-ntrks=7")
        if rng.random() < 0.3: opts.append("-m")
        if rng.random() < 0.2: opts.append("-invert")
        if rng.random() < 0.25 and ntrks == 9: opts.append("-skew=" + ",".join(str(int(x)) for x in rng.integers(0, 6, size=9)))
    elif kind == "pe":
        tape = synth.pe_tape(seed=seed, nblocks=int(rng.integers(2, 5)), minlen=30, maxlen=int(rng.choice([200, 900])), gap_samples=3000, **kw)
        if rng.random() < 0.3: opts.append("-m")
    else:
        kw["amplitude"] = max(amp, 1.0)
        tape = synth.gcr_tape(seed=seed, nblocks=int(rng.integers(2, 4)), minlen=40, maxlen=int(rng.choice([200, 900])), gap_samples=4000, **kw)
        if rng.random() < 0.3: opts.append("-m")
    hdr = tape.spec.header()
    with tempfile.TemporaryDirectory() as wd:
        att = oracle_attempts(hdr, tape.rows, opts, wd)
        for rec in ("default", "1"):
            if rec == "1": os.environ["RTFE_RECORD_PATH"] = "1"
            else: os.environ.pop("RTFE_RECORD_PATH", None)
            fe = frontend.FrontEnd(config_for(hdr, opts))
            msgs, stats = check_tape(fe, hdr, tape.rows, att)
            tag = f"{i:3d} {kind} seed {seed} amp {amp} noise {noise} jit {jit} opts {opts} record_path {rec}: attempts {len(att)} events {stats['events']} speculative {stats.get('speculative')} flags {stats.get('flags')}"
            print(("FAIL " if msgs else "ok   ") + tag, flush=True)
            if msgs:
                print("\n".join(msgs[:6]))
                bad += 1
sys.exit(1 if bad else 0)
