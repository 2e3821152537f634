"""Randomized Whirlwind sweep (build container: needs oracle/_ref): random tapes (block lengths, gaps down to a few bit times, block
marks, noise, jitter, weak alternate tracks, either polarity) and options (-fluxdir, -reverse, -deskew on skewed heads) through the compiled reference, the
oracle, and - every `emul_every`-th tape - the device path (pipeline.decode_tape_ww; emulated kernels, or the GPU with FUZZ_GPU=1).  .tap bytes and block lines must agree.
    python tests/fuzz_ww.py [seed] [ntapes] [emul_every]"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from readtape_amd import pipeline, synth

REF = os.path.join(ROOT, "oracle", "_ref", "readtape_ref")
ORACLE = os.path.join(ROOT, "oracle", "_build", "oracle_readtape")


def lines(text):
    return [l.strip() for l in text.splitlines() if l.startswith("wrote block") or "tapemark at" in l or "observed flux transitions" in l or "average peak height is" in l]


def main():
    rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    emul_every = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    bad = 0
    for i in range(n):
        kw = dict(seed=int(rng.integers(1, 1 << 30)), nblocks=int(rng.integers(1, 9)), minwords=1, maxwords=int(rng.integers(1, 20)),
                  marks_every=int(rng.integers(0, 4)), gap_samples=int(rng.choice([90, 130, 200, 400, 900])), noise_mv=float(rng.choice([5, 20, 50, 90])),
                  jitter=float(rng.choice([0.0, 0.02, 0.06])), amp_slope=float(rng.choice([0.02, -0.08, -0.14])), amplitude=float(rng.choice([1.3, 2.0, 3.0])))
        dsk = bool(rng.integers(0, 3) == 0)
        if dsk:                                           # heads out of line, -deskew (delays and pulse heights learned on the first blocks)
            kw["skew_cells"] = tuple(float(x) for x in rng.uniform(0, float(rng.choice([0.1, 0.35])), size=6))
        tape = synth.ww_tape(**kw)
        if rng.integers(0, 2):
            tape.rows = (-tape.rows.astype(np.int32)).clip(-32767, 32767).astype(np.int16)
        fd = str(rng.choice(["neg", "pos", "auto"]))
        rev = bool(rng.integers(0, 4) == 0)
        opts = [f"-fluxdir={fd}"] + (["-reverse"] if rev else []) + (["-deskew"] if dsk else [])
        with tempfile.TemporaryDirectory() as wd:
            tape.write(os.path.join(wd, "t.tbin"))
            p = subprocess.run([REF, "-v", "-tap", "-nolabels", "-nm"] + opts + ["t"], cwd=wd, capture_output=True, text=True)
            rtap = open(os.path.join(wd, "t.tap"), "rb").read() if os.path.exists(os.path.join(wd, "t.tap")) else b""
            q = subprocess.run([ORACLE, "-v", f"-out={wd}/o"] + opts + [os.path.join(wd, "t.tbin")], capture_output=True, text=True)
            otap = open(os.path.join(wd, "o.tap"), "rb").read() if os.path.exists(os.path.join(wd, "o.tap")) else b""
            msgs = []
            if p.returncode == 0 and (rtap != otap or lines(p.stdout) != lines(open(os.path.join(wd, "o.log")).read())):
                msgs.append("oracle != reference")
            if p.returncode == 0 and i % emul_every == 0:
                emul_frontend = None if os.environ.get("FUZZ_GPU") else __import__("emul_util").emul_frontend      # (FUZZ_GPU=1: the real kernels)
                try:
                    pipeline.decode_tape_ww(tape.spec.header(), tape.rows, os.path.join(wd, "g.tap"), log_path=os.path.join(wd, "g.log"), fluxdir=fd, reverse=rev, deskew=dsk,
                                            fe_factory=emul_frontend, chunk_rows=int(rng.choice([256, 1000, 4096])))
                    if open(os.path.join(wd, "g.tap"), "rb").read() != rtap or lines(open(os.path.join(wd, "g.log")).read()) != lines(p.stdout):
                        msgs.append("device path != reference")
                except Exception as e:
                    msgs.append(f"device path raised {e!r}")
            print(("BAD " if msgs else "ok  ") + f"{i:3d} rc {p.returncode} {kw} {opts} blocks {len(lines(p.stdout))} {msgs}", flush=True)
            bad += bool(msgs)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
