"""Build-container helper (needs /root/reference): randomized sweep of the CPU oracle against the UNMODIFIED reference
compiled by oracle/Makefile - .tap bytes, the event stream handed to the block decoders, and the per-block result lines -
over tape parameters and option combinations the committed cases do not pin.  usage: tests/fuzz_oracle.py <seed> <ntapes>"""
import os, subprocess, sys, tempfile, dataclasses
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))      # (this file lives in tests/: it runs the oracle, which only tests may)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import refdump
from readtape_amd import synth

REF = os.path.join(ROOT, "oracle", "_ref", "readtape_evt")
ORA = os.path.join(ROOT, "oracle", "_build", "oracle_readtape")
subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "all"], check=True)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
ntapes = int(sys.argv[2]) if len(sys.argv) > 2 else 20
bad = 0
LINES = lambda txt: [l.strip() for l in txt.splitlines() if l.startswith("wrote block") or "tapemark at" in l or "observed flux transitions" in l or "density was set to" in l or "average peak height is" in l]
for i in range(ntapes):
    kind = ["nrzi", "nrzi", "nrzi", "pe", "gcr"][int(rng.integers(0, 5))]
    amp = float(rng.choice([0.6, 1.0, 1.8, 2.5, 3.2])); noise = float(rng.choice([0.0, 5.0, 10.0, 25.0, 50.0, 80.0])); jit = float(rng.choice([0.0, 0.02, 0.05, 0.08]))
    kw = dict(amplitude=amp, noise_mv=noise, jitter=jit); seed = int(rng.integers(1, 1 << 30))
    ref, ora = [], []
    def both(o, r=None): ora.append(o); ref.append(r if r is not None else o)
    if kind == "nrzi":
        ntrks = int(rng.choice([9, 9, 7]))
        skc = tuple(float(x) for x in rng.choice([0.0, 0.1, 0.25, 0.4], size=ntrks)) if rng.random() < 0.3 else ()
        tape = synth.nrzi_tape(seed=seed, nblocks=int(rng.integers(1, 7)), minlen=8, maxlen=int(rng.choice([60, 300, 1500])), marks_every=int(rng.choice([0, 2])),
                               ntrks=ntrks, gap_samples=int(rng.choice([1200, 4000])), skew_cells=skc, **kw)
        if rng.random() < 0.15:                                 # another digitiser: 12..31 samples per bit cell, another full scale
            from readtape_amd import tbin as _tb
            spec = synth.TapeSpec(mode=_tb.MODE_NRZI, ntrks=ntrks, bpi=800.0, ips=50.0, tdelta_ns=int(rng.choice([800, 1000, 1600, 2000])),
                                  maxvolts=float(rng.choice([2.5, 4.4, 10.0])), pulse_w=0.22, seed=seed, **kw)
            r2 = np.random.default_rng(seed + 1000)
            items = [("block", pl) for pl in synth.random_payloads(r2, int(rng.integers(2, 6)), 16, int(rng.choice([200, 1200])), databits=ntrks - 1)]
            tape = synth.make_tape(spec, items, gap_samples=int(3000 * 1280 / spec.tdelta_ns))
        ref += ["-nrzi", f"-ntrks={ntrks}"]
        if ntrks == 7: ora.append("-ntrks=7")
        if rng.random() < 0.3: both("-m")
        if rng.random() < 0.2: both("-invert")
        if rng.random() < 0.2: both("-correct")
        if rng.random() < 0.1: both("-even")
        r = rng.random()
        if r < 0.15: both("-skew=" + ",".join(str(int(x)) for x in rng.integers(0, 6, size=ntrks)))
        elif r < 0.30: both("-deskew")
        r = rng.random()
        if r < 0.12: both("-zeros")
        elif r < 0.20: both("-differentiate")
        elif r < 0.26: both("-zeros"); both("-differentiate")
        if rng.random() < 0.12: tape.spec = dataclasses.replace(tape.spec, bpi=0.0)
        if rng.random() < 0.1: both("-subsample=2")
    elif kind == "pe":
        tape = synth.pe_tape(seed=seed, nblocks=int(rng.integers(1, 5)), minlen=20, maxlen=int(rng.choice([100, 600])), gap_samples=3000, **kw)
        ref.append("-pe")
        if rng.random() < 0.3: both("-m")
        r = rng.random()
        if r < 0.2: both("-zeros")
        elif r < 0.3: both("-differentiate")
        if rng.random() < 0.15: both("-deskew")
    else:
        kw["amplitude"] = max(amp, 1.0)
        tape = synth.gcr_tape(seed=seed, nblocks=int(rng.integers(1, 4)), minlen=20, maxlen=int(rng.choice([100, 600])), gap_samples=4000, **kw)
        ref.append("-gcr")
        if rng.random() < 0.3: both("-m")
        if rng.random() < 0.3: both("-correct")
        r = rng.random()
        if r < 0.15: both("-zeros")
        elif r < 0.25: both("-differentiate")
        if rng.random() < 0.15: both("-deskew")
    if rng.random() < 0.2:                                      # dropouts: stretches of a track at a fraction of its amplitude (AGC at its clamp,
        import dataclasses                                       # thresholds below the candidate screen, PE fake bits, NRZI corrections)
        rows2 = tape.rows.copy()
        for _ in range(int(rng.integers(1, 4))):
            t0 = int(rng.integers(0, rows2.shape[1])); a = int(rng.integers(0, max(1, rows2.shape[0] - 200))); b = min(rows2.shape[0], a + int(rng.choice([200, 800, 3000])))
            rows2[a:b, t0] = (rows2[a:b, t0].astype(np.float32) * float(rng.choice([0.5, 0.25, 0.1, 0.0]))).astype(np.int16)
        tape = dataclasses.replace(tape, rows=rows2)
    parms_text = None
    if rng.random() < 0.25:                                     # a <basename>.parms file with random front-end parameters (src/parmsets.c:337-372)
        base = {"nrzi": [0, 0.2, None, None, None, 0, 0.3, None, None, 0.5, 1.45, 2.35], "pe": [0, 0.2, None, None, None, 1.5, 0.4, None, None, 0, 1.45, 2.35],
                "gcr": [0, 0.015, None, None, None, 0, 0.3, None, None, 0, 1.45, 2.35]}[kind]
        spb = {"nrzi": 19.5, "pe": 19.5, "gcr": 13.8}[kind]
        lines = ["parms active, clk_window, clk_alpha, agc_window, agc_alpha, min_peak, clk_factor, pulse_adj, pkww_bitfrac, pkww_rise, midbit, z1pt, z2pt, id"]
        for _ in range(int(rng.integers(1, 5)) if "-m" in ref else 1):
            v = list(base)
            if rng.random() < 0.5: v[2], v[3] = 0, float(rng.choice([0.2, 0.3, 0.5, 0.8]))
            else: v[2], v[3] = int(rng.choice([1, 3, 5, 10])), 0.0
            v[4] = float(rng.choice([0.0, 0.1, 0.2, 0.5, 1.0]))
            v[7] = min(2.0, round(float(rng.choice([3, 5, 8, 9, 13, 20, 27, 38])) / spb + 0.01, 3))        # (pkww_bitfrac <= 2, src/parmsets.c:69)
            v[8] = float(rng.choice([0.05, 0.1, 0.14, 0.2, 0.3]))
            lines.append("{1, " + ", ".join(str(x) for x in v) + ", PRM}")
        parms_text = "\n".join(lines) + "\n"
    if rng.random() < 0.15:                                     # ragged: cut somewhere
        n = tape.rows.shape[0]; a, b = sorted(int(x) for x in rng.integers(0, n, size=2))
        if b - a > 50: tape = dataclasses.replace(tape, rows=np.ascontiguousarray(tape.rows[a:b]))
    if os.environ.get("FUZZ_ONLY") and int(os.environ["FUZZ_ONLY"]) != i: continue
    with tempfile.TemporaryDirectory() as wd:
        tape.write(os.path.join(wd, "t.tbin"))
        if parms_text:
            open(os.path.join(wd, "t.parms"), "w").write(parms_text); ora.append(f"-parms={wd}/t.parms"); ref.append("(t.parms)")
        ropts = ["-v", "-tap", "-nolabels"] + [o for o in ref if o != "(t.parms)"] + ([] if "-m" in ref else ["-nm"])
        pr = subprocess.run([REF] + ropts + ["t"], cwd=wd, env=dict(os.environ, RT_EVENT_DUMP=os.path.join(wd, "t.ref.evt")), capture_output=True, text=True)
        po = subprocess.run([ORA, "-v", f"-out={wd}/o", f"-evt={wd}/o.evt"] + ora + [os.path.join(wd, "t.tbin")], capture_output=True, text=True)
        msgs = []
        if po.returncode not in (0, 99): msgs.append(f"oracle rc {po.returncode}: {po.stderr[-200:]}")
        if (pr.returncode == 0) != (po.returncode == 0): msgs.append(f"exit codes differ: reference {pr.returncode}, oracle {po.returncode}")
        if os.path.exists(os.path.join(wd, "o.evt")) and os.path.exists(os.path.join(wd, "t.ref.evt")):
            a, b = refdump.load(os.path.join(wd, "o.evt")), refdump.load(os.path.join(wd, "t.ref.evt"))
            if pr.returncode != 0: n = min(a.size, b.size); a, b = a[:n], b[:n]
            msgs += refdump.compare(a, b)[:3]
        if pr.returncode == 0 and po.returncode == 0:
            rt = open(os.path.join(wd, "t.tap"), "rb").read() if os.path.exists(os.path.join(wd, "t.tap")) else b""
            ot = open(os.path.join(wd, "o.tap"), "rb").read() if os.path.exists(os.path.join(wd, "o.tap")) else b""
            if rt != ot: msgs.append(f".tap differs ({len(rt)} vs {len(ot)} bytes)")
            lo = LINES(open(os.path.join(wd, "o.log")).read()) if os.path.exists(os.path.join(wd, "o.log")) else []
            if LINES(pr.stdout) != lo: msgs.append("block log lines differ")
        tag = f"{i:3d} {kind} seed {seed} amp {amp} noise {noise} jit {jit} rows {tape.rows.shape[0]} bpi {tape.spec.bpi} ref {ref} rc {pr.returncode}"
        print(("FAIL " if msgs else "ok   ") + tag, flush=True)
        for m in msgs: print("     ", m[:300])
        if msgs and os.environ.get("FUZZ_ONLY"): print(pr.stdout[-1500:]); print(pr.stderr[-500:]); print(parms_text); print(po.stderr[-300:])
        bad += bool(msgs)
print(f"{ntapes - bad}/{ntapes} identical")
sys.exit(1 if bad else 0)
