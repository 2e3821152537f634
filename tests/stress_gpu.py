"""GPU box helper: randomized parity sweep (device events vs the CPU oracle) over tape parameters the committed tests do
not pin: amplitudes, noise, jitter, track counts, skews, parameter-set sweeps.  Prints one line per tape; exits non-zero
on the first mismatch."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))      # (this file lives in tests/: it runs the oracle, which only tests may)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
from parity_util import check_tape, config_for, oracle_attempts
from readtape_amd import frontend, synth, pipeline, tbin
sys.path.insert(0, os.path.join(ROOT, "tools"))
import refdump, subprocess
from parity_util import ORACLE, build_oracle
build_oracle()

def e2e_check(hdr, rows, opts, wd, parms_text=None):
    """-zeros: the device emits every confirmed crossing and the host replay applies the slope gate, so the comparison is
    end to end: .tap bytes and the stream of transitions the decoders were handed, against the oracle's."""
    tbin.write_tbin(os.path.join(wd, "t.tbin"), hdr, rows)
    popt = []
    if parms_text:
        open(os.path.join(wd, "p.parms"), "w").write(parms_text); popt = [f"-parms={wd}/p.parms"]
    p = subprocess.run([ORACLE, "-v", f"-out={wd}/o", f"-evt={wd}/o.evt"] + opts + popt + [os.path.join(wd, "t.tbin")], capture_output=True, text=True)
    skew = next(([int(x) for x in a[6:].split(",")] for a in opts if a.startswith("-skew=")), None)
    try:
        st, _ = pipeline.decode_tape(hdr, rows, os.path.join(wd, "g.tap"), log_path=os.path.join(wd, "g.log"), evt_path=os.path.join(wd, "g.evt"),
                                     opts=pipeline.DecodeOptions(multiple_tries="-m" in opts, correct="-correct" in opts, even_parity="-even" in opts), skew=skew, invert="-invert" in opts,
                                     find_zeros="-zeros" in opts, differentiate="-differentiate" in opts, deskew="-deskew" in opts,
                                     subsample=next((int(a[11:]) for a in opts if a.startswith("-subsample=")), 1), parms_text=parms_text,
                                     fe_factory=(__import__("emul_util").emul_frontend if os.environ.get("STRESS_EMUL") else None))
    except pipeline.ReferenceFatal as e:                       # the AGC assert (src/decoder.c:782): everything in front of it was delivered
        a, b = refdump.load(os.path.join(wd, "g.evt")), refdump.load(os.path.join(wd, "o.evt"))
        msgs = [] if p.returncode == 99 else [f"pipeline raised {e!r}, oracle rc {p.returncode}"]
        if a.size != b.size: msgs.append(f"{a.size} transitions delivered before the fatal assert vs the oracle's {b.size}")
        n = min(a.size, b.size)
        return msgs + refdump.compare(a[:n], b[:n]), {"events": int(e.stats["events_delivered"]), "speculative": None, "flags": None}
    except RuntimeError as e:                                  # what is fatal in the reference (exit 99) must be fatal here too
        ok = p.returncode == 99 and ("no transitions" in str(e) or "non-standard" in str(e) or "non-positive" in str(e))
        return ([] if ok else [f"pipeline raised {e!r}, oracle rc {p.returncode}"]), {"events": 0, "speculative": None, "flags": None}
    msgs = []
    a, b = refdump.load(os.path.join(wd, "g.evt")), refdump.load(os.path.join(wd, "o.evt"))
    if p.returncode == 0:
        if open(os.path.join(wd, "g.tap"), "rb").read() != open(os.path.join(wd, "o.tap"), "rb").read(): msgs.append(".tap differs")
    else:
        n = min(a.size, b.size); a, b = a[:n], b[:n]
    ign = ("v_avg_height",) if ("-zeros" in opts and "-differentiate" in opts) else ()
    msgs += refdump.compare(a, b, ignore_fields=ign)
    return msgs, {"events": int(st["events_delivered"]), "speculative": None, "flags": None, "exact_scans": int(st["exact_scans"])}

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
        tape = synth.nrzi_tape(seed=seed, nblocks=int(rng.integers(2, 9)), minlen=16, maxlen=int(rng.choice([200, 1200, 3000, 3000, 7000])),
                               marks_every=int(rng.choice([0, 3])), ntrks=ntrks, gap_samples=int(rng.choice([1500, 4000])), **kw)
        if rng.random() < 0.15:                                 # another digitiser: 12..31 samples per bit cell, another full scale
            from readtape_amd import tbin as _tb
            spec = synth.TapeSpec(mode=_tb.MODE_NRZI, ntrks=ntrks, bpi=800.0, ips=50.0, tdelta_ns=int(rng.choice([800, 1000, 1600, 2000])),
                                  maxvolts=float(rng.choice([2.5, 4.4, 10.0])), pulse_w=0.22, seed=seed, **kw)
            r2 = np.random.default_rng(seed + 1000)
            items = [("block", pl) for pl in synth.random_payloads(r2, int(rng.integers(2, 6)), 16, int(rng.choice([200, 1200])), databits=ntrks - 1)]
            tape = synth.make_tape(spec, items, gap_samples=int(3000 * 1280 / spec.tdelta_ns))
        if ntrks == 7: opts.append("-ntrks=7")
        if rng.random() < 0.3: opts.append("-m")
        if rng.random() < 0.2: opts.append("-invert")
        if rng.random() < 0.25 and ntrks == 9: opts.append("-skew=" + ",".join(str(int(x)) for x in rng.integers(0, int(rng.choice([6, 6, 20, 51])), size=9)))
        if rng.random() < 0.15 and "-m" not in opts: opts.append("-zeros")
        r = rng.random()
        if r < 0.08:
            opts.append("-differentiate")
            # (peak detection on the differentiated signal with noise above the dead band is one sequential burst per
            #  attempt - exact but slow, DESIGN.md §8: keep those tapes short and single-set so a sweep stays in minutes)
            if "-zeros" not in opts and noise >= 10.0:
                if "-m" in opts: opts.remove("-m")
                tape = synth.nrzi_tape(seed=seed, nblocks=2, minlen=16, maxlen=200, ntrks=ntrks, gap_samples=1500, **kw)
        elif r < 0.16 and not any(o.startswith("-skew") for o in opts) and "-zeros" not in opts: opts.append("-deskew")
        elif r < 0.24 and "-zeros" not in opts:
            import dataclasses
            tape.spec = dataclasses.replace(tape.spec, bpi=0.0); opts.append("(nobpi)")
        elif r < 0.32: opts.append("-correct")
        elif r < 0.38: opts.append("-even")
        elif r < 0.44: opts.append("-subsample=" + str(int(rng.choice([2, 3]))))
    elif kind == "pe":
        tape = synth.pe_tape(seed=seed, nblocks=int(rng.integers(2, 5)), minlen=30, maxlen=int(rng.choice([200, 900])), gap_samples=3000, **kw)
        if rng.random() < 0.3: opts.append("-m")
        elif rng.random() < 0.3: opts.append("-zeros")
    else:
        kw["amplitude"] = max(amp, 1.0)
        tape = synth.gcr_tape(seed=seed, nblocks=int(rng.integers(2, 4)), minlen=40, maxlen=int(rng.choice([200, 900])), gap_samples=4000, **kw)
        if rng.random() < 0.3: opts.append("-m")
        if rng.random() < 0.3: opts.append("-correct")
    if rng.random() < 0.2:                                      # dropouts: stretches of a track at a fraction of its amplitude (AGC at its clamp,
        import dataclasses                                       # thresholds below the candidate screen, PE fake bits, NRZI corrections)
        rows2 = tape.rows.copy()
        for _ in range(int(rng.integers(1, 4))):
            t0 = int(rng.integers(0, rows2.shape[1])); a = int(rng.integers(0, max(1, rows2.shape[0] - 200))); b = min(rows2.shape[0], a + int(rng.choice([200, 800, 3000])))
            rows2[a:b, t0] = (rows2[a:b, t0].astype(np.float32) * float(rng.choice([0.5, 0.25, 0.1, 0.0]))).astype(np.int16)
        tape = dataclasses.replace(tape, rows=rows2)
    parms_text = None
    if rng.random() < 0.2:                                      # a .parms file with random front-end parameters (window 3..47 samples, either AGC flavour)
        base = {"nrzi": [0, 0.2, None, None, None, 0, 0.3, None, None, 0.5, 1.45, 2.35], "pe": [0, 0.2, None, None, None, 1.5, 0.4, None, None, 0, 1.45, 2.35],
                "gcr": [0, 0.015, None, None, None, 0, 0.3, None, None, 0, 1.45, 2.35]}[kind]
        spb = {"nrzi": 19.5, "pe": 19.5, "gcr": 13.8}[kind]
        lines = ["parms active, clk_window, clk_alpha, agc_window, agc_alpha, min_peak, clk_factor, pulse_adj, pkww_bitfrac, pkww_rise, midbit, z1pt, z2pt, id"]
        for _ in range(int(rng.integers(1, 5)) if "-m" in opts else 1):
            v = list(base)
            if rng.random() < 0.5: v[2], v[3] = 0, float(rng.choice([0.2, 0.3, 0.5, 0.8]))
            else: v[2], v[3] = int(rng.choice([1, 3, 5, 10])), 0.0
            v[4] = float(rng.choice([0.0, 0.1, 0.2, 0.5, 1.0]))
            v[7] = min(2.0, round(float(rng.choice([3, 5, 8, 9, 13, 20, 27, 38])) / spb + 0.01, 3))        # (pkww_bitfrac <= 2, src/parmsets.c:69)
            v[8] = float(rng.choice([0.05, 0.1, 0.14, 0.2, 0.3]))
            lines.append("{1, " + ", ".join(str(x) for x in v) + ", PRM}")
        parms_text = "\n".join(lines) + "\n"
        opts.append("(parms)")
    if rng.random() < 0.15:                                     # ragged: a recording that starts and ends anywhere
        import dataclasses
        n = tape.rows.shape[0]; a, b = sorted(int(x) for x in rng.integers(0, n, size=2))
        if b - a > 2000: tape = dataclasses.replace(tape, rows=np.ascontiguousarray(tape.rows[a:b]))
    if os.environ.get("STRESS_SHAPES"):                        # (round 6: tests/fuzz_util.py's shapes over a share of the tapes - a generator of its own, off by default: see rng2 below)
        import dataclasses
        from fuzz_util import shape_rows
        rng3 = np.random.default_rng((int(sys.argv[1]) if len(sys.argv) > 1 else 1) * 200003 + i)
        if rng3.random() < 0.6:
            rows3, _ = shape_rows(tape.rows, rng3, density=float(rng3.choice([0.05, 0.15, 0.4])), wild=float(rng3.choice([0.0, 0.5, 1.0])))
            tape = dataclasses.replace(tape, rows=rows3); kind += "+shapes"
    hdr = tape.spec.header()
    # the peak path's knobs: the chains' general step for every detection / the general sift kernel, at random
    seg, warm = int(rng.choice([0, 1])), int(rng.choice([0, 1]))
    os.environ["RTFE_GAIN_FAST"] = str(1 - seg)
    if warm: os.environ["RTFE_SIFT_GENERIC"] = "1"
    else: os.environ.pop("RTFE_SIFT_GENERIC", None)
    # ... and the segments of the chains' steady stretches: short ones, warm-ups too short to join
    os.environ["RTFE_SEG_RECS"] = str(rng.choice([128, 128, 32, 16, 1024]))
    if rng.random() < 0.3: os.environ["RTFE_SEG_WARM"] = str(rng.choice([0, 2, 8]))
    else: os.environ.pop("RTFE_SEG_WARM", None)
    # (round 5: the walkers trust fewer rows of a record's margin block and make the rest from the samples.  A generator of its own: the tapes of
    #  a seed stay the tapes earlier rounds' findings are replayed by - test_emulated_dense_path_where_the_sample_path_underflows)
    rng2 = np.random.default_rng((int(sys.argv[1]) if len(sys.argv) > 1 else 1) * 100003 + i)
    if rng2.random() < 0.25: os.environ["RTFE_PK_MAR"] = str(rng2.choice([0, 1, 2]))
    else: os.environ.pop("RTFE_PK_MAR", None)
    if os.environ.get("STRESS_ONLY") and int(os.environ["STRESS_ONLY"]) != i:
        if rng.random() < 0.3: rng.choice([8, 24])             # (the draws the dense path's run of a tape makes below: the tapes behind it stay the same)
        continue
    if os.environ.get("STRESS_DRY"):
        print(i, kind, "seed", seed, "amp", amp, "noise", noise, "jit", jit, "opts", opts, "seg", seg, warm, "rows", tape.rows.shape[0], "spec", tape.spec.bpi, "parms", repr(parms_text), flush=True)
        continue
    import contextlib
    keep = os.environ.get("STRESS_KEEP")                        # (debugging: leave the oracle's and the pipeline's files in this directory)
    if keep: os.makedirs(keep, exist_ok=True)
    with (contextlib.nullcontext(keep) if keep else tempfile.TemporaryDirectory()) as wd:
        att = oracle_attempts(hdr, tape.rows, opts, wd) if not any(o in opts for o in ("(parms)", "-zeros", "-differentiate", "-deskew", "(nobpi)", "-correct", "-even", "-subsample=2", "-subsample=3")) else []
        r_peak, r_samp = None, None
        for rec in ("default", "1", "0", "0d"):                 # every format on all paths (peak path / sample path / the dense sample path, rtfe_dense.hip)
            t_case = time.perf_counter()
            if rec != "default": os.environ["RTFE_PEAK_PATH"] = rec[0]
            else: os.environ.pop("RTFE_PEAK_PATH", None)
            os.environ["RTFE_DENSE_PATH"] = "1" if rec == "0d" else "0"
            if rec == "0d" and rng.random() < 0.3: os.environ["RTFE_DS_WARM"] = str(rng.choice([8, 24]))      # (joins that fail)
            else: os.environ.pop("RTFE_DS_WARM", None)
            e2e = any(o in opts for o in ("(parms)", "-zeros", "-differentiate", "-deskew", "(nobpi)", "-correct", "-even", "-subsample=2", "-subsample=3"))
            if e2e: msgs, stats = e2e_check(hdr, tape.rows, [o for o in opts if o not in ("(nobpi)", "(parms)")], wd, parms_text)
            else:
                if os.environ.get("STRESS_EMUL"):
                    from emul_util import emul_frontend
                    fe = emul_frontend(config_for(hdr, opts))
                else: fe = frontend.FrontEnd(config_for(hdr, opts))
                msgs, stats = check_tape(fe, hdr, tape.rows, att)
                if rec != "default":                                # the two paths against each other, byte for byte, everywhere (not only where an attempt looks)
                    r = fe.scan(tape.rows).fetch()
                    if rec == "1": r_peak = r
                    elif rec == "0": r_samp = r
                    if rec == "0d": r_peak = r_samp                  # (the dense path against the sample path)
                    if rec != "1" and r_peak is not None:
                        cfgp = config_for(hdr, opts)
                        # (the dense path never raises RTFE_F_SCREEN_UNDERFLOW: its lists are used only where the thresholds lie inside their band, which begins at
                        #  the screen's level, and its literal detector has no screen - k_decode's flag only asks for an exact rescan of what is exact already)
                        # (... and since round 6 a handle's first scan of the PEAK path builds its screen for a floor it estimates from the samples, the sample path's for the floor the
                        #  handle was made with: a tape the estimate is too high for - 80 mV of noise under the fuzzer's shapes, seed 3300 tape 29 - is flagged on one path alone)
                        fmask = np.uint32(0xffffffff ^ frontend.F_SCREEN_UNDERFLOW)
                        if r.nbursts != r_peak.nbursts or any((r.bursts[k] != r_peak.bursts[k]).any() for k in ("zone_end", "reset_sample", "safe_last", "end_sample")) or ((r.bursts["flags"] & fmask) != (r_peak.bursts["flags"] & fmask)).any(): msgs.append("burst tables of the two paths differ")
                        else:
                            for bb in range(r.nbursts):
                                # (a burst k_decode flagged RTFE_F_SCREEN_UNDERFLOW holds what its screen let through - the host rescans it exactly, and
                                #  check_tape above compared THAT with the oracle; the dense path's events of the same burst are the exact ones already)
                                if rec == "0d" and (int(r_peak.bursts["flags"][bb]) & frontend.F_SCREEN_UNDERFLOW) and not (int(r.bursts["flags"][bb]) & frontend.F_SCREEN_UNDERFLOW):      # (flagged on both paths: k_decode's burst on both - an exact start, a chain that gave up)
                                    # ... so the dense path's burst is held against an exact rescan with the screen off: the literal detector from the same restart row
                                    rx = fe.scan_exact(tape.rows, int(r.bursts["reset_sample"][bb]), int(r.bursts["end_sample"][bb]), screen_off=True).fetch()
                                    for pp in range(len(cfgp.parmsets)):
                                        for tt in range(cfgp.ntrks):
                                            if r.track_events(bb, pp, tt).tobytes() != rx.track_events(0, pp, tt).tobytes(): msgs.append(f"dense path differs from the exact rescan: burst {bb} parmset {pp} track {tt}")
                                    continue
                                if (int(r.bursts["flags"][bb]) ^ int(r_peak.bursts["flags"][bb])) & frontend.F_SCREEN_UNDERFLOW: continue      # (one of them holds what its screen let through: check_tape held the exact rescan against the oracle)
                                for pp in range(len(cfgp.parmsets)):
                                    for tt in range(cfgp.ntrks):
                                        if r.track_events(bb, pp, tt).tobytes() != r_peak.track_events(bb, pp, tt).tobytes(): msgs.append(f"paths differ: burst {bb} parmset {pp} track {tt}")
            tag = f"{i:3d} {kind} seed {seed} amp {amp} noise {noise} jit {jit} opts {opts} general_step {seg} generic_sift {warm} peak_path {rec}: attempts {len(att)} events {stats['events']} speculative {stats.get('speculative')} flags {stats.get('flags')}"
            print(("FAIL " if msgs else "ok   ") + tag + f" [{time.perf_counter() - t_case:.1f} s]", flush=True)
            if msgs:
                print("\n".join(msgs[:6]))
                bad += 1
sys.exit(1 if bad else 0)
