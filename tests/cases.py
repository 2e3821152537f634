"""The parity case table shared by the golden-vector generator (tests/make_goldens.py), the
oracle-vs-reference tests and the GPU parity tests.  Each case = a synthetic tape recipe + the
reference command-line options it is decoded with (option coverage follows the reference's own
examples/*/Makefile test commands)."""
import numpy as np

from readtape_amd import synth, tbin


def _nrzi_small(seed, **kw):
    return synth.nrzi_tape(seed=seed, nblocks=2, minlen=40, maxlen=90, marks_every=2, gap_samples=1500, **kw)


def case_nrzi9(seed=11):
    return _nrzi_small(seed)


def case_nrzi9_weak(seed=12):
    # one weak track + more noise so that parmset 0 fails on some blocks and -m retries
    t = synth.nrzi_tape(seed=seed, nblocks=3, minlen=40, maxlen=80, gap_samples=1500,
                        amplitude=0.9, amp_slope=0.12, noise_mv=60.0, jitter=0.06)
    return t


def case_nrzi7(seed=13):
    return synth.nrzi_tape(seed=seed, nblocks=2, minlen=40, maxlen=90, marks_every=2, ntrks=7, gap_samples=1500)


def case_nrzi9_skew(seed=14):
    return _nrzi_small(seed, skew_cells=(0.0, 0.10, 0.05, 0.15, 0.0, 0.20, 0.10, 0.05, 0.12))


def case_pe(seed=15):
    return synth.pe_tape(seed=seed, nblocks=2, minlen=30, maxlen=60, gap_samples=1500)


def case_gcr(seed=16):
    return synth.gcr_tape(seed=seed, nblocks=2, minlen=40, maxlen=160, gap_samples=2500)


def case_ww(seed=41):
    return synth.ww_tape(seed=seed, nblocks=5, minwords=3, maxwords=14, marks_every=2, gap_samples=700)


def case_ww_unused(seed=47):
    # eight heads of which two are not Whirlwind tracks ('x' in the order string, src/readtape.c:891): one carries noise, one a copy of a data track
    import dataclasses
    t = synth.ww_tape(seed=seed, nblocks=5, minwords=3, maxwords=12, marks_every=2, gap_samples=650)
    rng = np.random.default_rng(seed)
    n = t.rows.shape[0]
    noise = np.round(rng.normal(0.0, 0.4 / t.spec.maxvolts * 32767, n)).astype(np.int16)
    cols = [t.rows[:, 0], noise, t.rows[:, 1], t.rows[:, 2], t.rows[:, 3], t.rows[:, 4], t.rows[:, 1].copy(), t.rows[:, 5]]      # C x M L c m x l
    t.rows = np.ascontiguousarray(np.stack(cols, 1))
    t.spec = dataclasses.replace(t.spec, ntrks=8, trkorder="CxMLcmxl")
    return t


def case_ww_pos(seed=42):
    # the other polarity: the positive half of every pulse first
    t = synth.ww_tape(seed=seed, nblocks=4, minwords=3, maxwords=12, marks_every=3, gap_samples=700)
    t.rows = (-t.rows.astype(np.int32)).clip(-32767, 32767).astype(np.int16)
    return t


def case_ww_rough(seed=43):
    # noise, jitter and a weak alternate set of tracks: missing-bit / missing-clock warnings, odd lengths
    return synth.ww_tape(seed=seed, nblocks=6, minwords=2, maxwords=10, marks_every=2, gap_samples=500, noise_mv=60.0, jitter=0.05, amp_slope=-0.1)


def case_ww_close(seed=44):
    # blocks and block marks only a few bit times apart (the state that survives from block to block matters most here)
    return synth.ww_tape(seed=seed, nblocks=8, minwords=1, maxwords=4, marks_every=1, gap_samples=130)


def case_ww_skew(seed=45):
    # heads out of line by up to 0.3 cell; the tape ends before the -deskew calibration has its 1000 transitions per track
    return synth.ww_tape(seed=seed, nblocks=5, minwords=3, maxwords=12, marks_every=2, gap_samples=600, skew_cells=(0.0, 0.22, 0.10, 0.30, 0.05, 0.16))


def case_ww_skew_long(seed=46):
    # enough blocks for the calibration to stop by itself (block limit or transitions), with blocks and marks left over
    return synth.ww_tape(seed=seed, nblocks=16, minwords=12, maxwords=30, marks_every=5, gap_samples=400, noise_mv=25.0,
                         skew_cells=(0.12, 0.0, 0.28, 0.07, 0.2, 0.33))


DESKEW_RESTART_PARMS = ("parms active, clk_window, clk_alpha, agc_window, agc_alpha, min_peak, clk_factor, pulse_adj, pkww_bitfrac, pkww_rise, midbit, z1pt, z2pt, id\n"
                        "{1, 0, 0.2, 0, 0.8, 0.5, 0, 0.3, 1.395, 0.3, 0.5, 1.45, 2.35, PRM}\n")


def case_nrzi7_deskew_restart(seed=619833729):
    # (found by tests/stress_gpu.py, seed 810 tape 100) weak, noisy 7-track tape, a window of 27 samples: the first block's attempt
    # of the -deskew pre-pass crosses a device restart and is replayed a second time from an exact scan - the pre-pass' peak
    # statistics must not count its transitions twice (the reference goes on to the second block: the first has < 1000 per track)
    return synth.nrzi_tape(seed=seed, nblocks=3, minlen=16, maxlen=3000, marks_every=0, ntrks=7, gap_samples=4000, amplitude=0.6, noise_mv=50.0, jitter=0.0)


case_nrzi7_deskew_restart.parms_text = DESKEW_RESTART_PARMS


def case_gcr_noisy(seed=17):
    return synth.gcr_tape(seed=seed, nblocks=2, minlen=40, maxlen=120, gap_samples=2500, noise_mv=45.0, jitter=0.05, amplitude=1.2)


def case_gcr_errors(seed=19):
    # recorded-bit errors inside data groups (written after ECC/parity were formed): what -correct repairs
    # (src/decode_gcr.c:588-611).  One track in several characters, the parity track, the check character, two tracks
    # (in one character: parity still good; in different characters: not a one-track pattern), three tracks.
    rng = np.random.default_rng(seed + 3000)
    spec = synth.gcr_spec(seed=seed)
    pay = synth.random_payloads(rng, 2, 130, 170)
    flips0 = {1: [(0, 3), (2, 3), (5, 3)], 3: [(1, 8)], 5: [(7, 2)], 7: [(0, 1), (0, 6)], 9: [(1, 1), (4, 5)],
              11: [(2, 0), (2, 4), (2, 7)], 13: [(6, 0)], 15: [(0, 7), (1, 7), (2, 7), (3, 7), (4, 7), (5, 7), (6, 7), (7, 7)]}
    flips1 = {0: [(3, 5)], 2: [(3, 4), (7, 4)], 4: [(0, 8), (5, 8)], 6: [(2, 2), (3, 6), (4, 2)]}
    return synth.make_tape(spec, [("block", pay[0], flips0), ("block", pay[1], flips1)], gap_samples=2500)


def case_nrzi9_skew_long(seed=20):
    # enough transitions per track for the -deskew calibration to stop by itself (1000 per track, src/decoder.h:99)
    # before the tape ends: the blocks after that are decoded with the calibrated delays only
    return synth.nrzi_tape(seed=seed, nblocks=6, minlen=380, maxlen=520, marks_every=4, gap_samples=1500,
                           skew_cells=(0.0, 0.30, 0.12, 0.45, 0.05, 0.38, 0.20, 0.08, 0.26))


def case_gcr_skew(seed=21):
    return synth.gcr_tape(seed=seed, nblocks=3, minlen=60, maxlen=200, gap_samples=2500,
                          skew_cells=(0.0, 0.9, 0.3, 1.4, 0.2, 1.1, 0.6, 0.1, 0.8))


def _without_bpi(t):
    import dataclasses
    t.spec = dataclasses.replace(t.spec, bpi=0.0)            # the header says "density unknown": the decoder estimates it
    return t


def case_nrzi9_nobpi(seed=22):
    # > 9999 transitions: the density pre-pass stops by itself (ESTDEN_COUNTNEEDED, src/decoder.c:336)
    return _without_bpi(synth.nrzi_tape(seed=seed, nblocks=5, minlen=600, maxlen=900, marks_every=3, gap_samples=1500))


def case_nrzi9_nobpi_short(seed=23):
    # the tape ends before the pre-pass has seen enough: it uses what there is
    return _without_bpi(_nrzi_small(seed))


def case_nrzi9_clean(seed=24):
    # no noise: the gaps differentiate to exact zeros (dead band, src/readtape.c:1387), so the device may restart in them
    return synth.nrzi_tape(seed=seed, nblocks=4, minlen=40, maxlen=120, marks_every=3, gap_samples=1500, noise_mv=0.0)


def case_nrzi9_cut(seed=25):
    # ragged input: the recording starts and ends in the middle of a block
    import dataclasses
    t = synth.nrzi_tape(seed=seed, nblocks=3, minlen=100, maxlen=300, gap_samples=2000)
    return dataclasses.replace(t, rows=np.ascontiguousarray(t.rows[3000:t.rows.shape[0] - 4500]))


def case_noise_only(seed=26):
    import dataclasses
    t = _nrzi_small(seed)
    rng = np.random.default_rng(seed)
    return dataclasses.replace(t, rows=rng.normal(0, 60, size=(6000, 9)).astype(np.int16))


def case_tiny(seed=27):
    import dataclasses
    t = _nrzi_small(seed)
    return dataclasses.replace(t, rows=np.ascontiguousarray(t.rows[:7]))


def case_nrzi9_nobpi_skew(seed=28):
    # both pre-passes in one run: the density is estimated first, then the head skew is calibrated with it
    return _without_bpi(synth.nrzi_tape(seed=seed, nblocks=5, minlen=500, maxlen=800, marks_every=3, gap_samples=1500,
                                        skew_cells=(0.0, 0.30, 0.12, 0.45, 0.05, 0.38, 0.20, 0.08, 0.26)))


def case_nrzi9_oversampled(seed=18):
    # sampled at 640 ns (39 samples per bit) while the header says 1280 ns: what "-subsample=2" is for
    import dataclasses
    spec = synth.TapeSpec(mode=tbin.MODE_NRZI, ntrks=9, bpi=800.0, ips=50.0, tdelta_ns=640, maxvolts=4.4, pulse_w=0.22, seed=seed)
    rng = np.random.default_rng(seed + 1000)
    items = [("block", p) for p in synth.random_payloads(rng, 2, 40, 90, databits=8)] + [("mark",)]
    t = synth.make_tape(spec, items, gap_samples=3000)
    t.spec = dataclasses.replace(spec, tdelta_ns=1280)
    return t


def _reorder(t, order):
    """The same recording with its heads wired in another order: column i of the file carries the track `order[i]` names
    (a digit = that track, p = the parity track), which is what the reference's -order= undoes (src/readtape.c:877-915)."""
    import dataclasses
    n = t.rows.shape[1]
    h2t = [n - 1 if ch in "pP" else int(ch) for ch in order]
    # TBIN_NO_REORDER: "the columns were NOT put into canonical order when the file was made" - without it the reference
    # ignores -order= and the TBINORD extension altogether (src/readtape.c:1646-1648)
    return dataclasses.replace(t, rows=np.ascontiguousarray(t.rows[:, h2t]), spec=dataclasses.replace(t.spec, flags=t.spec.flags | tbin.FLAG_NO_REORDER))


def case_nrzi7_order(seed=13):
    return _reorder(case_nrzi7(seed), "543210p")           # examples/7trk_NRZI/Makefile:7


def case_pe_order(seed=15):
    return _reorder(case_pe(seed), "01234576p")            # examples/9trk_PE/Makefile:8


def case_gcr_order(seed=16):
    # the order travels in the file (TBINORD header extension), no -order= on the command line
    import dataclasses
    t = _reorder(case_gcr(seed), "p76543210")
    return dataclasses.replace(t, spec=dataclasses.replace(t.spec, trkorder="p76543210"))


def case_nrzi7_order_ignored(seed=13):
    # the same columns but WITHOUT TBIN_NO_REORDER: the reference ignores -order= (and decodes garbage)
    import dataclasses
    t = case_nrzi7_order(seed)
    return dataclasses.replace(t, spec=dataclasses.replace(t.spec, flags=0))


AGCFATAL_PARMS = ("parms active, clk_window, clk_alpha, agc_window, agc_alpha, min_peak, clk_factor, pulse_adj, pkww_bitfrac, pkww_rise, midbit, z1pt, z2pt, id\n"
                  "{1, 0, 0.2, 10, 0.0, 0.0, 0, 0.3, 0.164, 0.1, 0.5, 1.45, 2.35, PRM}\n")


def case_nrzi7_agcfatal(seed=416690828):
    # (found by tests/stress_gpu.py) a window AGC of 10 heights over a 3-sample peak window on the differentiated signal: the
    # ring still holds a start-up entry <= 0 when the first adjustment divides by its minimum, the gain goes negative and the
    # reference dies with "AGC gain bad in lookfor_peak" (src/decoder.c:782) after some 100 transitions
    return synth.nrzi_tape(seed=seed, nblocks=2, minlen=16, maxlen=200, ntrks=7, gap_samples=1500, amplitude=3.2, noise_mv=10.0, jitter=0.02)


case_nrzi7_agcfatal.parms_text = AGCFATAL_PARMS


AGCFATAL_M_PARMS = ("parms active, clk_window, clk_alpha, agc_window, agc_alpha, min_peak, clk_factor, pulse_adj, pkww_bitfrac, pkww_rise, midbit, z1pt, z2pt, id\n"
                    "{1, 0, 0.2, 3, 0.0, 0.5, 0, 0.3, 0.266, 0.3, 0.5, 1.45, 2.35, PRM}\n"
                    "{1, 0, 0.2, 3, 0.0, 0.0, 0, 0.3, 0.266, 0.05, 0.5, 1.45, 2.35, PRM}\n"
                    "{1, 0, 0.2, 10, 0.0, 0.0, 0, 0.3, 0.42, 0.2, 0.5, 1.45, 2.35, PRM}\n"
                    "{1, 0, 0.2, 3, 0.0, 0.1, 0, 0.3, 0.266, 0.2, 0.5, 1.45, 2.35, PRM}\n")


def case_nrzi9_agcfatal_m(seed=412346456):
    # (found by tests/stress_gpu.py, seed 704 tape 59: the GPU hung) a 2 us digitiser (3- and 5-sample windows), 50 mV of noise, a
    # sweep of four sets with window AGC and no min_peak: the gain of the SECOND set goes negative in an attempt that other sets
    # have already been through, and the reference dies there (src/decoder.c:782) - no further set is tried.  The screened walk's
    # "blind for ever" row (1 << 60) must not be narrowed to an int.
    import dataclasses
    spec = synth.TapeSpec(mode=tbin.MODE_NRZI, ntrks=9, bpi=800.0, ips=50.0, tdelta_ns=2000, maxvolts=4.4, pulse_w=0.22, seed=seed,
                          amplitude=2.5, noise_mv=50.0, jitter=0.02)
    items = [("block", pl) for pl in synth.random_payloads(np.random.default_rng(seed + 1000), 2, 16, 1200, databits=8)]
    t = synth.make_tape(spec, items, gap_samples=1920)
    return dataclasses.replace(t, rows=np.ascontiguousarray(t.rows[:3000]))


case_nrzi9_agcfatal_m.parms_text = AGCFATAL_M_PARMS


AVGHEIGHT_PARMS = ("parms active, clk_window, clk_alpha, agc_window, agc_alpha, min_peak, clk_factor, pulse_adj, pkww_bitfrac, pkww_rise, midbit, z1pt, z2pt, id\n"
                   "{1, 0, 0.2, 10, 0.0, 0.1, 0, 0.3, 0.266, 0.3, 0.5, 1.45, 2.35, PRM}\n"
                   "{1, 0, 0.2, 0, 0.5, 0.0, 0, 0.3, 0.42, 0.1, 0.5, 1.45, 2.35, PRM}\n"
                   "{1, 0, 0.2, 0, 0.2, 1.0, 0, 0.3, 0.42, 0.1, 0.5, 1.45, 2.35, PRM}\n")


def case_nrzi9_avgheight_fatal(seed=40247490):
    # (found by tests/stress_gpu.py, seed 901 tape 85) 80 mV of noise on a 1 V signal and a second set without min_peak: a track's
    # "peaks" 5 .. 15 are noise wiggles whose tops lie below their bottoms, the learned average peak height comes out negative and the
    # reference dies INSIDE the block decoder's callback: "avg peak-to-peak voltage isn't positive" (src/decode_nrzi.c:227; the same
    # assert in src/decode_gcr.c:862 and src/decode_pe.c:144).  3 000 rows of that tape: 11 813 transitions, then the assert.
    import dataclasses
    t = synth.nrzi_tape(seed=seed, nblocks=4, minlen=16, maxlen=7000, marks_every=3, ntrks=9, gap_samples=4000, amplitude=1.0, noise_mv=80.0, jitter=0.08)
    return dataclasses.replace(t, rows=np.ascontiguousarray(t.rows[25000:28000]))


case_nrzi9_avgheight_fatal.parms_text = AVGHEIGHT_PARMS


# name -> (tape builder, reference options, oracle options); a builder's .parms_text, if any, is the NRZI/PE/GCR.parms file of the run
CASES = {
    "nrzi9":        (case_nrzi9,      ["-nrzi"],                       []),
    "nrzi9_m":      (case_nrzi9_weak, ["-nrzi", "-m"],                 ["-m"]),
    "nrzi9_correct":(case_nrzi9_weak, ["-nrzi", "-m", "-correct"],     ["-m", "-correct"]),
    "nrzi7":        (case_nrzi7,      ["-nrzi", "-ntrks=7"],           []),
    "nrzi9_skew":   (case_nrzi9_skew, ["-nrzi", "-ntrks=9", "-skew=3,1,2,0,3,0,1,2,1"], ["-skew=3,1,2,0,3,0,1,2,1"]),
    "nrzi9_invert": (case_nrzi9,      ["-nrzi", "-invert"],            ["-invert"]),
    "nrzi9_sub2":   (case_nrzi9_oversampled, ["-nrzi", "-subsample=2"], ["-subsample=2"]),
    "nrzi9_zeros":  (case_nrzi9,      ["-nrzi", "-zeros"],             ["-zeros"]),
    "nrzi9_diffz":  (case_nrzi9,      ["-nrzi", "-zeros", "-differentiate"], ["-zeros", "-differentiate"]),
    "nrzi9_diffpk": (case_nrzi9,      ["-nrzi", "-differentiate"],     ["-differentiate"]),
    "nrzi9_diffpk_clean": (case_nrzi9_clean, ["-nrzi", "-differentiate", "-m"], ["-differentiate", "-m"]),
    "nrzi9_diffpk_skew": (case_nrzi9_clean, ["-nrzi", "-ntrks=9", "-differentiate", "-skew=3,1,2,0,3,0,1,2,1"], ["-differentiate", "-skew=3,1,2,0,3,0,1,2,1"]),
    "gcr_diffpk":   (case_gcr,        ["-gcr", "-differentiate"],      ["-differentiate"]),
    "pe_diffpk":    (case_pe,         ["-pe", "-differentiate"],       ["-differentiate"]),
    "pe":           (case_pe,         ["-pe"],                         []),
    "pe_m":         (case_pe,         ["-pe", "-m"],                   ["-m"]),
    "pe_zeros":     (case_pe,         ["-pe", "-zeros"],               ["-zeros"]),
    "pe_diffz":     (case_pe,         ["-pe", "-zeros", "-differentiate"], ["-zeros", "-differentiate"]),
    "gcr":          (case_gcr,        ["-gcr"],                        []),
    "gcr_m":        (case_gcr_noisy,  ["-gcr", "-m"],                  ["-m"]),
    "gcr_zeros":    (case_gcr,        ["-gcr", "-zeros"],              ["-zeros"]),
    "gcr_diffz":    (case_gcr,        ["-gcr", "-zeros", "-differentiate"], ["-zeros", "-differentiate"]),
    "nrzi9_deskew": (case_nrzi9_skew, ["-nrzi", "-deskew"],            ["-deskew"]),
    "nrzi9_deskew_long": (case_nrzi9_skew_long, ["-nrzi", "-deskew"],  ["-deskew"]),
    "nrzi7_deskew_restart": (case_nrzi7_deskew_restart, ["-nrzi", "-ntrks=7", "-deskew"], ["-ntrks=7", "-deskew"]),
    "gcr_deskew":   (case_gcr_skew,   ["-gcr", "-deskew"],             ["-deskew"]),
    "nrzi9_nobpi":  (case_nrzi9_nobpi, ["-nrzi"],                      []),
    "nrzi9_nobpi_short": (case_nrzi9_nobpi_short, ["-nrzi"],           []),
    "nrzi9_nobpi_deskew": (case_nrzi9_nobpi_skew, ["-nrzi", "-deskew", "-m"], ["-deskew", "-m"]),
    "nrzi9_cut":    (case_nrzi9_cut,  ["-nrzi"],                       []),
    "nrzi9_cut_zeros": (case_nrzi9_cut, ["-nrzi", "-zeros"],            ["-zeros"]),
    "noise_only":   (case_noise_only, ["-nrzi"],                       []),
    "tiny":         (case_tiny,       ["-nrzi"],                       []),
    "ww":           (case_ww,         [],                              []),
    "ww_auto":      (case_ww,         ["-fluxdir=auto"],               ["-fluxdir=auto"]),
    "ww_unused":    (case_ww_unused,  ["-fluxdir=auto"],               ["-fluxdir=auto"]),
    "ww_pos":       (case_ww_pos,     ["-fluxdir=pos"],                ["-fluxdir=pos"]),
    "ww_pos_auto":  (case_ww_pos,     ["-fluxdir=auto"],               ["-fluxdir=auto"]),
    "ww_wrongdir":  (case_ww_pos,     [],                              []),
    "ww_reverse":   (case_ww,         ["-reverse"],                    ["-reverse"]),
    "ww_rough":     (case_ww_rough,   [],                              []),
    "ww_close":     (case_ww_close,   ["-fluxdir=auto"],               ["-fluxdir=auto"]),
    "ww_deskew":    (case_ww_skew,    ["-fluxdir=auto", "-deskew"],    ["-fluxdir=auto", "-deskew"]),
    "ww_deskew_long": (case_ww_skew_long, ["-deskew"],                 ["-deskew"]),
    "ww_deskew_pos":  (case_ww_pos,   ["-fluxdir=pos", "-deskew"],     ["-fluxdir=pos", "-deskew"]),
    "nrzi7_order":  (case_nrzi7_order, ["-nrzi", "-ntrks=7", "-order=543210p"], ["-order=543210p"]),
    "pe_order":     (case_pe_order,   ["-pe", "-order=01234576p"],     ["-order=01234576p"]),
    "gcr_order_m":  (case_gcr_order,  ["-gcr", "-m"],                  ["-m"]),
    "nrzi7_order_ignored": (case_nrzi7_order_ignored, ["-nrzi", "-ntrks=7", "-order=543210p"], ["-order=543210p"]),
    "nrzi7_agcfatal": (case_nrzi7_agcfatal, ["-nrzi", "-ntrks=7", "-invert", "-differentiate"], ["-invert", "-differentiate"]),
    "nrzi9_agcfatal_m": (case_nrzi9_agcfatal_m, ["-nrzi", "-m", "-even"],  ["-m", "-even"]),
    "nrzi9_avgheight_fatal": (case_nrzi9_avgheight_fatal, ["-nrzi", "-m", "-even"],  ["-m", "-even"]),
    "gcr_errs":     (case_gcr_errors, ["-gcr"],                        []),
    "gcr_correct":  (case_gcr_errors, ["-gcr", "-correct"],            ["-correct"]),
}
# every reference run also gets: -v -tap -nolabels (SIMH .tap output, no IBM label handling);
# "-nm" is added when "-m" is absent because the reference retries by default (src/readtape.c:511)
