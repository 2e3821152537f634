"""Kernel-logic parity on the CPU: the HIP kernel sources compiled against tests/cpu_emul and run
on threads, checked event-for-event against the oracle on the golden tapes.  (The real GPU run of
the same checks is tests/test_gpu_parity.py, marked gpu.)"""
import numpy as np
import pytest

from emul_util import emul_frontend
from golden_util import load_case
from parity_util import check_tape, config_for, oracle_attempts
from readtape_amd import frontend

PEAK_CASES = ["nrzi9", "nrzi9_m", "nrzi7", "nrzi9_skew", "nrzi9_invert", "pe", "pe_m", "gcr", "gcr_m", "nrzi7_order", "pe_order"]


@pytest.mark.parametrize("name", PEAK_CASES)
def test_emulated_kernels_match_oracle(name, tmp_path):
    g = load_case(name)
    att = oracle_attempts(g["hdr"], g["rows"], g["oracle_opts"], str(tmp_path))
    fe = emul_frontend(config_for(g["hdr"], g["oracle_opts"]))
    msgs, stats = check_tape(fe, g["hdr"], g["rows"], att)
    print(name, stats)
    assert not msgs, "\n".join(msgs[:12])
    assert stats["events"] > 0


@pytest.mark.parametrize("name", ["nrzi9_m", "pe", "gcr_m", "nrzi9_zeros", "tiny", "noise_only"])
def test_emulated_event_lists_packed_on_the_device(name, tmp_path, monkeypatch):
    """rtfe_pack_events (k_pack_plan + k_pack_copy behind the scan): the host reads every list from the packed buffer under the re-based burst table -
    the oracle's events, and the lists of the arena itself."""
    import numpy as np
    g = load_case(name)
    cfg = config_for(g["hdr"], g["oracle_opts"])
    monkeypatch.setenv("RTFE_PACK_EVENTS", "0")
    fe = emul_frontend(cfg)
    plain = fe.scan(g["rows"]).fetch()
    lists0 = [[[plain.track_events(b, p, t).copy() for t in range(cfg.ntrks)] for p in range(len(cfg.parmsets))] for b in range(plain.nbursts)]
    monkeypatch.setenv("RTFE_PACK_EVENTS", "1")
    res = fe.scan(g["rows"]).fetch()
    assert res.nbursts == plain.nbursts
    if res.nbursts:
        assert res._events.shape[0] == int((res.bursts["event_cap"].astype(np.int64) * len(cfg.parmsets) * cfg.ntrks).sum())
        assert res._events.shape[0] <= plain._events.shape[0]
    for b in range(res.nbursts):
        for p in range(len(cfg.parmsets)):
            for t in range(cfg.ntrks):
                assert np.array_equal(res.track_events(b, p, t), lists0[b][p][t]), (b, p, t)
    if "-zeros" not in g["oracle_opts"]:
        att = oracle_attempts(g["hdr"], g["rows"], g["oracle_opts"], str(tmp_path))
        msgs, stats = check_tape(fe, g["hdr"], g["rows"], att)
        assert not msgs, "\n".join(msgs[:12])


def test_emulated_screen_floor_follows_the_tape(tmp_path):
    """k_adapt_floor: behind a scan of the peak path the handle's candidate screen moves to half the smallest peak height the scan learned.  The events of
    the scans behind it are the oracle's all the same - also when the next tape is far weaker than the one the floor was learned on (its chains are flagged
    RTFE_F_SCREEN_UNDERFLOW, checked through exact rescans, and the floor comes down again) - and a floor the caller gave stands."""
    import dataclasses
    from readtape_amd import synth
    loud = synth.nrzi_tape(seed=311, nblocks=4, minlen=120, maxlen=300, gap_samples=3000, amplitude=3.2)
    weak = synth.nrzi_tape(seed=312, nblocks=4, minlen=120, maxlen=300, gap_samples=3000, amplitude=0.9)
    hdr = loud.spec.header()
    opts = ["-m"]
    cfg = config_for(hdr, opts)
    fe = emul_frontend(cfg)
    att_loud = oracle_attempts(hdr, loud.rows, opts, str(tmp_path))
    att_weak = oracle_attempts(hdr, weak.rows, opts, str(tmp_path))
    floors = []
    for rows, att in ((loud.rows, att_loud), (loud.rows, att_loud), (weak.rows, att_weak), (weak.rows, att_weak), (loud.rows, att_loud)):
        msgs, stats = check_tape(fe, hdr, rows, att)
        assert not msgs, "\n".join(msgs[:12])
        assert stats["events"] > 0
        st = fe.scan_stats(fe.scan(rows).fetch())
        floors.append((st["screen_floor_now"], stats["exact"], bool(stats["flags"] & frontend.F_SCREEN_UNDERFLOW)))
    print(floors)
    assert floors[0][0] > 1.5 and floors[1][0] == floors[0][0]            # learned on the loud tape (peaks of 6.4 V peak to peak: the floor is capped at 4 V)
    assert floors[2][0] < floors[1][0]                                     # the weak tape brought it down
    fixed = emul_frontend(dataclasses.replace(cfg, screen_floor_height=1.25))
    for rows, att in ((loud.rows, att_loud), (loud.rows, att_loud)):
        msgs, stats = check_tape(fixed, hdr, rows, att)
        assert not msgs
        assert fixed.scan_stats(fixed.scan(rows).fetch())["screen_floor_now"] == 1.25


def test_emulated_first_scan_estimates_the_screen_floor_from_the_samples(tmp_path, monkeypatch):
    """k_scan_begin: a handle's FIRST scan of the peak path looks at the samples before it screens them - windows inside blocks, the smallest peak-to-peak
    range of a track - and builds its candidate screen for 0.45 x that instead of the 1 V every tape clears (round 5 learned the floor behind a scan: a tape's
    first scan paid the default).  The events are the oracle's whatever the estimate; rtfe_reset_floor makes the next scan a first scan again;
    RTFE_FLOOR_PROBE=0 is round 5's behaviour; a floor the caller gave stands."""
    import dataclasses
    from readtape_amd import synth
    loud = synth.nrzi_tape(seed=321, nblocks=4, minlen=120, maxlen=300, gap_samples=3000, amplitude=3.2)
    weak = synth.nrzi_tape(seed=322, nblocks=4, minlen=120, maxlen=300, gap_samples=3000, amplitude=0.9)
    noisy = synth.nrzi_tape(seed=323, nblocks=4, minlen=120, maxlen=300, gap_samples=3000, noise_mv=60.0)
    hdr = loud.spec.header()
    opts = ["-m"]
    cfg = config_for(hdr, opts)
    for tape, lo, hi in ((loud, 1.8, 4.0), (weak, 1.0, 1.0), (noisy, 1.2, 4.0)):
        att = oracle_attempts(hdr, tape.rows, opts, str(tmp_path))
        fe = emul_frontend(cfg)                                        # a fresh handle
        res = fe.scan(tape.rows).fetch()
        st = fe.scan_stats(res)
        print(st["screen_floor_used"], st["screen_floor_now"], st["min_learned_height"], st["redone"], st["bursts"])
        assert lo <= st["screen_floor_used"] <= hi, st
        # the estimate stays below what the chains then learn (else they would flag an underflow and be redone on the samples)
        assert st["min_learned_height"] is None or st["screen_floor_used"] <= 0.5 * st["min_learned_height"] * 1.2 or st["screen_floor_used"] == 1.0, st
        assert not (res.bursts["flags"] & frontend.F_SCREEN_UNDERFLOW).any()
        msgs, stats = check_tape(fe, hdr, tape.rows, att)
        assert not msgs, "\n".join(msgs[:12])
        assert stats["events"] > 0
    # the second scan uses what the first one's chains learned, a reset makes the next one a first scan again
    fe = emul_frontend(cfg)
    a = fe.scan_stats(fe.scan(loud.rows).fetch())
    b = fe.scan_stats(fe.scan(loud.rows).fetch())
    assert b["screen_floor_used"] == a["screen_floor_now"]
    fe.reset_floor()
    c = fe.scan_stats(fe.scan(loud.rows).fetch())
    assert c["screen_floor_used"] == a["screen_floor_used"] and c["screen_floor_now"] == a["screen_floor_now"]
    monkeypatch.setenv("RTFE_FLOOR_PROBE", "0")
    old = emul_frontend(cfg)
    assert old.scan_stats(old.scan(loud.rows).fetch())["screen_floor_used"] == 1.0
    monkeypatch.delenv("RTFE_FLOOR_PROBE")
    fixed = emul_frontend(dataclasses.replace(cfg, screen_floor_height=1.25))
    assert fixed.scan_stats(fixed.scan(loud.rows).fetch())["screen_floor_used"] == 1.25


@pytest.mark.parametrize("parallel", ["1", "0"])
@pytest.mark.parametrize("name", PEAK_CASES)
def test_emulated_peak_record_path_matches_oracle(name, parallel, tmp_path, monkeypatch):
    """The peak path (k_sift -> k_zones -> k_gain -> k_emit, rtfe_sift.hip / rtfe_gain.hip): same events as the oracle, with the
    chains' steady-state fast path (events noted by k_gain, finished by k_emit) and with every detection through the general step."""
    monkeypatch.setenv("RTFE_PEAK_PATH", "1")
    monkeypatch.setenv("RTFE_GAIN_FAST", parallel)
    g = load_case(name)
    att = oracle_attempts(g["hdr"], g["rows"], g["oracle_opts"], str(tmp_path))
    fe = emul_frontend(config_for(g["hdr"], g["oracle_opts"]))
    res = fe.scan(g["rows"]).fetch()
    st = fe.scan_stats(res)
    msgs, stats = check_tape(fe, g["hdr"], g["rows"], att)
    print(name, stats, st)
    assert not msgs, "\n".join(msgs[:12])
    assert stats["events"] > 0
    assert st["parallel"] + st["sequential"] > 0 or st["redone"] == st["bursts"], "the record chains did not run"
    if parallel == "0":
        assert st["parallel"] == 0


@pytest.mark.parametrize("name", ["nrzi9_m", "nrzi9_invert", "nrzi9_skew", "gcr_m", "pe"])
@pytest.mark.parametrize("fast", ["1", "0"])
def test_emulated_margins_from_the_samples(name, fast, tmp_path, monkeypatch):
    """A record carries the margins of its first four rows; what a walker asks beyond them it makes from the samples (run_margin,
    rtfe_gain.hip).  RTFE_PK_MAR=0: every margin from the samples - with -invert, deskew delays, a parameter sweep, and GCR / PE by force
    on the peak path (their windows hold a top and a bottom: long lead runs, tail rows that fire)."""
    monkeypatch.setenv("RTFE_PEAK_PATH", "1")
    monkeypatch.setenv("RTFE_PK_MAR", "0")
    monkeypatch.setenv("RTFE_GAIN_FAST", fast)
    g = load_case(name)
    att = oracle_attempts(g["hdr"], g["rows"], g["oracle_opts"], str(tmp_path))
    fe = emul_frontend(config_for(g["hdr"], g["oracle_opts"]))
    msgs, stats = check_tape(fe, g["hdr"], g["rows"], att)
    assert not msgs, "\n".join(msgs[:12])
    assert stats["events"] > 0


@pytest.mark.parametrize("name,knobs", [("nrzi9", {"RTFE_PK_SLOT": "64"}), ("gcr", {"RTFE_PK_SLOT": "256"})])
def test_emulated_peak_record_path_gives_up_cleanly(name, knobs, tmp_path, monkeypatch):
    """Lists that outgrow their pool slot are marked unavailable; the bursts that need them are redone on the samples: same events."""
    monkeypatch.setenv("RTFE_PEAK_PATH", "1")
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    g = load_case(name)
    att = oracle_attempts(g["hdr"], g["rows"], g["oracle_opts"], str(tmp_path))
    fe = emul_frontend(config_for(g["hdr"], g["oracle_opts"]))
    res = fe.scan(g["rows"]).fetch()
    st = fe.scan_stats(res)
    msgs, stats = check_tape(fe, g["hdr"], g["rows"], att)
    assert not msgs, "\n".join(msgs[:12])
    assert st["redone"] > 0


@pytest.mark.parametrize("name,knobs", [("nrzi9", {"RTFE_SIFT_GENERIC": "1"}),           # the general k_sift where k_sift_s would run
                                        ("nrzi9", {"RTFE_PEAK_PATH": "0"}),             # NRZI on the sample path (k_decode walks every sample)
                                        ("nrzi9_m", {"RTFE_PEAK_PATH": "0"})])
def test_emulated_paths_that_are_not_the_default(name, knobs, tmp_path, monkeypatch):
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    g = load_case(name)
    att = oracle_attempts(g["hdr"], g["rows"], g["oracle_opts"], str(tmp_path))
    fe = emul_frontend(config_for(g["hdr"], g["oracle_opts"]))
    msgs, stats = check_tape(fe, g["hdr"], g["rows"], att)
    assert not msgs, "\n".join(msgs[:12])
    assert stats["events"] > 0


@pytest.mark.parametrize("name", PEAK_CASES)
def test_emulated_peak_path_equals_the_sample_path(name, monkeypatch):
    """Both paths on one tape: the same burst table and, per (burst, parameter set, track), the same events byte for byte - also where no
    oracle attempt looks (behind the block ends; bursts the replay would rescan exactly).  gcr: a chain that reaches its steady state
    in the middle of a chunk (the hand-over to k_gain_s once skipped the records it had stepped over)."""
    g = load_case(name)
    cfg = config_for(g["hdr"], g["oracle_opts"])
    res = []
    for pp in ("0", "1"):
        monkeypatch.setenv("RTFE_PEAK_PATH", pp)
        fe = emul_frontend(cfg)
        res.append((fe, fe.scan(g["rows"]).fetch()))
    (f0, r0), (f1, r1) = res
    st = f1.scan_stats(r1)
    assert r0.nbursts == r1.nbursts and (st["parallel"] + st["sequential"] > 0 or st["redone"] == st["bursts"]), st
    for k in ("zone_first", "zone_end", "reset_sample", "safe_last", "end_sample", "flags"):
        assert (r0.bursts[k] == r1.bursts[k]).all(), k
    for b in range(r0.nbursts):
        for p in range(len(cfg.parmsets)):
            for t in range(cfg.ntrks):
                assert r0.track_events(b, p, t).tobytes() == r1.track_events(b, p, t).tobytes(), (b, p, t)


@pytest.mark.parametrize("name,peak", [("nrzi9", None), ("nrzi9_m", None), ("gcr", "1"), ("nrzi9_skew", None), ("nrzi7", None), ("pe", "1"), ("gcr_m", "1"), ("nrzi9_invert", None)])
def test_emulated_clear_flags_equal_a_pass_over_the_streams(name, peak, monkeypatch, capfd):
    """k_prep marks a record clear from its list neighbours (the next entry of its list, the first of the next tile's list, deferred
    candidates resolved on the way); k_prep_check (emulator only) redoes that as a pass over the finished streams, record by record
    with its successor, and reports every record the two disagree on."""
    monkeypatch.setenv("RTFE_PREP_CHECK", "1")
    if peak: monkeypatch.setenv("RTFE_PEAK_PATH", peak)
    g = load_case(name)
    fe = emul_frontend(config_for(g["hdr"], g["oracle_opts"]))
    res = fe.scan(g["rows"]).fetch()
    assert res.nbursts > 0
    err = capfd.readouterr().err
    assert "prep_check" not in err, err[:2000]


@pytest.mark.parametrize("knobs", [{}, {"RTFE_GAIN_FAST": "0"}, {"RTFE_SEG_RECS": "32"}, {"RTFE_SEG_RECS": "32", "RTFE_SEG_WARM": "3"}, {"RTFE_SEG_RECS": "32", "RTFE_SEG_WARM": "3", "RTFE_SEG_REJOIN": "0"}, {"RTFE_SEG_RECS": "16", "RTFE_SEG_WARM": "40"}, {"RTFE_SEG_RECS": "1024"}, {"RTFE_SEG_RECS": "32", "RTFE_SEG_CAP": "70"},
                                   {"RTFE_SIFT_GENERIC": "1"}, {"RTFE_PK_SLOT": "128"}, {"RTFE_PK_MAR": "0"}, {"RTFE_PK_MAR": "1", "RTFE_GAIN_FAST": "0"}])
def test_emulated_long_blocks(knobs, tmp_path, monkeypatch):
    """Blocks of 500-640 bytes: chains that cross many tiles (k_gain's heads, the steady stretches in segments, the tails).
    RTFE_SEG_RECS=32: ~20 segments per chain; with a warm-up of 3 records the joins fail and k_gain (mode 1) finishes the chains from
    the last proven state; RTFE_SEG_CAP: the segment table runs full and the chains that find no room are walked as a whole.  The events are
    the oracle's whatever the joins do.  RTFE_PK_MAR: the walkers trust fewer rows of a record's margin block than it holds (none; one) and make
    the other rows' margins from the samples - as they do for a fifth lead row and for tail rows."""
    from readtape_amd import synth
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    tape = synth.nrzi_tape(seed=31, nblocks=2, minlen=500, maxlen=640, gap_samples=1500)
    hdr = tape.spec.header()
    att = oracle_attempts(hdr, tape.rows, [], str(tmp_path))
    fe = emul_frontend(config_for(hdr, []))
    msgs, stats = check_tape(fe, hdr, tape.rows, att)
    print(knobs, stats)
    assert not msgs, "\n".join(msgs[:12])
    assert stats["events"] > 4000


def test_every_seam_position_inside_a_gap_keeps_every_burst(tmp_path):
    """Time shards: wherever the seam falls relative to an inter-block zone - in front of it, inside it (a few quiet
    chunks before its end: the right rank cannot qualify the zone, the left rank must own the burst), behind it - the two ranks'
    bursts and events together are the whole-tape scan's.  (k_bursts' ownership rule: a zone that ends fewer than gap_chunks chunks
    behind the seam belongs to the left rank.)"""
    from readtape_amd import shard
    g = load_case("nrzi9")
    rows = g["rows"]
    cfg = config_for(g["hdr"], g["oracle_opts"])
    fe = emul_frontend(cfg)
    whole = fe.scan(rows).fetch()
    wb = shard.absolute_bursts(whole, 0)
    we = shard.flatten_events(whole, wb, 0)
    key = lambda e: e[np.lexsort((e[:, 1], e[:, 0]))]
    zone = wb[1]                                                   # the zone between the first two blocks
    lo, hi = int(zone["zone_first"]) - 256, int(zone["zone_end"]) + 512
    # every chunk position around the zone's two ends (where the ownership rule decides), every eighth one in between (the thread
    # emulation takes seconds per scan; the GPU test of the same name sweeps every position)
    ze, zf = int(zone["zone_end"]) // 64 * 64, int(zone["zone_first"]) // 64 * 64
    cuts = sorted(set(range(lo // 64 * 64, hi, 512)) | set(range(ze - 10 * 64, ze + 4 * 64, 64)) | set(range(zf - 128, zf + 192, 64)))
    assert len(cuts) > 16
    for cut in cuts:
        left = fe.scan(rows[: cut + 4096], row_base=0, first_is_tape_start=True, own_rows=cut).fetch()
        lb = shard.absolute_bursts(left, 0); le = shard.flatten_events(left, lb, 0)
        right = fe.scan(rows[cut:], row_base=cut, first_is_tape_start=False).fetch()
        rb = shard.absolute_bursts(right, cut); re_ = shard.flatten_events(right, rb, 0)
        assert left.nbursts + right.nbursts == whole.nbursts, (cut, left.nbursts, right.nbursts, whole.nbursts)
        assert not ((np.concatenate([lb["flags"], rb["flags"]]) & ~np.uint32(1)).any()), cut
        got = np.concatenate([le, re_])
        assert got.shape == we.shape and (key(got) == key(we)).all(), cut


@pytest.mark.parametrize("tdelta_ns,ntrks", [(2600, 9), (2300, 9), (2000, 9), (1800, 7), (1600, 9), (1500, 7), (1450, 7), (1200, 9), (1150, 9), (1060, 7), (1000, 9), (800, 9)])
def test_emulated_sample_rates_and_the_lean_sift_kernels(tdelta_ns, ntrks, tmp_path):
    """Other digitisers: 800 BPI NRZI sampled every 0.8 .. 2.6 us gives window widths of 6 .. 21 samples - every instantiation of k_sift_s
    (6 .. 17, nine and seven tracks) and, at 21, the general kernel: the events are the oracle's."""
    from readtape_amd import synth, tbin
    spec = synth.TapeSpec(mode=tbin.MODE_NRZI, ntrks=ntrks, bpi=800.0, ips=50.0, tdelta_ns=tdelta_ns, maxvolts=4.4, pulse_w=0.22, seed=tdelta_ns)
    rng = np.random.default_rng(tdelta_ns)
    items = [("block", pl) for pl in synth.random_payloads(rng, 3, 40, 160, databits=ntrks - 1)]
    tape = synth.make_tape(spec, items, gap_samples=int(3000 * 1280 / tdelta_ns))
    hdr = tape.spec.header()
    opts = ["-ntrks=7"] if ntrks == 7 else []
    att = oracle_attempts(hdr, tape.rows, opts, str(tmp_path))
    fe = emul_frontend(config_for(hdr, opts))
    res = fe.scan(tape.rows).fetch()
    st = fe.scan_stats(res)
    msgs, stats = check_tape(fe, hdr, tape.rows, att)
    print(tdelta_ns, fe.widths, stats, st)
    assert not msgs, "\n".join(msgs[:12])
    assert stats["events"] > 500 and st["redone"] == 0 and st["parallel"] > 0, (stats, st)


def _same_results(cfg, r0, r1):
    assert r0.nbursts == r1.nbursts
    for k in ("zone_first", "zone_end", "reset_sample", "safe_last", "end_sample", "flags"):
        assert (r0.bursts[k] == r1.bursts[k]).all(), (k, r0.bursts[k], r1.bursts[k])
    assert (r0.counts == r1.counts).all()
    for b in range(r0.nbursts):
        for p in range(len(cfg.parmsets)):
            for t in range(cfg.ntrks):
                assert r0.track_events(b, p, t).tobytes() == r1.track_events(b, p, t).tobytes(), (b, p, t)


DENSE_CASES = ["gcr", "gcr_m", "pe", "pe_m", "gcr_order_m", "pe_order", "gcr_deskew", "gcr_errs", "nrzi9", "nrzi9_m", "nrzi9_skew", "nrzi9_invert", "nrzi7", "noise_only", "tiny", "nrzi9_cut"]


@pytest.mark.parametrize("name", DENSE_CASES)
def test_emulated_dense_path_equals_the_sample_path(name, monkeypatch):
    """rtfe_dense.hip (k_dseg + k_dchain) against k_decode on one tape: the same burst table and, per (burst, parameter set, track), the
    same events byte for byte.  The NRZI tapes take the dense path by force (RTFE_DENSE_PATH=1 with the peak path off)."""
    g = load_case(name)
    cfg = config_for(g["hdr"], g["oracle_opts"])
    monkeypatch.setenv("RTFE_PEAK_PATH", "0")
    res = []
    for dp in ("0", "1"):
        monkeypatch.setenv("RTFE_DENSE_PATH", dp)
        fe = emul_frontend(cfg)
        res.append((fe, fe.scan(g["rows"]).fetch()))
    (f0, r0), (f1, r1) = res
    st = f1.scan_stats(r1)
    assert st["parallel"] + st["sequential"] > 0 or st["redone"] == st["bursts"] or int(r1.counts.sum()) == 0, st      # (literal rows + events from records: the dense path ran)
    _same_results(cfg, r0, r1)


@pytest.mark.parametrize("kind,knobs", [("gcr", {}), ("pe", {}), ("gcr", {"RTFE_DS_WARM": "8"}), ("gcr", {"RTFE_DS_CAP": "3"}), ("pe", {"RTFE_DS_BAND_LO": "0.9"}),
                                        ("gcr", {"RTFE_DS_BAND_HI": "0.6"}), ("pe", {"RTFE_DS_QUIET_S": "0.0"}), ("gcr", {"RTFE_DENSE_DEDUP": "0"}), ("nrzi", {}), ("gcr", {"RTFE_DS_LEAN": "0"}), ("gcr", {"RTFE_DS_UP": "1"}), ("pe", {"RTFE_DS_UP": "3"}), ("pe", {"RTFE_DS_UP": "2"}), ("gcr", {"RTFE_DS_ORDER": "0"})])
def test_emulated_dense_sweep_equals_the_sample_path(kind, knobs, monkeypatch):
    """The shape bench.py's C4 runs - an eight-set sweep with four window widths, sets that the front end cannot tell apart - on noisy
    tapes with blocks long enough for chains to cross many sub-segments; knobs force what clean tapes rarely do: joins that fail (a
    warm-up of 8 rows), lists that run full, thresholds outside the band (doubts and literal stretches), no small-signal bands."""
    from readtape_amd import synth
    make = {"gcr": lambda: synth.gcr_tape(seed=31, nblocks=3, minlen=300, maxlen=700, gap_samples=9000, noise_mv=11.0),
            "pe": lambda: synth.pe_tape(seed=32, nblocks=3, minlen=150, maxlen=300, gap_samples=6000, noise_mv=7.0),
            "nrzi": lambda: synth.nrzi_tape(seed=33, nblocks=3, minlen=150, maxlen=300, gap_samples=5000, noise_mv=25.0)}[kind]
    tape = make()
    hdr = tape.spec.header()
    extra = [(1.4, 0.20, 0.2, 0.5, 0, 0.0), (1.6, 0.14, 0.0, 0.5, 0, 0.0), (1.5, 0.10, 0.1, 0.5, 0, 0.0), (1.3, 0.25, 0.2, 0.5, 0, 0.0)]
    base = list(frontend.DEFAULT_PARMSETS[hdr.mode])
    parmsets = (base + extra)[:8] if kind == "gcr" else base[:8]
    cfg = frontend.FrontEndConfig.from_header(hdr, nparmsets=len(parmsets), parmsets=parmsets)
    monkeypatch.setenv("RTFE_PEAK_PATH", "0")
    for k, v in knobs.items(): monkeypatch.setenv(k, v)
    res = []
    for dp in ("0", "1"):
        monkeypatch.setenv("RTFE_DENSE_PATH", dp)
        fe = emul_frontend(cfg)
        res.append((fe, fe.scan(tape.rows).fetch()))
    (f0, r0), (f1, r1) = res
    st = f1.scan_stats(r1)
    print(kind, knobs, st, int(r1.counts.sum()))
    assert st["redone"] == 0 and st["parallel"] + st["sequential"] > 0, st
    _same_results(cfg, r0, r1)


def _wpr_pair(cfg, rows, wpr, monkeypatch):
    monkeypatch.delenv("RTFE_BURSTS_WPR", raising=False)
    f0 = emul_frontend(cfg)
    r0 = f0.scan(rows).fetch()
    monkeypatch.setenv("RTFE_BURSTS_WPR", wpr)
    f1 = emul_frontend(cfg)
    r1 = f1.scan(rows).fetch()
    for k in ("event_base", "event_cap"):
        assert (r0.bursts[k] == r1.bursts[k]).all(), k
    _same_results(cfg, r0, r1)
    return r1


@pytest.mark.parametrize("wpr", ["1", "2"])
@pytest.mark.parametrize("name", ["nrzi9", "nrzi9_m", "gcr", "pe", "pe_zeros", "nrzi9_cut", "noise_only", "tiny", "nrzi7"])
def test_emulated_burst_search_a_workgroup_per_round(name, wpr, monkeypatch):
    """The zone search with a workgroup per round (k_bursts_cnt / _emit / _tail: what tapes of 5e7 rows and more take) against the single
    workgroup that does the rounds in turn: RTFE_BURSTS_WPR cuts the rounds down to a word or two of the quiet map (64 / 128 groups of 64
    rows), so that short tapes have several."""
    g = load_case(name)
    _wpr_pair(config_for(g["hdr"], g["oracle_opts"]), g["rows"], wpr, monkeypatch)


@pytest.mark.parametrize("wpr", ["1", "3", "8"])
def test_emulated_burst_search_many_rounds(wpr, monkeypatch):
    """... on a tape of 70 words of quiet map: zones that begin rounds in front of the round they end in (gaps of 3 .. 4 words), a tape that
    does not begin in a gap (the exact-start burst of round 0), rounds without any zone end."""
    from readtape_amd import synth
    tape = synth.nrzi_tape(seed=77, nblocks=14, minlen=200, maxlen=1500, marks_every=5, gap_samples=14000)
    hdr = tape.spec.header()
    cfg = frontend.FrontEndConfig.from_header(hdr, nparmsets=1)
    rows = tape.rows[15000:]                                   # (it begins inside the first block)
    r1 = _wpr_pair(cfg, rows, wpr, monkeypatch)
    assert rows.shape[0] >= 64 * 64 * 40 and r1.nbursts >= 14 and (int(r1.bursts["flags"][0]) & frontend.F_EXACT_START)
    _wpr_pair(cfg, tape.rows, wpr, monkeypatch)                # ... and in its first gap


@pytest.mark.parametrize("seed,index", [(1001, 32), (974, 74)])
def test_emulated_dense_path_where_the_sample_path_underflows(seed, index):
    """Two tapes of the round-4 GPU stress sweeps (tests/stress_gpu.py, replayed here on the emulator): small signals whose thresholds sink
    below the candidate screen.  k_decode flags such a burst RTFE_F_SCREEN_UNDERFLOW - its events are what the screen let through, the host
    rescans it exactly - while the dense path uses its lists only inside their band and the literal detector elsewhere: no flag, and its
    events must be the exact rescan's (screen off) byte for byte, which is what the stress script checks for flagged bursts."""
    import os, subprocess, sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, STRESS_EMUL="1", STRESS_ONLY=str(index))
    for k in ("RTFE_PEAK_PATH", "RTFE_DENSE_PATH", "RTFE_DS_WARM"): env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "stress_gpu.py"), str(seed), str(index + 1)], capture_output=True, text=True, env=env, timeout=900)
    lines = [l for l in p.stdout.splitlines() if l.startswith(("ok", "FAIL"))]
    assert p.returncode == 0 and len(lines) == 4 and all(l.startswith("ok") for l in lines), p.stdout[-2000:] + p.stderr[-2000:]
    assert "peak_path 0d" in lines[-1] and "flags 0" in lines[-1] and "flags 8" in lines[-2], lines
