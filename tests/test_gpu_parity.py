"""GPU parity (run with -m gpu on an MI355X): the HIP front end through the C ABI (librtfe.so) vs the
CPU oracle — bit-exact on every event field — on the committed golden tapes, on fresh synthetic
tapes, and at benchmark scale through size-independent properties."""
import numpy as np
import pytest

from golden_util import load_case
from parity_util import check_tape, config_for, oracle_attempts
from readtape_amd import frontend, synth

pytestmark = pytest.mark.gpu

PEAK_CASES = ["nrzi9", "nrzi9_m", "nrzi7", "nrzi9_skew", "nrzi9_invert", "pe", "pe_m", "gcr", "gcr_m", "nrzi7_order", "pe_order", "gcr_order_m", "nrzi7_order_ignored"]


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


@pytest.mark.parametrize("name", PEAK_CASES)
def test_golden_tapes(name, tmp_path, gpu):
    g = load_case(name)
    att = oracle_attempts(g["hdr"], g["rows"], g["oracle_opts"], str(tmp_path))
    fe = frontend.FrontEnd(config_for(g["hdr"], g["oracle_opts"]))
    msgs, stats = check_tape(fe, g["hdr"], g["rows"], att)
    assert not msgs, "\n".join(msgs[:12])
    assert stats["events"] > 0


@pytest.mark.parametrize("name,knobs", [("nrzi9", {"RTFE_PK_SLOT": "64"}), ("nrzi9_m", {"RTFE_PK_SLOT": "96"}),      # lists that outgrow their slot: bursts redone on the samples
                                        ("nrzi9", {"RTFE_SIFT_GENERIC": "1"}), ("nrzi7", {"RTFE_SIFT_GENERIC": "1"}),  # the general k_sift where k_sift_s would run
                                        ("nrzi9", {"RTFE_SIFT_PLAIN": "0"}), ("nrzi7", {"RTFE_SIFT_PLAIN": "0"}), ("nrzi9_m", {"RTFE_SIFT_PLAIN": "0"}),  # k_sift_s with its knobs at run time (what -invert, the tools' counters and cut-offs take)
                                        ("nrzi9", {"RTFE_GAIN_FAST": "0"}), ("nrzi9_m", {"RTFE_GAIN_FAST": "0"}),      # every detection through the chains' general step
                                        ("nrzi9_skew", {"RTFE_GAIN_FAST": "0", "RTFE_SIFT_GENERIC": "1"}),
                                        ("gcr", {"RTFE_PEAK_PATH": "1"}), ("gcr_m", {"RTFE_PEAK_PATH": "1", "RTFE_PK_SLOT": "256"}),
                                        ("pe", {"RTFE_PEAK_PATH": "1"}), ("pe_m", {"RTFE_PEAK_PATH": "1", "RTFE_GAIN_FAST": "0"}),
                                        ("nrzi9", {"RTFE_PEAK_PATH": "0"}), ("nrzi9_m", {"RTFE_PEAK_PATH": "0"}), ("nrzi7", {"RTFE_PEAK_PATH": "0"})])
def test_rare_paths_of_the_peak_path(name, knobs, tmp_path, gpu, monkeypatch):
    """The peak path's (k_sift -> k_gain -> k_emit) rare branches forced by knobs - pool slots too small (the burst is redone by
    k_decode), the general sift kernel, the chains' general step for every detection - and each format on the path that is not its
    default: the events must not change."""
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    g = load_case(name)
    att = oracle_attempts(g["hdr"], g["rows"], g["oracle_opts"], str(tmp_path))
    fe = frontend.FrontEnd(config_for(g["hdr"], g["oracle_opts"]))
    res = fe.scan(g["rows"]).fetch()
    st = fe.scan_stats(res)
    msgs, stats = check_tape(fe, g["hdr"], g["rows"], att)
    assert not msgs, "\n".join(msgs[:12])
    assert stats["events"] > 0
    if "RTFE_PK_SLOT" in knobs and name.startswith("nrzi"):
        assert st["redone"] > 0
    if knobs.get("RTFE_GAIN_FAST") == "0":
        assert st["parallel"] == 0 and (st["sequential"] > 0 or st["redone"] == st["bursts"])


@pytest.mark.parametrize("knobs", [{}, {"RTFE_GAIN_FAST": "0"}])
def test_parameter_sweep_on_the_peak_path(knobs, gpu, monkeypatch):
    """-m on NRZI: 8 parameter sets with three window widths go through the general k_sift (a list per width and head) and 72 chains
    per burst; bursts, counts and events are the sample path's, byte for byte, and no burst needs the sample path."""
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    tape = synth.nrzi_tape(seed=47, nblocks=12, minlen=64, maxlen=1500, gap_samples=4000, noise_mv=10.0)
    hdr = tape.spec.header()
    cfg = frontend.FrontEndConfig.from_header(hdr, parmsets=frontend.DEFAULT_PARMSETS[frontend.NRZI])
    out = []
    for pp in ("1", "0"):
        monkeypatch.setenv("RTFE_PEAK_PATH", pp)
        fe = frontend.FrontEnd(cfg)
        out.append(fe.scan(tape.rows).fetch())
        if pp == "1":
            st = fe.scan_stats(out[-1])
            assert st["redone"] == 0 and st["parallel"] + st["sequential"] > 50_000, st
    a, b = out
    assert a.nbursts == b.nbursts >= 12 and (a.counts == b.counts).all() and (a.bursts["flags"] == b.bursts["flags"]).all()
    for i in range(a.nbursts):
        for p in range(len(cfg.parmsets)):
            for t in range(hdr.ntrks):
                assert a.track_events(i, p, t).tobytes() == b.track_events(i, p, t).tobytes(), (i, p, t)


@pytest.mark.parametrize("knobs", [{}, {"RTFE_GAIN_FAST": "0"}, {"RTFE_SIFT_GENERIC": "1"}, {"RTFE_PK_SLOT": "128"}, {"RTFE_PEAK_PATH": "0"},
                                   {"RTFE_SEG_RECS": "64"}, {"RTFE_SEG_RECS": "64", "RTFE_SEG_WARM": "4"}, {"RTFE_SEG_RECS": "64", "RTFE_SEG_WARM": "4", "RTFE_SEG_REJOIN": "0"}, {"RTFE_SEG_REJOIN": "0"}, {"RTFE_SEG_RECS": "512", "RTFE_SEG_WARM": "16"}, {"RTFE_SEG_RECS": "1024"}, {"RTFE_SEG_RECS": "64", "RTFE_SEG_CAP": "300"}])
def test_long_blocks(knobs, tmp_path, gpu, monkeypatch):
    """Blocks of 1500-4096 bytes (chains of tens of thousands of records: k_gain's heads, k_gain_s' steady stretches across many
    tiles, the tails): the events are the oracle's, every block start speculative."""
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    tape = synth.nrzi_tape(seed=33, nblocks=10, minlen=1500, maxlen=4096, marks_every=4, gap_samples=4000)
    hdr = tape.spec.header()
    att = oracle_attempts(hdr, tape.rows, [], str(tmp_path))
    fe = frontend.FrontEnd(config_for(hdr, []))
    msgs, stats = check_tape(fe, hdr, tape.rows, att)
    assert not msgs, "\n".join(msgs[:12])
    assert stats["speculative"] == len(att) and stats["flags"] == 0


def test_long_blocks_peak_path_equals_sample_path(gpu, monkeypatch):
    """32 KB blocks (~1 M rows each): the peak path and the sample path (RTFE_PEAK_PATH=0: k_decode walks every sample) must
    produce the same bursts, counts and events, byte for byte."""
    import torch
    tape = synth.nrzi_tape(seed=35, nblocks=5, minlen=30000, maxlen=32768, gap_samples=6000)
    hdr = tape.spec.header()
    rows = torch.from_numpy(tape.rows).cuda()
    out = []
    for pp in (None, "0"):
        if pp is None: monkeypatch.delenv("RTFE_PEAK_PATH", raising=False)
        else: monkeypatch.setenv("RTFE_PEAK_PATH", pp)
        fe = frontend.FrontEnd(frontend.FrontEndConfig.from_header(hdr))
        out.append(fe.scan(rows).fetch())
        if pp is None:
            st = fe.scan_stats(out[-1])
            assert st["redone"] == 0 and st["parallel"] > 500_000, st
    a, b = out
    assert a.nbursts == b.nbursts >= 5 and (a.counts == b.counts).all() and (a.bursts["flags"] == b.bursts["flags"]).all()
    for i in range(a.nbursts):
        for t in range(hdr.ntrks):
            assert a.events(i, 0)[a.events(i, 0)["trk"] == t].tobytes() == b.events(i, 0)[b.events(i, 0)["trk"] == t].tobytes()
    assert int(a.counts.sum()) > 500_000


@pytest.mark.parametrize("seed,nblocks,maxlen", [(21, 12, 600), (22, 30, 2000), (23, 6, 4096)])
def test_fresh_nrzi_tapes(seed, nblocks, maxlen, tmp_path, gpu):
    tape = synth.nrzi_tape(seed=seed, nblocks=nblocks, minlen=16, maxlen=maxlen, marks_every=5, gap_samples=4000)
    hdr = tape.spec.header()
    att = oracle_attempts(hdr, tape.rows, [], str(tmp_path))
    fe = frontend.FrontEnd(config_for(hdr, []))
    msgs, stats = check_tape(fe, hdr, tape.rows, att)
    assert not msgs, "\n".join(msgs[:12])
    assert stats["speculative"] == len(att)          # clean tape: every block start lies in a proven-safe zone
    assert stats["flags"] == 0


def test_text_tape_with_a_silent_track(tmp_path, gpu):
    """7-bit text: one track carries no flux transitions inside the blocks (its walker never reaches the AGC's steady
    state); the other eight must still take the parallel tile path, and every event must match."""
    spec = synth.nrzi_spec(seed=51)
    rng = np.random.default_rng(51)
    items = [("block", bytes(rng.integers(0, 128, size=int(n), dtype=np.int64).astype(np.uint8))) for n in (700, 1500, 90, 2500)]
    tape = synth.make_tape(spec, items, gap_samples=4000)
    hdr = tape.spec.header()
    att = oracle_attempts(hdr, tape.rows, [], str(tmp_path))
    fe = frontend.FrontEnd(config_for(hdr, []))
    msgs, stats = check_tape(fe, hdr, tape.rows, att)
    assert not msgs, "\n".join(msgs[:12])
    assert stats["events"] > 0


def test_fresh_pe_tape(tmp_path, gpu):
    tape = synth.pe_tape(seed=31, nblocks=8, minlen=64, maxlen=1500, gap_samples=4000)
    hdr = tape.spec.header()
    att = oracle_attempts(hdr, tape.rows, [], str(tmp_path))
    fe = frontend.FrontEnd(config_for(hdr, []))
    msgs, stats = check_tape(fe, hdr, tape.rows, att)
    assert not msgs, "\n".join(msgs[:12])


def test_multi_parmset_sweep_reads_once(tmp_path, gpu):
    """8 parameter sets in one scan == 8 single-set scans (the batched sweep is exact per set)."""
    tape = synth.nrzi_tape(seed=41, nblocks=6, minlen=64, maxlen=800, gap_samples=4000, noise_mv=25.0)
    hdr = tape.spec.header()
    sets = frontend.DEFAULT_PARMSETS[frontend.NRZI]
    fe8 = frontend.FrontEnd(frontend.FrontEndConfig.from_header(hdr, parmsets=sets))
    r8 = fe8.scan(tape.rows).fetch()
    for p, ps in enumerate(sets):
        fe1 = frontend.FrontEnd(frontend.FrontEndConfig.from_header(hdr, parmsets=[ps], quiet_volts=0.49, gap_min_samples=0))
        r1 = fe1.scan(tape.rows).fetch()
        # same zones are not guaranteed (the quiet band depends on the set list), so compare per block by events
        e8 = np.concatenate([np.stack([r8.bursts[b]["reset_sample"] + r8.events(b, p)["sample"].astype(np.int64),
                                       r8.events(b, p)["trk"].astype(np.int64),
                                       r8.events(b, p)["v_peak"].view("u4").astype(np.int64)], 1) for b in range(r8.nbursts)])
        e1 = np.concatenate([np.stack([r1.bursts[b]["reset_sample"] + r1.events(b, 0)["sample"].astype(np.int64),
                                       r1.events(b, 0)["trk"].astype(np.int64),
                                       r1.events(b, 0)["v_peak"].view("u4").astype(np.int64)], 1) for b in range(r1.nbursts)])
        assert e8.shape == e1.shape and (e8 == e1).all(), f"parmset {p}"


def test_large_tape_properties(gpu):
    """Benchmark-scale input (tiled synthetic tape, > 1e7 rows): properties that need no oracle.
    * tiling k copies of a tape gives k copies of its events (shift invariance of the front end);
    * every burst is flag-free; per-track events are strictly ordered with spacing > left_distance."""
    torch = gpu
    base = synth.nrzi_tape(seed=51, nblocks=40, minlen=256, maxlen=2048, marks_every=10, gap_samples=6000)
    hdr = base.spec.header()
    fe = frontend.FrontEnd(config_for(hdr, []))
    r1 = fe.scan(base.rows).fetch()
    k = 12
    rows = torch.from_numpy(base.rows).cuda().repeat(k, 1).contiguous()
    rk = fe.scan(rows).fetch()
    n = base.rows.shape[0]
    assert rows.shape[0] == k * n and k * n > 1e7
    assert not (rk.bursts["flags"] & ~np.uint32(frontend.F_EXACT_START)).any()
    def flat(r, lo, hi):
        out = []
        for b in range(r.nbursts):
            ev = r.events(b, 0)
            a = r.bursts[b]["reset_sample"] + ev["sample"].astype(np.int64)
            m = (a >= lo) & (a < hi)
            out.append(np.stack([a[m] - lo, ev["trk"][m].astype(np.int64), ev["v_peak"][m].view("u4").astype(np.int64),
                                 ev["agc_gain"][m].view("u4").astype(np.int64), ev["left_distance"][m].astype(np.int64)], 1))
        return np.concatenate(out)
    ref = flat(r1, 0, n)
    for j in (0, 1, k // 2, k - 1):
        got = flat(rk, j * n, (j + 1) * n)
        assert got.shape == ref.shape and (got == ref).all(), f"copy {j}"


@pytest.mark.parametrize("name", ["nrzi9", "nrzi9_m", "nrzi9_correct", "nrzi7", "nrzi9_skew", "nrzi9_invert", "nrzi9_sub2", "pe", "pe_m", "nrzi9_zeros", "pe_zeros", "gcr", "gcr_m", "gcr_zeros", "gcr_errs", "gcr_correct", "nrzi9_deskew", "nrzi9_deskew_long", "nrzi7_deskew_restart", "gcr_deskew", "nrzi9_nobpi", "nrzi9_nobpi_short", "nrzi9_diffz", "pe_diffz", "gcr_diffz", "nrzi9_diffpk", "nrzi9_diffpk_clean", "nrzi9_diffpk_skew", "gcr_diffpk", "pe_diffpk", "nrzi9_cut", "nrzi9_cut_zeros", "noise_only", "tiny", "nrzi9_nobpi_deskew", "nrzi7_order", "pe_order", "gcr_order_m", "nrzi7_order_ignored"])
def test_end_to_end_tap_bytes_match_reference(name, tmp_path, gpu):
    """GPU front end -> event replay -> block decoders -> SIMH .tap == the unmodified reference's .tap (golden)."""
    from test_emul_replay import decode_case
    g = load_case(name)
    tap, stats = decode_case(g, tmp_path, None)
    assert tap == g["tap"]
    assert stats["agc_mismatches"] == 0 and (stats["events_delivered"] > 0 or g["events"].size <= 1)
    assert not stats["event_diffs"], stats["event_diffs"]


def test_reference_agc_assert_stops_the_decode_where_the_reference_stops(tmp_path, gpu):
    """src/decoder.c:782: see tests/test_emul_replay.py (the same check on the real kernels)."""
    import os
    import refdump
    from readtape_amd import pipeline
    g = load_case("nrzi7_agcfatal")
    tap = os.path.join(str(tmp_path), "out.tap")
    with pytest.raises(pipeline.ReferenceFatal):
        pipeline.decode_tape(g["hdr"], g["rows"], tap, invert=True, differentiate=True, evt_path=tap + ".evt", parms_text=g["parms_text"])
    mine = refdump.load(tap + ".evt")
    assert mine.size == g["events"].size and not refdump.compare(mine, g["events"])


@pytest.mark.parametrize("peak_path", ["0", "1"])
def test_agc_assert_inside_a_parameter_sweep_ends_everything(peak_path, tmp_path, gpu, monkeypatch):
    """-m, the gain of the second set goes negative (stress seed 704 tape 59 hung the GPU): see tests/test_emul_replay.py."""
    import os
    import refdump
    from readtape_amd import pipeline
    monkeypatch.setenv("RTFE_PEAK_PATH", peak_path)
    g = load_case("nrzi9_agcfatal_m")
    tap = os.path.join(str(tmp_path), "out.tap")
    with pytest.raises(pipeline.ReferenceFatal):
        pipeline.decode_tape(g["hdr"], g["rows"], tap, evt_path=tap + ".evt", parms_text=g["parms_text"],
                             opts=pipeline.DecodeOptions(multiple_tries=True, even_parity=True))
    mine = refdump.load(tap + ".evt")
    assert mine.size == g["events"].size and not refdump.compare(mine, g["events"])


@pytest.mark.parametrize("peak_path", ["0", "1"])
def test_a_learned_peak_height_that_is_not_positive_ends_everything(peak_path, tmp_path, gpu, monkeypatch):
    """src/decode_nrzi.c:227: the reference exits inside the block decoder's callback (stress seed 901 tape 85): see tests/test_emul_replay.py."""
    import os
    import refdump
    from readtape_amd import pipeline
    monkeypatch.setenv("RTFE_PEAK_PATH", peak_path)
    g = load_case("nrzi9_avgheight_fatal")
    tap = os.path.join(str(tmp_path), "out.tap")
    with pytest.raises(pipeline.ReferenceFatal):
        pipeline.decode_tape(g["hdr"], g["rows"], tap, evt_path=tap + ".evt", parms_text=g["parms_text"],
                             opts=pipeline.DecodeOptions(multiple_tries=True, even_parity=True))
    mine = refdump.load(tap + ".evt")
    assert mine.size == g["events"].size and not refdump.compare(mine, g["events"])


@pytest.mark.parametrize("name", PEAK_CASES + ["nrzi9_nobpi", "nrzi9_cut", "noise_only", "tiny", "gcr_errs"])
def test_peak_record_path_equals_the_sample_path(name, gpu, monkeypatch):
    """The peak path (RTFE_PEAK_PATH=1: k_sift -> k_gain -> k_emit) against the sample path (RTFE_PEAK_PATH=0: k_decode): the same burst
    table and, per (burst, parameter set, track), the same events byte for byte - also behind the block ends, where no oracle
    attempt looks."""
    g = load_case(name)
    cfg = config_for(g["hdr"], g["oracle_opts"])
    res = []
    for pp in ("0", "1"):
        monkeypatch.setenv("RTFE_PEAK_PATH", pp)
        fe = frontend.FrontEnd(cfg)
        res.append((fe, fe.scan(g["rows"]).fetch()))
    (f0, r0), (f1, r1) = res
    st = f1.scan_stats(r1)
    assert r0.nbursts == r1.nbursts
    for k in ("zone_first", "zone_end", "reset_sample", "safe_last", "end_sample", "flags"):
        assert (r0.bursts[k] == r1.bursts[k]).all(), k
    for b in range(r0.nbursts):
        for p in range(len(cfg.parmsets)):
            for t in range(cfg.ntrks):
                assert r0.track_events(b, p, t).tobytes() == r1.track_events(b, p, t).tobytes(), (b, p, t)
    assert st["parallel"] + st["sequential"] > 0 or st["redone"] == st["bursts"] or int(r1.counts.sum()) == 0


def test_differentiated_peak_path_restarts_in_exact_zero_gaps(tmp_path, gpu):
    """-differentiate without -zeros: on a noise-free tape the gaps differentiate to exact zeros (dead band), every block
    becomes its own device burst and no attempt needs an exact rescan; with noise the whole tape is one burst."""
    from test_emul_replay import decode_case
    g = load_case("nrzi9_diffpk_clean")
    tap, stats = decode_case(g, tmp_path, None)
    assert tap == g["tap"] and stats["bursts"] >= 5 and stats["exact_scans"] == 0
    g = load_case("nrzi9_diffpk")
    tap, stats = decode_case(g, tmp_path, None)
    assert tap == g["tap"] and stats["bursts"] == 1


@pytest.mark.parametrize("name", ["nrzi9", "pe", "gcr"])
def test_short_burst_tails_do_not_change_the_tap(name, tmp_path, gpu, monkeypatch):
    """tail_rows absurdly short: the .tap does not change (a zone starts a whole quiet KiB after the last transition, by
    when the block has ended; otherwise the replay falls back to an exact rescan) - DESIGN.md §3 item 5."""
    from test_emul_replay import decode_case
    monkeypatch.setenv("RTFE_TAIL_ROWS", "8")
    g = load_case(name)
    tap, stats = decode_case(g, tmp_path, None)
    assert tap == g["tap"] and not stats["event_diffs"]


def test_zeros_excursions_without_events_are_history(tmp_path, gpu):
    from test_emul_replay import _noisy_pe_zeros_case
    _noisy_pe_zeros_case(tmp_path, None)


@pytest.mark.parametrize("kind,world", [("nrzi", 2), ("nrzi", 3), ("nrzi", 8), ("pe", 4), ("gcr", 3), ("nrzi_zeros", 4)])
def test_time_shards_on_one_gpu_equal_the_whole_scan(kind, world, gpu):
    """What the N ranks of `bench.py --gpus N` / readtape_amd.shard do, run one after another on this GPU: each shard scans
    its rows plus the right neighbour's halo and owns the bursts whose zone ends in its rows; together they must
    reproduce the whole-tape scan's events bit for bit (the seams fall wherever they fall: in blocks, in gaps)."""
    import torch
    from readtape_amd import shard
    if kind == "pe":
        tape = synth.pe_tape(seed=41, nblocks=14, minlen=200, maxlen=1500, gap_samples=5000)
    elif kind == "gcr":
        tape = synth.gcr_tape(seed=42, nblocks=10, minlen=400, maxlen=2500, gap_samples=7000)
    else:
        tape = synth.nrzi_tape(seed=43, nblocks=40, minlen=64, maxlen=3000, marks_every=6, gap_samples=5000)
    hdr = tape.spec.header()
    zeros = kind.endswith("_zeros")
    cfg = frontend.FrontEndConfig.from_header(hdr, find_zeros=zeros)
    fe = frontend.FrontEnd(cfg)
    rows = torch.from_numpy(tape.rows).cuda()
    whole = fe.scan(rows).fetch()
    wb = shard.absolute_bursts(whole, 0)
    we = shard.flatten_events(whole, wb, 0)
    halo = 1 << 18                                             # > the longest block + gap of these tapes
    parts_b, parts_e = [], []
    for rank, (lo, hi) in enumerate(shard.plan_shards(int(rows.shape[0]), world)):
        end = min(int(rows.shape[0]), hi + halo) if rank < world - 1 else hi
        res = fe.scan(rows[lo:end].contiguous(), row_base=lo, first_is_tape_start=(rank == 0), own_rows=hi - lo).fetch()
        b = shard.absolute_bursts(res, lo)
        parts_b.append(b); parts_e.append(shard.flatten_events(res, b, 0))
    got_b, got_e = np.concatenate(parts_b), np.concatenate(parts_e)
    for f in ("zone_end", "reset_sample", "safe_last", "end_sample"):
        assert list(got_b[f]) == list(wb[f]), f
    key = lambda e: e[np.lexsort((e[:, 1], e[:, 0]))]
    assert got_e.shape == we.shape and (key(got_e) == key(we)).all()
    assert we.shape[0] > 10000


def test_event_floods_are_decoded_not_dropped(tmp_path, gpu):
    from test_emul_replay import _event_flood_case
    _event_flood_case(tmp_path, None)


def test_deskew_calibration_on_a_growing_prefix(tmp_path, gpu):
    """-deskew: the pre-pass scans a prefix of the tape and grows it until the reference's stopping rule is met inside
    it; whatever the first prefix size, delays and .tap are the reference's."""
    from readtape_amd import pipeline
    g = load_case("nrzi9_deskew_long")
    ref_delays = [int(l.split("delayed by")[1].split()[0]) for l in g["blocklog"] if "observed flux transitions" in l]
    for first in (8192, 1 << 22):
        tap = str(tmp_path / f"d{first}.tap")
        stats, _ = pipeline.decode_tape(g["hdr"], g["rows"], tap, deskew=True, deskew_prefix_rows=first)
        assert stats["skew"] == ref_delays
        assert open(tap, "rb").read() == g["tap"]


def test_end_to_end_fresh_tape_vs_oracle_tap(tmp_path, gpu):
    """A fresh 60-block tape: .tap from the GPU pipeline == .tap from the CPU oracle; every block start speculative."""
    import subprocess
    from parity_util import ORACLE, build_oracle
    from readtape_amd import pipeline, tbin
    tape = synth.nrzi_tape(seed=61, nblocks=60, minlen=16, maxlen=3000, marks_every=7, gap_samples=5000)
    hdr = tape.spec.header()
    build_oracle()
    tbin.write_tbin(str(tmp_path / "t.tbin"), hdr, tape.rows)
    subprocess.run([ORACLE, f"-out={tmp_path}/o", str(tmp_path / "t.tbin")], check=True)
    stats, _ = pipeline.decode_tape(hdr, tape.rows, str(tmp_path / "g.tap"))
    assert open(tmp_path / "g.tap", "rb").read() == open(tmp_path / "o.tap", "rb").read()
    assert stats["exact_scans"] == 0 and stats["agc_mismatches"] == 0 and stats["blocks"] == 60


def _flat_all(r, p, lo, hi):
    out = []
    for b in range(r.nbursts):
        ev = r.events(b, p)
        a = r.bursts[b]["reset_sample"] + ev["sample"].astype(np.int64)
        m = (a >= lo) & (a < hi)
        out.append(np.stack([a[m] - lo, ev["trk"][m].astype(np.int64), ev["flags"][m].astype(np.int64), ev["v_peak"][m].view("u4").astype(np.int64),
                             ev["agc_gain"][m].view("u4").astype(np.int64), ev["left_distance"][m].astype(np.int64)], 1))
    return np.concatenate(out)


def test_config_c3_pe_zero_cross_at_scale(tmp_path, gpu):
    """BASELINE config 3 shape (9-track 1600 BPI PE, 640 ns, zero-cross path), scaled to ~2e7 rows by tiling:
    the base tape decodes to the oracle's .tap, and every copy of it yields the base tape's events."""
    import subprocess
    from parity_util import ORACLE, build_oracle
    from readtape_amd import pipeline, tbin
    torch = gpu
    base = synth.pe_tape(seed=71, nblocks=12, minlen=200, maxlen=1500, gap_samples=6000)
    hdr = base.spec.header()
    build_oracle()
    tbin.write_tbin(str(tmp_path / "t.tbin"), hdr, base.rows)
    subprocess.run([ORACLE, "-zeros", f"-out={tmp_path}/o", str(tmp_path / "t.tbin")], check=True)
    stats, r1 = pipeline.decode_tape(hdr, base.rows, str(tmp_path / "g.tap"), find_zeros=True)
    assert open(tmp_path / "g.tap", "rb").read() == open(tmp_path / "o.tap", "rb").read()
    fe = frontend.FrontEnd(frontend.FrontEndConfig.from_header(hdr, nparmsets=1, find_zeros=True))
    n = base.rows.shape[0]
    k = int(2e7 // n) + 1
    rk = fe.scan(torch.from_numpy(base.rows).cuda().repeat(k, 1).contiguous()).fetch()
    ref = _flat_all(r1, 0, 0, n)
    for j in (0, k // 2, k - 1):
        got = _flat_all(rk, 0, j * n, (j + 1) * n)
        assert got.shape == ref.shape and (got == ref).all(), f"copy {j}"


def test_config_c4_gcr_eight_parmset_sweep_at_scale(tmp_path, gpu):
    """BASELINE config 4 shape (9-track GCR 9042 BPI, 160 ns, 8-parmset batched sweep from a .parms text), tiled to
    ~1e7 rows: one scan yields, for every parameter set, k copies of the base tape's events; the base tape's events per
    set equal the oracle's single-set runs."""
    from parity_util import oracle_attempts
    torch = gpu
    base = synth.gcr_tape(seed=81, nblocks=6, minlen=300, maxlen=1500, gap_samples=8000)
    hdr = base.spec.header()
    sets = [(bf, rise, mp, al, 0, 0.0) for bf in (1.2, 1.5) for rise, mp in ((0.14, 0.0), (0.2, 0.2)) for al in (0.3, 0.5)]
    assert len(sets) == 8
    fe = frontend.FrontEnd(frontend.FrontEndConfig.from_header(hdr, parmsets=sets))
    r1 = fe.scan(base.rows).fetch()
    n = base.rows.shape[0]
    k = int(1e7 // n) + 1
    rk = fe.scan(torch.from_numpy(base.rows).cuda().repeat(k, 1).contiguous()).fetch()
    for p in range(8):
        ref = _flat_all(r1, p, 0, n)
        assert ref.shape[0] > 1000
        for j in (0, k - 1):
            got = _flat_all(rk, p, j * n, (j + 1) * n)
            assert got.shape == ref.shape and (got == ref).all(), f"parmset {p} copy {j}"
    # oracle cross-check of two of the sets on the base tape (each as the only set of a .parms file)
    for p in (0, 5):
        bf, rise, mp, al, _, _ = sets[p]
        parms = ("parms active, clk_window, clk_alpha, agc_window, agc_alpha, min_peak, pulse_adj, pkww_bitfrac, pkww_rise, z1pt, z2pt, id\n"
                 f"{{1, 0, 0.015, 0, {al}, {mp}, 0.3, {bf}, {rise}, 1.45, 2.35, PRM}}\n")
        (tmp_path / f"p{p}.parms").write_text(parms)
        att = oracle_attempts(hdr, base.rows, [f"-parms={tmp_path}/p{p}.parms"], str(tmp_path))
        fe1 = frontend.FrontEnd(frontend.FrontEndConfig.from_header(hdr, parmsets=[sets[p]]))
        msgs, stats = check_tape(fe1, hdr, base.rows, att)
        assert not msgs, "\n".join(msgs[:8])


@pytest.mark.parametrize("name", ["files_nrzi9_bin", "files_nrzi9_tap", "files_nrzi7_bin", "files_pe_m_tap", "files_gcr_bin"])
def test_output_files_and_summary_match_the_reference(name, tmp_path, monkeypatch, gpu):
    """The numbered .bin files / the lazily created .tap and the end-of-run report (src/readtape.c:1091-1111, 2021-2044)."""
    import os
    from golden_util import load_files_case, report_lines
    from readtape_amd import pipeline
    g = load_files_case(name)
    o = g["ref_opts"]
    monkeypatch.chdir(tmp_path)
    opts = pipeline.DecodeOptions(multiple_tries="-m" in o, verbose="-v" in o)
    pipeline.decode_tape(g["hdr"], g["rows"], None, log_path="t.log", opts=opts, out_base="t", in_name="t.tbin", tap_format="-tap" in o)
    made = sorted(f for f in os.listdir(".") if f.endswith(".bin") or f.endswith(".tap"))
    assert made == sorted(g["files"])
    for f in made:
        assert open(f, "rb").read() == g["files"][f], f
    assert report_lines(open("t.log").read()) == g["report"]


@pytest.mark.parametrize("name", ["nrzi9_zeros", "pe_zeros", "gcr_zeros", "nrzi9_cut_zeros"])
@pytest.mark.parametrize("knobs", [{}, {"RTFE_ZEROS_KERNEL": "0"}, {"RTFE_ZC_WARM": "16"}, {"RTFE_ZC_WARM": "64", "RTFE_TILE_ROWS": "512"}, {"RTFE_ZC_PARALLEL": "0"}])
def test_zeros_kernel_variants(name, knobs, tmp_path, gpu, monkeypatch):
    """-zeros: k_zeros (default) against the oracle is in test_golden_tapes / the replay tests; here the same tapes through k_decode's
    zero-crossing mode, with a warm-up too short to converge (joins fail and are repaired in place), with 512-row tiles, and with
    the sequential walk only: the events must not change."""
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    from test_emul_replay import decode_case
    g = load_case(name)
    tap, stats = decode_case(g, tmp_path, None)          # (the delivered transitions against the reference's event dump, and the .tap)
    assert tap == g["tap"]
    assert not stats["event_diffs"], stats["event_diffs"]


@pytest.mark.parametrize("kind", ["pe", "nrzi", "gcr"])
def test_zeros_kernel_equals_the_decode_mode_on_long_tapes(kind, gpu, monkeypatch):
    """k_zeros (two tracks per lane on packed 16-bit arithmetic, samples read from HBM, short warm-up with repairs) against k_decode's
    zero-crossing mode (512-row tiles in LDS, 64-row warm-up) and against the purely sequential walk, on tapes of a few million rows with
    weak and noisy stretches: the same burst table and the same events, byte for byte."""
    import torch
    make = {"pe": lambda: synth.pe_tape(seed=91, nblocks=60, minlen=300, maxlen=3000, gap_samples=7000, noise_mv=40.0, amp_slope=0.1),
            "nrzi": lambda: synth.nrzi_tape(seed=92, nblocks=60, minlen=300, maxlen=3000, marks_every=9, gap_samples=5000, noise_mv=40.0),
            "gcr": lambda: synth.gcr_tape(seed=93, nblocks=30, minlen=300, maxlen=3000, gap_samples=20000, noise_mv=40.0)}[kind]
    tape = make()
    hdr = tape.spec.header()
    rows = torch.from_numpy(tape.rows).cuda()
    cfg = frontend.FrontEndConfig.from_header(hdr, find_zeros=True)
    out = []
    for knobs in ({}, {"RTFE_ZEROS_KERNEL": "0", "RTFE_TILE_ROWS": "512", "RTFE_ZC_WARM": "64"}, {"RTFE_ZC_PARALLEL": "0"}):
        for k in ("RTFE_ZEROS_KERNEL", "RTFE_TILE_ROWS", "RTFE_ZC_WARM", "RTFE_ZC_PARALLEL"):
            monkeypatch.delenv(k, raising=False)
        for k, v in knobs.items():
            monkeypatch.setenv(k, v)
        fe = frontend.FrontEnd(cfg)
        out.append(fe.scan(rows).fetch())
    r0 = out[0]
    assert r0.nbursts > 20 and int(r0.counts.sum()) > 100000
    for r in out[1:]:
        assert r.nbursts == r0.nbursts and (r.counts == r0.counts).all()
        for k in ("zone_first", "zone_end", "reset_sample", "safe_last", "end_sample", "flags"):
            assert (r.bursts[k] == r0.bursts[k]).all(), k
        for b in range(r0.nbursts):
            for t in range(cfg.ntrks):
                assert r.track_events(b, 0, t).tobytes() == r0.track_events(b, 0, t).tobytes(), (b, t)


@pytest.mark.parametrize("ntrks,clip,order", [(9, True, None), (7, False, None), (8, True, None), (6, False, [5, 3, 1, 0, 2, 4]), (2, True, [1, 0]), (12, False, None), (19, True, None)])
def test_zeros_kernel_against_the_decode_mode(ntrks, clip, order, gpu, monkeypatch):
    """See tests/test_emul_replay.py (the same check on tapes thirty times as long, and the widest row the ABI takes)."""
    import torch
    import zeros_util
    hdr, rows = zeros_util.zeros_rows(ntrks, nblocks=120, clip=clip)
    variants = [{}, {"RTFE_ZC_WARM": "8"}, {"RTFE_ZC_WARM": "64"}, {"RTFE_ZEROS_KERNEL": "0"}, {"RTFE_ZC_PARALLEL": "0"}]
    out = zeros_util.scan_variants(frontend.FrontEnd, hdr, torch.from_numpy(rows).cuda(), monkeypatch, variants, head_to_trk=order)
    assert out[0].nbursts >= 100 and int(out[0].counts.sum()) > 200000
    for r in out[1:]:
        zeros_util.same_scan(out[0], r, ntrks)


@pytest.mark.parametrize("chunk_rows", [4096, 300])
@pytest.mark.parametrize("name", ["ww", "ww_auto", "ww_unused", "ww_pos", "ww_pos_auto", "ww_wrongdir", "ww_reverse", "ww_rough", "ww_close", "ww_deskew", "ww_deskew_long", "ww_deskew_pos"])
def test_whirlwind_tap_bytes_match_reference(name, chunk_rows, tmp_path, gpu):
    """Whirlwind through k_ww (rtfe_ww_scan: detector state handed from attempt to attempt): see tests/test_emul_replay.py."""
    from test_emul_replay import decode_ww_case
    g = load_case(name)
    tap, stats = decode_ww_case(g, tmp_path, None, chunk_rows)
    assert tap == g["tap"]
    assert stats["agc_mismatches"] == 0 and stats["events_delivered"] > 0
    assert not stats["event_diffs"], stats["event_diffs"]


def test_every_seam_position_inside_a_gap_keeps_every_burst(tmp_path, gpu):
    """Time shards: the cut swept chunk by chunk from in front of an inter-block zone to behind it - the two ranks'
    bursts and events together are the whole-tape scan's at every position (tests/test_emul_parity.py sweeps a subset on the emulator)."""
    from readtape_amd import shard
    g = load_case("nrzi9")
    rows = g["rows"]
    fe = frontend.FrontEnd(config_for(g["hdr"], g["oracle_opts"]))
    whole = fe.scan(rows).fetch()
    wb = shard.absolute_bursts(whole, 0)
    we = shard.flatten_events(whole, wb, 0)
    key = lambda e: e[np.lexsort((e[:, 1], e[:, 0]))]
    for zi in (1, 2):
        zone = wb[zi]
        lo, hi = int(zone["zone_first"]) - 256, int(zone["zone_end"]) + 512
        for cut in range(lo // 64 * 64, hi, 64):
            left = fe.scan(np.ascontiguousarray(rows[: cut + 4096]), row_base=0, first_is_tape_start=True, own_rows=cut).fetch()
            lb = shard.absolute_bursts(left, 0); le = shard.flatten_events(left, lb, 0)
            right = fe.scan(np.ascontiguousarray(rows[cut:]), row_base=cut, first_is_tape_start=False).fetch()
            rb = shard.absolute_bursts(right, cut); re_ = shard.flatten_events(right, rb, 0)
            assert left.nbursts + right.nbursts == whole.nbursts, (cut, left.nbursts, right.nbursts, whole.nbursts)
            got = np.concatenate([le, re_])
            assert got.shape == we.shape and (key(got) == key(we)).all(), cut


@pytest.mark.parametrize("name,nshards", [("nrzi9_deskew_long", 4), ("nrzi9", 3), ("gcr", 2), ("pe", 3)])
def test_fragments_concatenate_to_the_whole_tap(name, nshards, tmp_path, gpu):
    """pipeline.decode_tape_fragments on the GPU: the fragments' pieces concatenate to the reference's .tap."""
    from readtape_amd import pipeline, shard
    g = load_case(name)
    tap = str(tmp_path / "f.tap")
    spans = shard.plan_shards(g["rows"].shape[0], nshards, align=64)
    pipeline.decode_tape_fragments(g["hdr"], g["rows"], tap, spans, halo_rows=1024)
    want = g["tap"]
    if name == "nrzi9_deskew_long":                       # (its golden was made with -deskew: compare with the unsharded decode)
        pipeline.decode_tape(g["hdr"], g["rows"], str(tmp_path / "w.tap"))
        want = open(tmp_path / "w.tap", "rb").read()
    assert open(tap, "rb").read() == want


@pytest.mark.parametrize("name", ["nrzi9", "nrzi9_m", "pe_zeros", "gcr_m"])
def test_packed_event_fetch_holds_the_same_lists(name, gpu, monkeypatch):
    """ScanResult.fetch packs the event arena on the device before the copy (every burst's lists to the burst's longest list) and
    re-bases the host copy of the burst table: every (burst, parameter set, track) list is what the unpacked fetch returns."""
    g = load_case(name)
    cfg = config_for(g["hdr"], g["oracle_opts"])
    fe = frontend.FrontEnd(cfg)
    r = fe.scan(g["rows"])
    monkeypatch.setenv("RTFE_PACK_EVENTS", "0")
    r.fetch()
    plain = {(b, p, t): r.track_events(b, p, t).tobytes() for b in range(r.nbursts) for p in range(len(cfg.parmsets)) for t in range(cfg.ntrks)}
    cap_plain = r.bursts["event_cap"].copy()
    monkeypatch.setenv("RTFE_PACK_EVENTS", "1")
    r.fetch()
    assert r.nbursts and (r.bursts["event_cap"] <= cap_plain).all() and len(r._events) <= int(cap_plain.astype("int64").sum()) * len(cfg.parmsets) * cfg.ntrks
    for key, blob in plain.items():
        assert r.track_events(*key).tobytes() == blob, key
    assert sum(len(v) for v in plain.values()) > 0


@pytest.mark.parametrize("tdelta_ns,ntrks", [(2600, 9), (2300, 9), (2000, 9), (1800, 7), (1600, 9), (1500, 7), (1450, 7), (1200, 9), (1150, 9), (1060, 7), (1000, 9), (800, 9), (640, 9)])
def test_sample_rates_and_the_lean_sift_kernels(tdelta_ns, ntrks, tmp_path, gpu):
    """Other digitisers: 800 BPI NRZI sampled every 0.64 .. 2.6 us gives window widths of 6 .. 27 samples - every instantiation of k_sift_s
    (6 .. 17, nine and seven tracks) and, above, the general kernel: the events are the oracle's and no burst needs the sample path."""
    from readtape_amd import tbin
    spec = synth.TapeSpec(mode=tbin.MODE_NRZI, ntrks=ntrks, bpi=800.0, ips=50.0, tdelta_ns=tdelta_ns, maxvolts=4.4, pulse_w=0.22, seed=tdelta_ns)
    rng = np.random.default_rng(tdelta_ns)
    items = [("block", pl) for pl in synth.random_payloads(rng, 8, 40, 1200, databits=ntrks - 1)]
    tape = synth.make_tape(spec, items, gap_samples=int(3000 * 1280 / tdelta_ns))
    hdr = tape.spec.header()
    opts = ["-ntrks=7"] if ntrks == 7 else []
    att = oracle_attempts(hdr, tape.rows, opts, str(tmp_path))
    fe = frontend.FrontEnd(config_for(hdr, opts))
    res = fe.scan(tape.rows).fetch()
    st = fe.scan_stats(res)
    msgs, stats = check_tape(fe, hdr, tape.rows, att)
    assert not msgs, "\n".join(msgs[:12])
    assert stats["events"] > 5000 and st["redone"] == 0 and st["parallel"] > 0, (stats, st)


DENSE_GPU_CASES = ["pe", "pe_m", "gcr", "gcr_m", "gcr_errs", "gcr_deskew", "pe_order", "gcr_order_m", "nrzi9", "nrzi9_m", "nrzi9_skew", "nrzi9_invert", "nrzi7"]


@pytest.mark.parametrize("name", DENSE_GPU_CASES)
def test_dense_path_against_the_oracle_and_the_reference(name, tmp_path, gpu, monkeypatch):
    """rtfe_dense.hip (k_dseg + k_dchain; opt-in this round) on the real kernels: every event field against the oracle's attempts, and end
    to end the unmodified reference's .tap, transitions and block lines.  NRZI takes it by force (peak path off)."""
    from test_emul_replay import decode_case
    monkeypatch.setenv("RTFE_DENSE_PATH", "1")
    monkeypatch.setenv("RTFE_PEAK_PATH", "0")
    g = load_case(name)
    if "-deskew" not in g["oracle_opts"]:
        att = oracle_attempts(g["hdr"], g["rows"], g["oracle_opts"], str(tmp_path))
        fe = frontend.FrontEnd(config_for(g["hdr"], g["oracle_opts"]))
        msgs, stats = check_tape(fe, g["hdr"], g["rows"], att)
        assert not msgs, "\n".join(msgs[:12])
        assert stats["events"] > 0
    tap, stats = decode_case(g, tmp_path, None)
    assert tap == g["tap"]
    assert stats["agc_mismatches"] == 0 and not stats["event_diffs"], stats


@pytest.mark.parametrize("kind,nparm,knobs", [("gcr", 8, {}), ("gcr", 1, {}), ("pe", 8, {}), ("pe", 1, {}), ("gcr", 8, {"RTFE_DS_WARM": "8"}), ("gcr", 8, {"RTFE_DS_CAP": "5"}),
                                             ("pe", 1, {"RTFE_DS_BAND_LO": "0.9"}), ("gcr", 8, {"RTFE_DS_LEAN": "0"}), ("gcr", 8, {"RTFE_DENSE_DEDUP": "0"})])
def test_dense_path_equals_the_sample_path_at_bench_shape(kind, nparm, knobs, gpu, monkeypatch):
    """One base tape of bench.py's C4 / C3 shapes (5e6 rows, blocks of 512..4096 bytes: chains that cross hundreds of sub-segments): the dense
    path and k_decode give the same burst table, counts and events byte for byte; knobs force failing joins, full lists, thresholds outside
    their bands, the general step for every record."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    torch = gpu
    tape = bench.make_base_tape(seed=1002, target_rows=5e6, kind=kind)
    hdr = tape.spec.header()
    base = list(frontend.DEFAULT_PARMSETS[hdr.mode])
    extra = [(1.4, 0.20, 0.2, 0.5, 0, 0.0), (1.6, 0.14, 0.0, 0.5, 0, 0.0), (1.5, 0.10, 0.1, 0.5, 0, 0.0), (1.3, 0.25, 0.2, 0.5, 0, 0.0)]
    parmsets = ((base + extra)[:nparm] if kind == "gcr" else base[:nparm]) if nparm > 1 else None
    cfg = frontend.FrontEndConfig.from_header(hdr, nparmsets=nparm, parmsets=parmsets)
    rows = torch.from_numpy(tape.rows).cuda()
    for k, v in knobs.items(): monkeypatch.setenv(k, v)
    res = []
    for dp in ("0", "1"):
        monkeypatch.setenv("RTFE_DENSE_PATH", dp)
        fe = frontend.FrontEnd(cfg)
        res.append((fe, fe.scan(rows).fetch()))
    (f0, r0), (f1, r1) = res
    st = f1.scan_stats(r1)
    assert st["redone"] == 0 and (knobs or st["sequential"] > 0.5 * int(r1.counts.sum()) * (1 if nparm == 1 else 0.4)), st      # (no knob: most events came from the lists)
    assert r0.nbursts == r1.nbursts and r0.nbursts > 20
    for k in ("zone_first", "zone_end", "reset_sample", "safe_last", "end_sample", "flags"):
        assert (r0.bursts[k] == r1.bursts[k]).all(), k
    assert (r0.counts == r1.counts).all()
    for b in range(r0.nbursts):
        for p in range(nparm):
            for t in range(cfg.ntrks):
                assert r0.track_events(b, p, t).tobytes() == r1.track_events(b, p, t).tobytes(), (b, p, t)


@pytest.mark.parametrize("name", ["nrzi9", "nrzi9_m", "pe", "gcr_m", "pe_zeros"])
def test_graph_replayed_scans_equal_direct_launches(name, gpu):
    """rtfe_set_graphs: a scan's launches captured into a HIP graph at the first scan of a set of arguments and replayed afterwards - the capture, the
    replays, a second set of buffers, and more sets than the handle keeps graphs for (the least recently used one is captured again) all leave
    the burst table and every event list the direct launches leave."""
    torch = gpu
    g = load_case(name)
    cfg = config_for(g["hdr"], g["oracle_opts"])

    def lists(r):
        r.fetch()
        return r.nbursts, r.bursts.tobytes(), {(b, p, t): r.track_events(b, p, t).tobytes() for b in range(r.nbursts) for p in range(len(cfg.parmsets)) for t in range(cfg.ntrks)}
    fe0 = frontend.FrontEnd(cfg)
    want = lists(fe0.scan(g["rows"]))
    assert want[0] > 0 and sum(len(v) for v in want[2].values()) > 0
    fe = frontend.FrontEnd(cfg)
    fe.set_graphs(True)
    st = torch.cuda.Stream()
    rows = torch.from_numpy(np.ascontiguousarray(g["rows"])).cuda()
    with torch.cuda.stream(st):
        for i in range(3):                                   # capture, replay, replay
            assert lists(fe.scan(rows, stream=st.cuda_stream)) == want, i
        n = rows.shape[0]
        cut = max(64, (n * 3 // 4) // 64 * 64)
        part = lists(fe0.scan(rows[:cut]))
        others = [rows[:cut].clone() for _ in range(9)]      # nine more sets of arguments: more than the handle keeps
        for rep in range(2):
            for o in others:
                assert lists(fe.scan(o, stream=st.cuda_stream)) == part
        assert lists(fe.scan(rows, stream=st.cuda_stream)) == want      # (evicted in between: captured again)
    st.synchronize()
