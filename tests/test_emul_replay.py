"""End-to-end on the CPU (kernel sources under tests/cpu_emul): front end -> event replay -> block decoders
-> SIMH .tap, byte-compared with the UNMODIFIED reference's .tap stored in the golden vectors."""
import os

import pytest

from emul_util import emul_frontend
from golden_util import load_case
from readtape_amd import frontend, pipeline

CASES = ["nrzi9", "nrzi9_m", "nrzi9_correct", "nrzi7", "nrzi9_skew", "nrzi9_invert", "nrzi9_sub2", "pe", "pe_m", "nrzi9_zeros", "pe_zeros", "gcr", "gcr_m", "gcr_zeros", "gcr_errs", "gcr_correct", "nrzi9_deskew", "nrzi9_deskew_long", "nrzi7_deskew_restart", "gcr_deskew", "nrzi9_nobpi", "nrzi9_nobpi_short", "nrzi9_diffz", "pe_diffz", "gcr_diffz", "nrzi9_diffpk", "nrzi9_diffpk_clean", "nrzi9_diffpk_skew", "gcr_diffpk", "pe_diffpk", "nrzi9_cut", "nrzi9_cut_zeros", "noise_only", "tiny", "nrzi9_nobpi_deskew", "nrzi7_order", "pe_order", "gcr_order_m", "nrzi7_order_ignored"]
EMUL_CASES = ["nrzi9", "nrzi9_m", "nrzi9_skew", "nrzi9_sub2", "pe", "nrzi9_zeros", "pe_zeros", "gcr", "gcr_zeros", "gcr_correct", "nrzi9_deskew", "nrzi7_deskew_restart", "nrzi9_nobpi_short", "nrzi9_diffz", "pe_diffz", "gcr_diffz", "nrzi9_diffpk", "nrzi9_cut", "nrzi9_cut_zeros", "noise_only", "tiny", "nrzi7_order", "pe_order", "nrzi7_order_ignored"]     # the thread emulation is slow: a subset here, all on the GPU


def decode_case(g, tmp_path, fe_factory):
    o = g["oracle_opts"]
    skew = next(([int(x) for x in a[6:].split(",")] for a in o if a.startswith("-skew=")), None)
    opts = pipeline.DecodeOptions(multiple_tries="-m" in o, correct="-correct" in o)
    subsample = next((int(a[11:]) for a in o if a.startswith("-subsample=")), 1)
    trkorder = next((a[7:] for a in o if a.startswith("-order=")), None)
    tap = os.path.join(str(tmp_path), "out.tap")
    stats, res = pipeline.decode_tape(g["hdr"], g["rows"], tap, log_path=tap + ".log", opts=opts, fe_factory=fe_factory,
                                      skew=skew, invert="-invert" in o, find_zeros="-zeros" in o, evt_path=tap + ".evt", differentiate="-differentiate" in o,
                                      subsample=subsample, deskew="-deskew" in o, trkorder=trkorder, parms_text=g.get("parms_text"))
    import refdump
    # every transition the decoders were handed == what the reference's front end handed its decoders
    # (differentiated zero-crossing: the decoder-side baseline bookkeeping v_avg_height depends on the opposite
    #  excursion at each callback, which the event does not carry; it is inert with -zeros, src/decoder.c:501)
    ign = ("v_avg_height",) if ("-zeros" in o and "-differentiate" in o) else ()
    stats["event_diffs"] = refdump.compare(refdump.load(tap + ".evt"), g["events"], ignore_fields=ign)
    if g["returncode"] == 0:   # the reference's per-block result lines: error / parity / ECC / corrected-bit counts, AGC range, offsets
        mine = [l.strip() for l in open(tap + ".log").read().splitlines() if l.startswith("wrote block") or "tapemark at" in l or "observed flux transitions" in l or "density was set to" in l or "average peak height is" in l]
        assert mine == list(g["blocklog"]), (mine, list(g["blocklog"]))
    return open(tap, "rb").read(), stats


@pytest.mark.parametrize("name", EMUL_CASES)
def test_tap_bytes_match_reference(name, tmp_path):
    g = load_case(name)
    tap, stats = decode_case(g, tmp_path, emul_frontend)
    print(name, stats)
    assert tap == g["tap"], f"{name}: .tap differs from the reference's ({len(tap)} vs {len(g['tap'])} bytes)"
    assert stats["agc_mismatches"] == 0
    assert stats["events_delivered"] > 0 or g["events"].size <= 1          # (noise only / a few rows: nothing to deliver)
    assert not stats["event_diffs"], stats["event_diffs"]


@pytest.mark.parametrize("name", ["pe", "pe_m", "gcr", "gcr_m", "gcr_correct", "gcr_deskew", "pe_order", "gcr_order_m", "nrzi9", "nrzi9_m", "nrzi9_skew"])
def test_tap_bytes_match_reference_on_the_dense_path(name, tmp_path, monkeypatch):
    """rtfe_dense.hip (k_dseg + k_dchain; opt-in) end to end: the reference's .tap, its transitions and its block lines.  The NRZI cases take
    it by force (the peak path off)."""
    monkeypatch.setenv("RTFE_DENSE_PATH", "1")
    monkeypatch.setenv("RTFE_PEAK_PATH", "0")
    g = load_case(name)
    tap, stats = decode_case(g, tmp_path, emul_frontend)
    assert tap == g["tap"], f"{name}: .tap differs from the reference's ({len(tap)} vs {len(g['tap'])} bytes)"
    assert stats["agc_mismatches"] == 0 and not stats["event_diffs"], stats


def test_reference_agc_assert_stops_the_decode_where_the_reference_stops(tmp_path):
    """src/decoder.c:782 ("AGC gain bad in lookfor_peak") kills the reference after 104 transitions of this tape (found by
    tests/stress_gpu.py).  The device marks the row in the track's event list, the replay delivers everything in front of it -
    the very transitions the reference's callbacks saw - and the pipeline raises where the reference exits."""
    import refdump
    g = load_case("nrzi7_agcfatal")
    assert g["returncode"] == 99
    tap = os.path.join(str(tmp_path), "out.tap")
    with pytest.raises(pipeline.ReferenceFatal) as ei:
        pipeline.decode_tape(g["hdr"], g["rows"], tap, fe_factory=emul_frontend, invert=True, differentiate=True, evt_path=tap + ".evt",
                             parms_text=g["parms_text"])
    mine = refdump.load(tap + ".evt")
    n = min(mine.size, g["events"].size)
    assert mine.size == g["events"].size and n > 50, (mine.size, g["events"].size)
    assert not refdump.compare(mine[:n], g["events"][:n])
    assert ei.value.stats["reference_fatal"]


@pytest.mark.parametrize("peak_path", ["0", "1"])
def test_agc_assert_inside_a_parameter_sweep_ends_everything(peak_path, tmp_path, monkeypatch):
    """-m: the second set's gain goes negative in an attempt other sets have been through (stress seed 704 tape 59: the GPU hung,
    the emulator crashed - the screened walk narrowed its "blind for ever" row to an int).  The reference exits at the assert: no
    further set is tried, and the transitions delivered up to there are the reference's."""
    import refdump
    monkeypatch.setenv("RTFE_PEAK_PATH", peak_path)
    g = load_case("nrzi9_agcfatal_m")
    assert g["returncode"] == 99
    tap = os.path.join(str(tmp_path), "out.tap")
    with pytest.raises(pipeline.ReferenceFatal):
        pipeline.decode_tape(g["hdr"], g["rows"], tap, fe_factory=emul_frontend, evt_path=tap + ".evt", parms_text=g["parms_text"],
                             opts=pipeline.DecodeOptions(multiple_tries=True, even_parity=True))
    mine = refdump.load(tap + ".evt")
    assert mine.size == g["events"].size and mine.size > 10000, (mine.size, g["events"].size)
    assert not refdump.compare(mine, g["events"])


@pytest.mark.parametrize("peak_path", ["0", "1"])
def test_a_learned_peak_height_that_is_not_positive_ends_everything(peak_path, tmp_path, monkeypatch):
    """src/decode_nrzi.c:227 ("avg peak-to-peak voltage isn't positive"): noise wiggles taken for peaks make a track's learned average
    height negative, and the reference exits INSIDE the block decoder's callback (stress seed 901 tape 85: the replay used to go on,
    43 transitions further, to the next assert).  The transitions delivered up to there are the reference's, the last one included."""
    import refdump
    monkeypatch.setenv("RTFE_PEAK_PATH", peak_path)
    g = load_case("nrzi9_avgheight_fatal")
    assert g["returncode"] == 99
    tap = os.path.join(str(tmp_path), "out.tap")
    with pytest.raises(pipeline.ReferenceFatal) as ei:
        pipeline.decode_tape(g["hdr"], g["rows"], tap, fe_factory=emul_frontend, evt_path=tap + ".evt", parms_text=g["parms_text"],
                             opts=pipeline.DecodeOptions(multiple_tries=True, even_parity=True))
    mine = refdump.load(tap + ".evt")
    assert mine.size == g["events"].size and mine.size > 10000, (mine.size, g["events"].size)
    assert not refdump.compare(mine, g["events"])
    assert ei.value.stats["reference_fatal"]


def test_short_burst_tails_do_not_change_the_tap(tmp_path, monkeypatch):
    """A burst's walkers stop tail_rows into the next quiet zone.  Even with an absurdly short tail
    the .tap must not change: a zone only starts a whole quiet KiB after the last flux transition, by when the block
    decoders have ended the block - and if one had not, the replay would ask for an exact rescan."""
    monkeypatch.setenv("RTFE_TAIL_ROWS", "8")
    g = load_case("nrzi9")
    tap, stats = decode_case(g, tmp_path, emul_frontend)
    assert tap == g["tap"]
    assert not stats["event_diffs"]


def _noisy_pe_zeros_case(tmp_path, fe_factory):
    """-zeros on a PE tape whose gap noise (50 mV rms) now and then crosses the 0.2 V threshold: such an excursion is
    detector history without any event, so an attempt must not be continued into the next device burst as if it were
    fresh (RTFE_F_STATE_AT_END).  Found by tests/stress_gpu.py."""
    import subprocess
    import refdump
    from parity_util import ORACLE, build_oracle
    from readtape_amd import synth, tbin
    build_oracle()
    tape = synth.pe_tape(seed=833875458, nblocks=2, minlen=30, maxlen=200, gap_samples=3000, amplitude=2.5, noise_mv=50.0, jitter=0.08)
    hdr = tape.spec.header()
    wd = str(tmp_path)
    tbin.write_tbin(os.path.join(wd, "t.tbin"), hdr, tape.rows)
    subprocess.run([ORACLE, "-v", f"-out={wd}/o", f"-evt={wd}/o.evt", "-zeros", os.path.join(wd, "t.tbin")], check=True)
    st, res = pipeline.decode_tape(hdr, tape.rows, os.path.join(wd, "g.tap"), evt_path=os.path.join(wd, "g.evt"), find_zeros=True, fe_factory=fe_factory)
    assert not refdump.compare(refdump.load(os.path.join(wd, "g.evt")), refdump.load(os.path.join(wd, "o.evt")))
    assert open(os.path.join(wd, "g.tap"), "rb").read() == open(os.path.join(wd, "o.tap"), "rb").read()
    assert any(int(f) & 64 for f in res.bursts["flags"])


def test_zeros_excursions_without_events_are_history(tmp_path):
    _noisy_pe_zeros_case(tmp_path, emul_frontend)


def _event_flood_case(tmp_path, fe_factory):
    """-zeros -differentiate on a tape whose noise (80 mV rms) is above the differentiator's dead band: a confirmed crossing
    on almost every other row, three times what the event regions are sized for.  The reference plods through them; the
    pipeline must too (exact rescans with worst-case event regions), not stop quietly.  Found by tests/stress_gpu.py."""
    import subprocess
    import refdump
    from parity_util import ORACLE, build_oracle
    from readtape_amd import synth, tbin
    build_oracle()
    tape = synth.nrzi_tape(seed=615258906, nblocks=2, minlen=16, maxlen=200, ntrks=7, gap_samples=1500, amplitude=1.8, noise_mv=80.0, jitter=0.0)
    hdr = tape.spec.header()
    wd = str(tmp_path)
    tbin.write_tbin(os.path.join(wd, "t.tbin"), hdr, tape.rows)
    subprocess.run([ORACLE, "-v", f"-out={wd}/o", f"-evt={wd}/o.evt", "-ntrks=7", "-zeros", "-differentiate", os.path.join(wd, "t.tbin")], check=True)
    st, res = pipeline.decode_tape(hdr, tape.rows, os.path.join(wd, "g.tap"), evt_path=os.path.join(wd, "g.evt"), find_zeros=True, differentiate=True,
                                   fe_factory=fe_factory)
    a, b = refdump.load(os.path.join(wd, "g.evt")), refdump.load(os.path.join(wd, "o.evt"))
    assert not refdump.compare(a, b, ignore_fields=("v_avg_height",))
    assert open(os.path.join(wd, "g.tap"), "rb").read() == open(os.path.join(wd, "o.tap"), "rb").read()
    assert st["events_delivered"] > 0.2 * tape.rows.shape[0] * 7 and st["exact_scans"] > 0


def test_event_floods_are_decoded_not_dropped(tmp_path):
    _event_flood_case(tmp_path, emul_frontend)


@pytest.mark.parametrize("name,nshards", [("nrzi9", 2), ("nrzi9", 3), ("gcr", 2), ("pe", 2)])
def test_fragments_concatenate_to_the_whole_tap(name, nshards, tmp_path):
    """Time shards / streamed windows: every fragment scans its rows + a halo, replays the bursts it owns and writes its piece of
    the .tap; the pieces concatenated (+ the end-of-medium marker) are the reference's .tap.  The cuts fall wherever nrows / nshards
    puts them on the 64-row grid: inside blocks and inside gaps."""
    import numpy as np
    from readtape_amd import shard
    g = load_case(name)
    if g["oracle_opts"] and name != "nrzi9_deskew_long":
        pytest.skip("options")
    rows = g["rows"]
    tap = os.path.join(str(tmp_path), "frag.tap")
    spans = shard.plan_shards(rows.shape[0], nshards, align=64)
    sts = pipeline.decode_tape_fragments(g["hdr"], rows, tap, spans, fe_factory=emul_frontend, halo_rows=2048)
    got = open(tap, "rb").read()
    if name == "nrzi9_deskew_long":          # (the golden was made with -deskew; without it the skewed tape decodes differently: compare with the unsharded decode)
        whole = os.path.join(str(tmp_path), "whole.tap")
        pipeline.decode_tape(g["hdr"], rows, whole, fe_factory=emul_frontend)
        assert got == open(whole, "rb").read()
    else:
        assert got == g["tap"], (len(got), len(g["tap"]))
    assert sum(s["blocks"] + s["tapemarks"] for s in sts) > 0


@pytest.mark.parametrize("name,nsub", [("nrzi9", 2), ("nrzi9", 4), ("gcr", 2), ("pe", 3)])
def test_sub_fragments_on_native_threads_concatenate_to_the_whole_tap(name, nsub, tmp_path):
    """rt_replay_run_fragments (what the streaming reader hands a window to): ONE scan's bursts replayed as nsub sub-fragments side by side, a native thread each,
    cut at the zone starts of evenly spaced bursts - the pieces are the reference's .tap, and each piece is what the one-fragment call writes."""
    g = load_case(name)
    if g["oracle_opts"]:
        pytest.skip("options")
    hdr, rows = g["hdr"], g["rows"]
    opts = pipeline.DecodeOptions()
    full = pipeline.default_parmsets(hdr.mode, 1)
    cfg = frontend.FrontEndConfig.from_header(hdr, parmsets=pipeline.frontend_parmsets(full))
    fe = emul_frontend(cfg)
    res, nb, bound = pipeline.scan_fragment(fe, rows, rows.shape[0], 0, True, True)()
    assert res.nbursts >= nsub
    cuts = [int(res.bursts[(res.nbursts * j) // nsub]["zone_first"]) for j in range(1, nsub)]
    starts, stops = [0] + cuts, cuts + [None]
    paths = [os.path.join(str(tmp_path), f"p{j}.tap") for j in range(nsub)]
    sts = pipeline.decode_fragments(hdr, cfg, fe, res, rows, 0, starts, stops, paths, full, opts)
    pieces = [open(p, "rb").read() for p in paths]
    assert b"".join(pieces) + b"\xff\xff\xff\xff" == g["tap"]
    assert sum(st["blocks"] + st["tapemarks"] for st, _ in sts) > 0 and all(secs >= 0 for _, secs in sts)
    for j in range(nsub):
        one = os.path.join(str(tmp_path), f"one{j}.tap")
        pipeline.decode_fragment(hdr, cfg, fe, res, rows, 0, starts[j], stops[j], one, full, opts)
        assert open(one, "rb").read() == pieces[j]


def test_native_positional_read_in_pieces(tmp_path):
    """rt_read_mt: a window's bytes as pieces read side by side - any offset, any length, more threads than megabytes; a read past the file's end fails."""
    import ctypes as C
    import numpy as np
    lib = pipeline._load_decode_lib()
    data = np.random.default_rng(5).integers(0, 256, size=(40 << 20) + 12345, dtype=np.uint8)
    path = os.path.join(str(tmp_path), "f.bin")
    data.tofile(path)
    fd = os.open(path, os.O_RDONLY)
    try:
        for off, n, th in ((0, data.size, 8), (777, (33 << 20) + 5, 16), (12345, 1 << 20, 4), (5, 100, 64), (data.size - 10, 10, 3), (0, 0, 4)):
            buf = np.zeros(n + 8, np.uint8)
            assert lib.rt_read_mt(fd, buf.ctypes.data, off, n, th) == 0
            assert np.array_equal(buf[:n], data[off: off + n]) and not buf[n:].any()
        buf = np.zeros(16 << 20, np.uint8)
        assert lib.rt_read_mt(fd, buf.ctypes.data, data.size - (9 << 20), 16 << 20, 4) != 0
    finally:
        os.close(fd)


@pytest.mark.parametrize("name", ["files_nrzi9_bin", "files_nrzi9_tap", "files_nrzi7_bin", "files_pe_m_tap", "files_gcr_bin"])
def test_output_files_and_summary_match_the_reference(name, tmp_path, monkeypatch):
    """f3: the numbered .bin files (a new one behind every tapemark, src/readtape.c:1091-1111), the lazily created .tap, and the log
    lines about them + the end-of-run summary incl. the "samples were processed" count (src/readtape.c:2021-2044)."""
    from golden_util import load_files_case, report_lines
    g = load_files_case(name)
    o = g["ref_opts"]
    monkeypatch.chdir(tmp_path)
    opts = pipeline.DecodeOptions(multiple_tries="-m" in o, verbose="-v" in o)
    pipeline.decode_tape(g["hdr"], g["rows"], None, log_path="t.log", opts=opts, fe_factory=emul_frontend, out_base="t", in_name="t.tbin", tap_format="-tap" in o)
    made = sorted(f for f in os.listdir(".") if f.endswith(".bin") or f.endswith(".tap"))
    assert made == sorted(g["files"])
    for f in made:
        assert open(f, "rb").read() == g["files"][f], f
    assert report_lines(open("t.log").read()) == g["report"]


@pytest.mark.parametrize("name", ["nrzi9_zeros", "pe_zeros", "gcr_zeros", "nrzi9_cut_zeros"])
@pytest.mark.parametrize("knobs", [{}, {"RTFE_ZEROS_KERNEL": "0"}, {"RTFE_ZC_WARM": "16"}, {"RTFE_ZC_PARALLEL": "0"}])
def test_zeros_with_the_gpu_default_tile(name, knobs, tmp_path, monkeypatch):
    """-zeros through k_zeros (default: two tracks per lane, 128-row sub-segments), through k_decode's zero-crossing mode with the GPU's
    896-row tiles, with a warm-up too short to converge (the joins fail and are repaired) and as k_decode's sequential walk."""
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    g = load_case(name)
    tap, stats = decode_case(g, tmp_path, lambda cfg: emul_frontend(cfg, tile_rows=896))
    assert tap == g["tap"]
    assert not stats["event_diffs"], stats["event_diffs"]


@pytest.mark.parametrize("ntrks,clip,order", [(9, True, None), (7, False, None), (8, True, None), (6, False, [5, 3, 1, 0, 2, 4]), (2, True, [1, 0]), (12, False, None), (19, True, None)])
def test_zeros_kernel_against_the_decode_mode(ntrks, clip, order, monkeypatch):
    """k_zeros for odd and even track counts (its three instantiations: 9, 7, any), a permuted column order, samples at the ends of the int16
    range and exact zeros at sign changes, short and long warm-ups: the burst table and every event as k_decode's zero-crossing mode and
    its sequential walk make them."""
    import zeros_util
    hdr, rows = zeros_util.zeros_rows(ntrks, nblocks=4, clip=clip)
    variants = [{}, {"RTFE_ZC_WARM": "8"}, {"RTFE_ZC_WARM": "64"}, {"RTFE_ZEROS_KERNEL": "0"}, {"RTFE_ZC_PARALLEL": "0"}]
    out = zeros_util.scan_variants(emul_frontend, hdr, rows, monkeypatch, variants, head_to_trk=order)
    assert out[0].nbursts >= 4 and int(out[0].counts.sum()) > 5000
    for r in out[1:]:
        zeros_util.same_scan(out[0], r, ntrks)


def test_rows_of_the_wrong_width_are_refused():
    """The C ABI takes a pointer and a row count: the host wrapper is where a [n, 18] array for a 19-track configuration must stop."""
    import zeros_util
    hdr, rows = zeros_util.zeros_rows(9, nblocks=1)
    fe = emul_frontend(frontend.FrontEndConfig.from_header(hdr, find_zeros=True))
    with pytest.raises(ValueError):
        fe.scan(rows[:, :8])
    with pytest.raises(ValueError):
        fe.scan(rows.reshape(-1))


WW_CASES = ["ww", "ww_auto", "ww_unused", "ww_pos", "ww_pos_auto", "ww_wrongdir", "ww_reverse", "ww_rough", "ww_close", "ww_deskew", "ww_deskew_long", "ww_deskew_pos"]


def decode_ww_case(g, tmp_path, fe_factory, chunk_rows):
    import refdump
    o = g["oracle_opts"]
    tap = os.path.join(str(tmp_path), "out.tap")
    stats = pipeline.decode_tape_ww(g["hdr"], g["rows"], tap, log_path=tap + ".log", evt_path=tap + ".evt", fe_factory=fe_factory, chunk_rows=chunk_rows,
                                    fluxdir=next((a[9:] for a in o if a.startswith("-fluxdir=")), "neg"), reverse="-reverse" in o, deskew="-deskew" in o)
    stats["event_diffs"] = refdump.compare(refdump.load(tap + ".evt"), g["events"])
    mine = [l.strip() for l in open(tap + ".log").read().splitlines() if l.startswith("wrote block") or "tapemark at" in l or "observed flux transitions" in l or "average peak height is" in l]
    assert mine == list(g["blocklog"]), (mine, list(g["blocklog"]))
    return open(tap, "rb").read(), stats


@pytest.mark.parametrize("chunk_rows", [4096, 300])
@pytest.mark.parametrize("name", WW_CASES)
def test_whirlwind_tap_bytes_match_reference(name, chunk_rows, tmp_path):
    """Whirlwind: the device detector's state is handed from attempt to attempt (rtfe_ww_scan), incl. the re-seeding of the tracks at
    every attempt start; the events the decoder is handed, the block lines and the .tap are the reference's - both polarities,
    -fluxdir=auto, the wrong polarity, -reverse, a rough tape (missing-bit / missing-clock warnings), blocks and block marks a few bit
    times apart - whether an attempt's rows come in one chunk or in many."""
    g = load_case(name)
    tap, stats = decode_ww_case(g, tmp_path, emul_frontend, chunk_rows)
    assert tap == g["tap"]
    assert stats["agc_mismatches"] == 0 and stats["events_delivered"] > 0
    assert not stats["event_diffs"], stats["event_diffs"]


def _tape_with_an_eventless_burst():
    """A 9-track NRZI tape whose second gap holds a slow 1 V trapezoid on one track: rows that are not quiet (a burst boundary on
    either side of it) but that the detector sees no peak in - a burst without events."""
    import numpy as np
    from readtape_amd import synth
    tape = synth.nrzi_tape(seed=77, nblocks=4, minlen=60, maxlen=120, gap_samples=9000)
    rows = tape.rows.copy()
    (_, _, e1, _), (_, s2, _, _) = tape.blocks[1], tape.blocks[2]
    mid = (e1 + s2) // 2
    ramp = np.concatenate([np.linspace(0, 1.0, 600), np.full(300, 1.0), np.linspace(1.0, 0, 600)])
    lo = mid - ramp.size // 2
    rows[lo: lo + ramp.size, 4] += np.round(ramp / tape.spec.maxvolts * 32767).astype(np.int16)
    return tape.spec.header(), rows, lo, lo + ramp.size


@pytest.mark.parametrize("where", ["behind", "inside_gap_before", "far_behind"])
def test_a_cut_behind_an_eventless_burst_does_not_duplicate_a_block(where, tmp_path):
    """The replay of a fragment must not chain from a burst without events into a burst the NEXT fragment owns (rt_replay.c: the
    chaining branch checks stop_row): [(0, n)] and [(0, cut), (cut, n)] write the same .tap for cuts around the eventless burst."""
    hdr, rows, lo, hi = _tape_with_an_eventless_burst()
    n = rows.shape[0]
    whole = os.path.join(str(tmp_path), "whole.tap")
    sts = pipeline.decode_tape_fragments(hdr, rows, whole, [(0, n)], fe_factory=emul_frontend)
    ref = open(whole, "rb").read()
    assert sum(s["blocks"] for s in sts) == 4
    cut = {"behind": (hi + 640) // 64 * 64, "inside_gap_before": (lo - 1200) // 64 * 64, "far_behind": (hi + 2600) // 64 * 64}[where]
    tap = os.path.join(str(tmp_path), "cut.tap")
    sts = pipeline.decode_tape_fragments(hdr, rows, tap, [(0, cut), (cut, n)], fe_factory=emul_frontend, halo_rows=1 << 14)
    assert open(tap, "rb").read() == ref, (where, [s["blocks"] for s in sts])
    # three fragments, the middle one holding nothing but the eventless burst
    a, b = (lo - 1200) // 64 * 64, (hi + 640) // 64 * 64
    sts = pipeline.decode_tape_fragments(hdr, rows, tap, [(0, a), (a, b), (b, n)], fe_factory=emul_frontend, halo_rows=1 << 14)
    assert open(tap, "rb").read() == ref, [s["blocks"] for s in sts]
