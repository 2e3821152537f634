"""GPU: the streaming reader (readtape_amd/ingest.py) — a .tbin file decoded through device windows much smaller than the tape
writes the .tap the whole-tape decode writes (which the replay tests pin to the reference's)."""
import numpy as np
import pytest

from readtape_amd import ingest, pipeline, synth, tbin

pytestmark = pytest.mark.gpu


def _whole(hdr, rows, path):
    pipeline.decode_tape(hdr, rows, path)
    return open(path, "rb").read()


@pytest.mark.parametrize("threads", [1, 4])
@pytest.mark.parametrize("kind,window,halo", [("nrzi", 1 << 16, 1 << 12), ("nrzi", 1 << 14, 1 << 10), ("pe", 1 << 15, 1 << 13), ("gcr", 1 << 15, 1 << 13)])
def test_streamed_windows_write_the_whole_tape_tap(kind, window, halo, threads, tmp_path):
    """Windows far shorter than the tape, halos shorter than a block (so the halo has to grow): same bytes."""
    if kind == "nrzi":
        tape = synth.nrzi_tape(seed=71, nblocks=40, minlen=200, maxlen=1500, marks_every=7, gap_samples=3000)
    elif kind == "pe":
        tape = synth.pe_tape(seed=72, nblocks=12, minlen=200, maxlen=900, gap_samples=3000)
    else:
        tape = synth.gcr_tape(seed=73, nblocks=8, minlen=300, maxlen=1200, gap_samples=3000)
    hdr = tape.spec.header()
    want = _whole(hdr, tape.rows, str(tmp_path / "whole.tap"))
    path = str(tmp_path / "t.tbin")
    tbin.write_tbin(path, hdr, tape.rows)
    st = ingest.decode_file_streaming(path, str(tmp_path / "s.tap"), window_rows=window, halo_rows=halo, replay_threads=threads,
                                      replay_split=3 if threads > 1 else 1)      # (with threads: every window's bursts as three sub-fragments side by side)
    got = open(tmp_path / "s.tap", "rb").read()
    assert got == want
    assert st["rows"] == tape.rows.shape[0] and st["windows"] >= 3 and st["blocks"] > 0


def test_end_marker_inside_the_payload_ends_the_tape(tmp_path):
    """src/readtape.c:1410: the data end at the first row whose head-0 sample is 0x8000, wherever it is."""
    tape = synth.nrzi_tape(seed=74, nblocks=20, minlen=200, maxlen=900, gap_samples=3000)
    hdr = tape.spec.header()
    cut = tape.rows.shape[0] * 5 // 8
    want = _whole(hdr, tape.rows[:cut], str(tmp_path / "whole.tap"))
    path = str(tmp_path / "t.tbin")
    rows = tape.rows.copy()
    with open(path, "wb") as f:
        f.write(tbin.pack_header(hdr))
        rows[cut, 0] = tbin.END_MARK
        f.write(rows.tobytes())
    st = ingest.decode_file_streaming(path, str(tmp_path / "s.tap"), window_rows=1 << 15, halo_rows=1 << 12)
    assert st["rows"] == cut
    assert open(tmp_path / "s.tap", "rb").read() == want


@pytest.mark.parametrize("threads", [1, 4])
@pytest.mark.parametrize("behind", [700, 2000, 4090])
def test_end_marker_inside_a_windows_halo(behind, threads, tmp_path):
    """a marker `behind` rows behind a window boundary lies in the halo of the window in front of it.  That window keeps its own rows only
    (its bounding burst, its stop row); the window behind it decodes what is left in front of the marker - no block twice, none missing."""
    tape = synth.nrzi_tape(seed=77, nblocks=40, minlen=200, maxlen=900, marks_every=9, gap_samples=1500)
    hdr = tape.spec.header()
    win, halo = 1 << 15, 1 << 12
    assert tape.rows.shape[0] > 4 * win
    for k in (2, 3):
        cut = k * win + behind
        want = _whole(hdr, tape.rows[:cut], str(tmp_path / "whole.tap"))
        rows = tape.rows.copy()
        rows[cut, 0] = tbin.END_MARK
        path = str(tmp_path / "t.tbin")
        with open(path, "wb") as f:
            f.write(tbin.pack_header(hdr))
            f.write(rows.tobytes())
        st = ingest.decode_file_streaming(path, str(tmp_path / "s.tap"), window_rows=win, halo_rows=halo, replay_threads=threads, replay_split=2 if threads > 1 else 1)
        assert st["rows"] == cut
        assert open(tmp_path / "s.tap", "rb").read() == want


@pytest.mark.parametrize("threads", [1, 6])
def test_windows_read_in_parallel_pieces_with_an_end_marker_in_one_of_them(threads, tmp_path):
    """Windows of 9.4 MB are read as four pieces side by side, each piece looks for the end marker in its own rows; two scans are in
    flight while a third window is read.  Same bytes as the whole-tape decode of the rows in front of the marker."""
    tape = synth.nrzi_tape(seed=75, nblocks=40, minlen=200, maxlen=1500, marks_every=9, gap_samples=3000)
    hdr = tape.spec.header()
    rows = np.tile(tape.rows, (4, 1))
    win = 1 << 19
    cut = 3 * win + win * 5 // 8 + 12345                  # inside the third piece of the fourth window
    assert cut < rows.shape[0] and win * 2 * hdr.ntrks >= (8 << 20)
    want = _whole(hdr, rows[:cut], str(tmp_path / "whole.tap"))
    rows = rows.copy()
    rows[cut, 0] = tbin.END_MARK
    path = str(tmp_path / "t.tbin")
    with open(path, "wb") as f:
        f.write(tbin.pack_header(hdr))
        f.write(rows.tobytes())
    st = ingest.decode_file_streaming(path, str(tmp_path / "s.tap"), window_rows=win, halo_rows=1 << 15, replay_threads=threads, read_threads=4)
    assert st["rows"] == cut and st["windows"] >= 4
    assert open(tmp_path / "s.tap", "rb").read() == want


def test_a_noisy_tape_raises_the_screen_floor(tmp_path):
    """60 mV rms of noise on 2 - 3 V peaks: with a candidate screen built for a learned peak height of 1 V the lists outgrow their slots and the bursts are
    redone on the samples.  A scan context's first window estimates the floor from its samples (k_scan_begin), the windows behind it screen against half the
    smallest peak height a chain learned (and, should a quarter of the first windows' bursts be redone all the same, the reader makes contexts with a
    calibrated floor).  Same bytes as the whole-tape decode (whose events the oracle pins), and the floor was above the default."""
    tape = synth.nrzi_tape(seed=76, nblocks=60, minlen=400, maxlen=1500, marks_every=9, gap_samples=3000, noise_mv=60.0)
    hdr = tape.spec.header()
    want = _whole(hdr, tape.rows, str(tmp_path / "whole.tap"))
    path = str(tmp_path / "t.tbin")
    tbin.write_tbin(path, hdr, tape.rows)
    st = ingest.decode_file_streaming(path, str(tmp_path / "s.tap"), window_rows=1 << 17, halo_rows=1 << 14, replay_threads=4, replay_split=2)
    assert open(tmp_path / "s.tap", "rb").read() == want
    assert st["windows"] >= 4 and st["blocks"] > 50
    assert st["screen_floor_height"] is not None and 1.0 < st["screen_floor_height"] <= 4.0, st
