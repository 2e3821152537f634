"""The shape fuzzer's tapes (tests/fuzz_util.py) on the MI355X through the C ABI, every event against the oracle: the seeds that found the successor-only kCrClear
of rounds 3 - 6, and a spread of fresh ones over every format the peak and dense paths take."""
import pytest

from fuzz_util import draw, shape_tape
from parity_util import check_tape, config_for, oracle_attempts
from readtape_amd import frontend

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [6, 14, 23, 42, 47, 49, 57, 61, 63, 79] + list(range(900, 930)) + list(range(100000, 100016)))      # (from 100000 on: shapes over a block's first peaks too)
def test_shaped_peaks(seed, tmp_path):
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    d = draw(seed)
    tape, rows, nsites, opts = shape_tape(seed, **d)
    hdr = tape.spec.header()
    att = oracle_attempts(hdr, rows, opts, str(tmp_path))
    fe = frontend.FrontEnd(config_for(hdr, opts))
    for rep in range(2):
        msgs, stats = check_tape(fe, hdr, rows, att)
        assert not msgs, "\n".join(msgs[:12])
        assert stats["events"] > 0


def test_work_list_that_runs_full(tmp_path, monkeypatch):
    """k_prep's work list for k_clear with no room (RTFE_WORK_CAP): what it could not take stays unmarked - more general steps, the same events."""
    seed = 79
    d = draw(seed)
    tape, rows, nsites, opts = shape_tape(seed, **d)
    hdr = tape.spec.header()
    att = oracle_attempts(hdr, rows, opts, str(tmp_path))
    full = frontend.FrontEnd(config_for(hdr, opts))
    st_full = full.scan_stats(full.scan(rows).fetch())
    monkeypatch.setenv("RTFE_WORK_CAP", "300")
    fe = frontend.FrontEnd(config_for(hdr, opts))
    msgs, stats = check_tape(fe, hdr, rows, att)
    assert not msgs, "\n".join(msgs[:12])
    st = fe.scan_stats(fe.scan(rows).fetch())
    assert st["sequential"] > st_full["sequential"] and st["parallel"] + st["sequential"] == st_full["parallel"] + st_full["sequential"]


@pytest.mark.parametrize("seed,window,halo", [(14, 1 << 13, 1 << 11), (79, 1 << 12, 1 << 10), (6, 1 << 13, 1 << 10)])
def test_shaped_tape_in_streamed_windows(seed, window, halo, tmp_path):
    """The shapes across fragment boundaries: a shaped tape through device windows far shorter than its blocks (the halo has to grow) writes the .tap of the
    whole-tape decode - whose events test_shaped_peaks holds against the oracle."""
    from readtape_amd import ingest, pipeline, tbin
    d = draw(seed)
    tape, rows, nsites, opts = shape_tape(seed, **d)
    hdr = tape.spec.header()
    pipeline.decode_tape(hdr, rows, str(tmp_path / "whole.tap"))
    want = open(tmp_path / "whole.tap", "rb").read()
    path = str(tmp_path / "t.tbin")
    tbin.write_tbin(path, hdr, rows)
    st = ingest.decode_file_streaming(path, str(tmp_path / "s.tap"), window_rows=window, halo_rows=halo, replay_threads=4, replay_split=3)
    assert open(tmp_path / "s.tap", "rb").read() == want
    assert st["rows"] == rows.shape[0] and st["windows"] >= 3
