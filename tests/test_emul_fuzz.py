"""What the shape fuzzer found (tools/fuzz_shapes.py), pinned on the CPU emulator: tapes on which the chains' lean step fired a record although a record further
behind it than its successor - the top between a bottom and the next bottom - fired at the same row, and tops go first (kCrClear looked at the successor alone
up to round 6; k_prep's look ahead, the general step's look back: DESIGN.md 3).  The GPU run of the same tapes and of fresh ones is tests/test_gpu_fuzz.py."""
import pytest

from emul_util import emul_frontend
from fuzz_util import draw, shape_tape
from parity_util import check_tape, config_for, oracle_attempts

# seeds whose tapes the code of round 6's first part got wrong (one wrong event each: a bottom where the reference has the top behind it), mild shapes among them
SEEDS = [6, 14, 79]


@pytest.mark.parametrize("seed", SEEDS)
def test_emulated_lean_step_on_shaped_peaks(seed, tmp_path, monkeypatch, capfd):
    monkeypatch.setenv("RTFE_PREP_CHECK", "1")           # (emulator: what kCrClear promises, checked on the finished streams)
    d = draw(seed)
    tape, rows, nsites, opts = shape_tape(seed, **d)
    hdr = tape.spec.header()
    att = oracle_attempts(hdr, rows, opts, str(tmp_path))
    fe = emul_frontend(config_for(hdr, opts))
    for rep in range(2):                                  # (the second scan runs under the floor the first one learned)
        msgs, stats = check_tape(fe, hdr, rows, att)
        assert not msgs, "\n".join(msgs[:12])
        assert stats["events"] > 0 and nsites > 100
    st = fe.scan_stats(fe.scan(rows).fetch())
    assert st["parallel"] > 2 * st["sequential"]          # the lean step still takes most of the tape
    assert "prep_check:" not in capfd.readouterr().err


@pytest.mark.parametrize("cap", [0, 300])
def test_emulated_work_list_that_runs_full(cap, tmp_path, monkeypatch):
    """k_prep's work list for k_clear with no room (RTFE_WORK_CAP): the records it could not take stay unmarked - more general steps, the same events."""
    seed = 79
    d = draw(seed)
    tape, rows, nsites, opts = shape_tape(seed, **d)
    hdr = tape.spec.header()
    att = oracle_attempts(hdr, rows, opts, str(tmp_path))
    full = emul_frontend(config_for(hdr, opts))
    st_full = full.scan_stats(full.scan(rows).fetch())
    monkeypatch.setenv("RTFE_WORK_CAP", str(cap))
    monkeypatch.setenv("RTFE_PREP_CHECK", "1")
    fe = emul_frontend(config_for(hdr, opts))
    msgs, stats = check_tape(fe, hdr, rows, att)
    assert not msgs, "\n".join(msgs[:12])
    st = fe.scan_stats(fe.scan(rows).fetch())
    assert st["sequential"] > st_full["sequential"] and st["parallel"] + st["sequential"] == st_full["parallel"] + st_full["sequential"]
