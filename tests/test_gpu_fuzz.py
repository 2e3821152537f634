"""The shape fuzzer's tapes (tests/fuzz_util.py) on the MI355X through the C ABI, every event against the oracle: the seeds that found the successor-only kCrClear
of rounds 3 - 6, and a spread of fresh ones over every format the peak and dense paths take."""
import pytest

from fuzz_util import draw, shape_tape
from parity_util import check_tape, config_for, oracle_attempts
from readtape_amd import frontend

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [6, 14, 23, 42, 47, 49, 57, 61, 63, 79] + list(range(900, 930)))
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
