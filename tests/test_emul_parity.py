"""Kernel-logic parity on the CPU: the HIP kernel sources compiled against tests/cpu_emul and run
on threads, checked event-for-event against the oracle on the golden tapes.  (The real GPU run of
the same checks is tests/test_gpu_parity.py, marked gpu.)"""
import pytest

from emul_util import emul_frontend
from golden_util import load_case
from parity_util import check_tape, config_for, oracle_attempts

PEAK_CASES = ["nrzi9", "nrzi9_m", "nrzi7", "nrzi9_skew", "nrzi9_invert", "pe", "pe_m", "gcr", "gcr_m"]


@pytest.mark.parametrize("name", PEAK_CASES)
def test_emulated_kernels_match_oracle(name, tmp_path):
    g = load_case(name)
    att = oracle_attempts(g["hdr"], g["rows"], g["oracle_opts"], str(tmp_path))
    fe = emul_frontend(config_for(g["hdr"], g["oracle_opts"]))
    msgs, stats = check_tape(fe, g["hdr"], g["rows"], att)
    print(name, stats)
    assert not msgs, "\n".join(msgs[:12])
    assert stats["events"] > 0
