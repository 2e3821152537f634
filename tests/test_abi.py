"""The C-ABI shared libraries load on a GPU-less machine and export every symbol the headers declare
(no compute calls here).  Also checks the product refuses to run without its native library / a GPU."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header, prefix):
    text = open(header).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(" + prefix + r"\w+)\s*\(", text)))


@pytest.fixture(scope="module")
def built():
    from readtape_amd import build
    build.build_frontend()
    build.build_host()


def test_frontend_library_exports_every_declared_symbol(built):
    names = declared_functions(os.path.join(ROOT, "include", "rt_frontend.h"), "rtfe_")
    assert len(names) >= 12
    lib = ctypes.CDLL(os.path.join(ROOT, "readtape_amd", "librtfe.so"))
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert lib.rtfe_abi_version() == 2
    assert lib.rtfe_kernel_count() == 12


def test_host_decode_library_exports_every_declared_symbol(built):
    names = declared_functions(os.path.join(ROOT, "readtape_amd", "csrc", "host", "rt_decode.h"), "rt_")
    names = [n for n in names if n not in ("rt_reader",)]
    lib = ctypes.CDLL(os.path.join(ROOT, "readtape_amd", "librtdecode.so"))
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_event_and_burst_layouts():
    from readtape_amd import frontend
    assert frontend.EVENT_DTYPE.itemsize == 16
    assert frontend.BURST_DTYPE.itemsize == ctypes.sizeof(frontend._Burst) == 56


def test_product_fails_loudly_without_gpu_or_library(tmp_path):
    import torch
    from readtape_amd import frontend, synth
    cfg = frontend.FrontEndConfig.from_header(synth.nrzi_spec().header())
    from emul_util import NumpyBackend
    with pytest.raises(RuntimeError, match="missing"):
        frontend.FrontEnd(cfg, _lib_path=str(tmp_path / "nope.so"), _backend=NumpyBackend())
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no CPU path"):
            frontend.FrontEnd(cfg)


def test_config_validation_mirrors_reference_asserts():
    """rtfe_create rejects what the reference asserts on (src/decoder.c:502,516; src/decoder.h:97)."""
    from emul_util import emul_frontend
    from readtape_amd import frontend, synth
    hdr = synth.nrzi_spec().header()
    bad = [dict(parmsets=[(0.7, 0.2, 1.0, 0.3, 2, 0.0)]),        # agc_window and agc_alpha both set
           dict(parmsets=[(0.7, 0.2, 1.0, 0.0, 11, 0.0)]),       # agc_window > AGC_MAX_WINDOW
           dict(skew=[0, 0, 0, 0, 51, 0, 0, 0, 0]),              # > MAXSKEWSAMP
           dict(head_to_trk=[0, 0, 1, 2, 3, 4, 5, 6, 7])]        # not a permutation
    for kw in bad:
        with pytest.raises(ValueError):
            emul_frontend(frontend.FrontEndConfig.from_header(hdr, **kw))
    fe = emul_frontend(frontend.FrontEndConfig.from_header(hdr, nparmsets=8))
    assert fe.widths == [13, 11, 13, 11, 17, 13, 13, 11]          # W for 0.7 / 0.6 / 0.9 at C2
