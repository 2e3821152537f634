"""The C-ABI shared libraries load on a GPU-less machine and export every symbol the headers declare
(no compute calls here).  Also checks the product refuses to run without its native library / a GPU."""
import ctypes
import os
import re
import sys

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
    assert lib.rtfe_abi_version() == 6
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


def build_abi_client(out_dir):
    """gcc (C, not C++) on tests/abi_client.c: the header as a C translation unit, linked against librtfe.so and the HIP runtime."""
    import subprocess
    exe = os.path.join(str(out_dir), "abi_client")
    subprocess.run(["gcc", "-std=gnu99", "-Wall", "-Werror=implicit-function-declaration", "-D__HIP_PLATFORM_AMD__", f"-I{ROOT}/include", "-I/opt/rocm/include",
                    "-o", exe, os.path.join(ROOT, "tests", "abi_client.c"), f"-L{ROOT}/readtape_amd", "-lrtfe", "-L/opt/rocm/lib", "-lamdhip64",
                    f"-Wl,-rpath,{ROOT}/readtape_amd", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return exe


def test_the_header_compiles_as_c_and_links(built, tmp_path):
    """The INTEGRATION.md stub as a real C program: include/rt_frontend.h is C, every entry point it calls resolves in librtfe.so."""
    exe = build_abi_client(tmp_path)
    assert os.path.exists(exe)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["nrzi9", "pe_zeros", "gcr_m", "nrzi9_skew"])
def test_a_c_client_gets_what_the_python_binding_gets(name, built, tmp_path):
    """tests/abi_client.c fills rtfe_config field by field from a text file, scans a golden tape through the C ABI and dumps bursts, counts and
    events as the HEADER lays them out; the same tape through frontend.py's ctypes mirrors must give the same bytes."""
    import subprocess
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
    from golden_util import load_case
    from parity_util import config_for
    from readtape_amd import frontend
    g = load_case(name)
    cfg = config_for(g["hdr"], g["oracle_opts"])
    exe = build_abi_client(tmp_path)
    lines = [f"mode {cfg.mode}", f"ntrks {cfg.ntrks}", f"maxvolts {cfg.maxvolts!r}", f"bpi {cfg.bpi!r}", f"ips {cfg.ips!r}", f"tdelta_ns {cfg.tdelta_ns}",
             f"tstart_ns {cfg.tstart_ns}", f"invert {int(cfg.invert)}", f"differentiate {int(cfg.differentiate)}", f"find_zeros {int(cfg.find_zeros)}"]
    for i, t in enumerate(cfg.head_to_trk or []): lines.append(f"head {i} {t}")
    for i, n in enumerate(cfg.skew or []): lines.append(f"skew {i} {n}")
    for p in cfg.parmsets: lines.append("parmset " + " ".join(repr(float(x)) if j != 4 else str(int(x)) for j, x in enumerate(p)))
    (tmp_path / "c.txt").write_text("\n".join(lines) + "\n")
    rows = np.ascontiguousarray(g["rows"], dtype=np.int16)
    rows.tofile(str(tmp_path / "rows.bin"))
    p = subprocess.run([exe, str(tmp_path / "c.txt"), str(tmp_path / "rows.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr + p.stdout
    fe = frontend.FrontEnd(cfg)
    r = fe.scan(rows).fetch()
    raw = open(tmp_path / "out.bin", "rb").read()
    nb = int(np.frombuffer(raw[:4], np.int32)[0])
    assert nb == r.nbursts and nb > 0
    off = 4
    bursts = np.frombuffer(raw[off: off + nb * frontend.BURST_DTYPE.itemsize], frontend.BURST_DTYPE); off += nb * frontend.BURST_DTYPE.itemsize
    P, T = len(cfg.parmsets), cfg.ntrks
    counts = np.frombuffer(raw[off: off + nb * P * T * 4], np.uint32).reshape(nb, P, T); off += nb * P * T * 4
    for k in ("zone_first", "zone_end", "reset_sample", "safe_last", "end_sample", "flags", "event_cap"):
        assert (bursts[k] == r.bursts[k][:nb]).all(), k
    assert (counts == r.counts[:nb]).all()
    it = frontend.EVENT_DTYPE.itemsize
    for b in range(nb):
        for pi in range(P):
            for t in range(T):
                n = int(counts[b, pi, t])
                assert raw[off: off + n * it] == r.track_events(b, pi, t).tobytes(), (b, pi, t)
                off += n * it
    assert off == len(raw) and int(counts.sum()) > 100
