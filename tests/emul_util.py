"""Runs the UNMODIFIED kernel sources on the CPU through tests/cpu_emul (threads + barriers) so
that kernel logic can be checked in the GPU-less container.  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

from readtape_amd import frontend

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMUL_DIR = os.path.join(ROOT, "tests", "cpu_emul")
EMUL_SO = os.path.join(EMUL_DIR, "librtfe_emul.so")


def build_emul():
    srcs = [os.path.join(ROOT, "readtape_amd", "csrc", f) for f in ("rtfe_api.hip", "rtfe_kernels.hip", "rtfe_zeros.hip", "rtfe_ww.hip", "rtfe_sift.hip", "rtfe_gain.hip", "rtfe_dense.hip", "rtfe_pack.hip", "rtfe_pk.h", "rtfe_device.h")]
    srcs += [os.path.join(ROOT, "include", "rt_frontend.h"), os.path.join(EMUL_DIR, "hip", "hip_runtime.h"), os.path.join(EMUL_DIR, "emul_main.cpp")]
    stale = lambda: not os.path.exists(EMUL_SO) or any(os.path.getmtime(s) > os.path.getmtime(EMUL_SO) for s in srcs)
    if stale():
        import fcntl
        with open(EMUL_SO + ".lock", "w") as lk:           # (pytest-xdist workers: one builds, the others wait and find it done; the library appears whole)
            fcntl.flock(lk, fcntl.LOCK_EX)
            if stale():
                tmp = EMUL_SO + f".{os.getpid()}.tmp"
                subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fPIC", "-shared", "-ffp-contract=off", "-pthread", "-x", "c++",
                                f"-I{EMUL_DIR}", f"-I{ROOT}/include", f"-I{ROOT}/readtape_amd/csrc", "-o", tmp,
                                os.path.join(EMUL_DIR, "emul_main.cpp")], check=True)
                os.replace(tmp, EMUL_SO)
    return EMUL_SO


class NumpyBackend:
    def empty(self, nbytes):
        return np.zeros(max(int(nbytes), 16) + 64, dtype=np.uint8)

    def ptr(self, a):
        p = a.ctypes.data
        return p

    def rows(self, rows):
        a = np.ascontiguousarray(rows, dtype=np.int16)
        # the C ABI wants 16-byte alignment
        buf = np.zeros(a.size + 16, dtype=np.int16)
        off = (-buf.ctypes.data % 16) // 2
        v = buf[off: off + a.size].reshape(a.shape)
        v[...] = a
        return v

    def to_numpy(self, a, dtype, count=None):
        n = (a.size // np.dtype(dtype).itemsize) * np.dtype(dtype).itemsize
        v = a[:n].view(dtype)
        return v if count is None else v[:count]

    def upload(self, a, host_bytes):
        a[: len(host_bytes)] = np.frombuffer(host_bytes, dtype=np.uint8)

    def stream(self):
        return None

    def sync(self):
        pass


def emul_frontend(cfg, tile_rows=512):
    # big tiles = few barrier rendezvous: the thread emulation pays ~0.1 ms per __syncthreads
    old = os.environ.get("RTFE_TILE_ROWS")
    os.environ["RTFE_TILE_ROWS"] = str(tile_rows)
    try:
        return frontend.FrontEnd(cfg, _lib_path=build_emul(), _backend=NumpyBackend())
    finally:
        if old is None:
            os.environ.pop("RTFE_TILE_ROWS", None)
        else:
            os.environ["RTFE_TILE_ROWS"] = old
