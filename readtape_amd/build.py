"""Builds the native pieces in-tree (no pip, no JIT cache):

  readtape_amd/librtfe.so       HIP front end (gfx950) + C ABI            hipcc
  readtape_amd/librtdecode.so   host block decoders / driver (C99)        gcc
  oracle/_build/oracle_readtape the CPU parity oracle (test infra)        make -C oracle

hipcc cross-compiles for gfx950 without a GPU."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
             "-ffp-contract=off",            # bit-exact parity: the reference is C99 on SSE2, no FMA
             "-fno-fast-math", "-Wall", "-Wno-unused-function",
             f"-I{os.path.join(ROOT, 'include')}", f"-I{CSRC}"]
HOST_SRCS = ["rt_decode_common.c", "rt_decode_nrzi.c", "rt_decode_pe.c", "rt_decode_gcr.c", "rt_decode_ww.c", "rt_parmsets.c", "rt_driver.c", "rt_replay.c", "rt_csv.c"]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_frontend(force=False, verbose=False):
    out = os.path.join(HERE, "librtfe.so")
    deps = [os.path.join(CSRC, f) for f in ("rtfe_api.hip", "rtfe_kernels.hip", "rtfe_zeros.hip", "rtfe_ww.hip", "rtfe_sift.hip", "rtfe_gain.hip", "rtfe_dense.hip", "rtfe_pack.hip", "rtfe_pk.h", "rtfe_device.h")] + [os.path.join(ROOT, "include", "rt_frontend.h")]
    if force or _newer(out, deps):
        cmd = [HIPCC] + HIP_FLAGS + os.environ.get("RTFE_EXTRA_HIPFLAGS", "").split() + ["-o", out, os.path.join(CSRC, "rtfe_api.hip")]
        if verbose:
            cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        subprocess.run(cmd, check=True)
    return out


def build_host(force=False):
    out = os.path.join(HERE, "librtdecode.so")
    srcs = [os.path.join(CSRC, "host", f) for f in HOST_SRCS]
    if force or _newer(out, srcs + [os.path.join(CSRC, "host", "rt_decode.h"), os.path.join(CSRC, "host", "rt_replay.h")]):
        subprocess.run(["gcc", "-std=gnu99", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-D_DEFAULT_SOURCE",
                        "-Wall", f"-I{os.path.join(CSRC, 'host')}", f"-I{os.path.join(ROOT, 'include')}", "-o", out] + srcs + ["-lm", "-lpthread"], check=True)
    return out


def build_oracle():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle"], check=True)
    if os.path.isdir("/root/reference/src"):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"], check=True)


def build_all(force=False, verbose=False):
    build_frontend(force, verbose)
    build_host(force)
    build_oracle()


if __name__ == "__main__":
    build_all(force="--force" in sys.argv, verbose="-v" in sys.argv)
    print("built")
