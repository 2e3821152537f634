"""Rebuilds librtfe.so with -Rpass-analysis=kernel-resource-usage and prints one line per kernel (registers, spills, occupancy, static LDS).
usage: python tools/kernel_resources.py [substring ...]   (only kernels whose demangled name contains one of the substrings)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from readtape_amd import build as b
    out = os.path.join(b.HERE, "librtfe.so")
    cmd = [b.HIPCC, "-Rpass-analysis=kernel-resource-usage"] + b.HIP_FLAGS + ["-o", out, os.path.join(b.CSRC, "rtfe_api.hip")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stderr[-4000:])
        sys.exit(1)
    want = sys.argv[1:]
    keys = [("SGPR", "TotalSGPRs"), ("VGPR", "VGPRs"), ("AGPR", "AGPRs"), ("scratch", r"ScratchSize \[bytes/lane\]"), ("occ", r"Occupancy \[waves/SIMD\]"),
            ("sspill", "SGPRs Spill"), ("vspill", "VGPRs Spill"), ("LDS", r"LDS Size \[bytes/block\]")]
    for blk in re.split(r"remark: Function Name: ", r.stderr)[1:]:
        name = blk.split()[0]
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dem = re.sub(r"\(.*", "", dem).replace("void ", "").replace("rtfe::", "")
        if want and not any(w in dem for w in want):
            continue
        vals = []
        for label, k in keys:
            m = re.search(k + r": (\d+)", blk)
            vals.append(f"{label} {m.group(1) if m else '?':>4}")
        print(f"{dem:44s} " + "  ".join(vals))


if __name__ == "__main__":
    main()
