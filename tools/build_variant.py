"""A build of the front end with other compile-time knobs, beside the product's: readtape_amd/librtfe_<name>.so (RTFE_LIB_PATH selects it; tools only).
usage: python tools/build_variant.py <name> [hipcc flags, e.g. -DRTFE_SIFT_PROF]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from readtape_amd import build as b

name, flags = sys.argv[1], sys.argv[2:]
out = os.path.join(b.HERE, f"librtfe_{name}.so")
subprocess.run([b.HIPCC] + b.HIP_FLAGS + flags + ["-o", out, os.path.join(b.CSRC, "rtfe_api.hip")], check=True)
print(out)
