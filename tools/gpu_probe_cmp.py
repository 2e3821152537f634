"""Compares gpurun_out/pk_<case>.pkl (tools/gpu_probe.py on a GPU) with the emulated kernels' records."""
import os, sys, pickle
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]
os.environ["RTFE_PEAK_STOP"] = "3"
from emul_util import emul_frontend
from golden_util import load_case
from parity_util import config_for
import pk_dump
name = sys.argv[1] if len(sys.argv) > 1 else "nrzi9"
g = load_case(name)
fe = emul_frontend(config_for(g["hdr"], g["oracle_opts"]))
r = fe.scan(g["rows"])
mine = pk_dump.dump(fe, r, g["rows"].shape[0])
gpu = pickle.load(open(os.path.join(ROOT, "gpurun_out", f"pk_{name}.pkl"), "rb"))
msgs = pk_dump.compare(gpu, mine)
print(name, len(mine), "lists;", "identical" if not msgs else "\n".join(msgs))
