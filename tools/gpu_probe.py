"""GPU debugging aid: k_peaks' records of one golden tape (only k_peaks / k_bursts launched), saved for comparison with the
emulated kernels' (tools/gpu_probe_cmp.py)."""
import os, sys, pickle
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]
os.environ["RTFE_PEAK_STOP"] = "3"
import torch
from golden_util import load_case
from parity_util import config_for
from readtape_amd import frontend
import pk_dump
name = sys.argv[1] if len(sys.argv) > 1 else "nrzi9"
g = load_case(name)
cfg = config_for(g["hdr"], g["oracle_opts"])
fe = frontend.FrontEnd(cfg)
r = fe.scan(g["rows"])
torch.cuda.synchronize()
lists = pk_dump.dump(fe, r, g["rows"].shape[0])
pickle.dump(lists, open(os.path.join(ROOT, "gpurun_out", f"pk_{name}.pkl"), "wb"))
print("dumped", len(lists), "lists", flush=True)
