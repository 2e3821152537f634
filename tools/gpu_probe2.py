import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]
import torch
from golden_util import load_case
from parity_util import config_for
from readtape_amd import frontend
g = load_case(sys.argv[1] if len(sys.argv) > 1 else "nrzi9")
fe = frontend.FrontEnd(config_for(g["hdr"], g["oracle_opts"]))
t0 = time.time(); r = fe.scan(g["rows"]); torch.cuda.synchronize()
print("scan done in %.3f s" % (time.time() - t0), flush=True)
r.fetch(); print("bursts", r.nbursts, "events", int(r.counts.sum()), fe.scan_stats(r), flush=True)
