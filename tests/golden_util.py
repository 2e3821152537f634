"""Loads the committed golden vectors (tests/golden/, made by tests/make_goldens.py)."""
import glob
import os

import numpy as np

from readtape_amd import tbin

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def case_names():
    return sorted(os.path.basename(p)[5:-4] for p in glob.glob(os.path.join(GOLDEN, "case_*.npz")))


def load_case(name):
    c = np.load(os.path.join(GOLDEN, f"case_{name}.npz"))
    t = np.load(os.path.join(GOLDEN, f"tape_{str(c['tape'])}.npz"))
    ntrks, tdelta, mode, tstart = (int(x) for x in t["hdr"][:4])
    flags = int(t["hdr"][4]) if t["hdr"].size > 4 else 0
    trkorder = str(t["trkorder"]) if "trkorder" in t.files else ""
    maxvolts, bpi, ips = (float(x) for x in t["hdrf"])
    hdr = tbin.TbinHeader(ntrks=ntrks, tdelta_ns=tdelta, maxvolts=maxvolts, mode=mode, bpi=bpi, ips=ips, tstart_ns=tstart, flags=flags, trkorder=trkorder)
    return dict(name=name, hdr=hdr, rows=t["rows"], ref_opts=[str(x) for x in c["ref_opts"]],
                oracle_opts=[str(x) for x in c["oracle_opts"]], tap=c["tap"].tobytes(), events=c["events"],
                returncode=int(c["returncode"]), blocklog=[str(x) for x in c["blocklog"]],
                parms_text=(str(c["parms_text"]) if "parms_text" in c.files else "") or None)
