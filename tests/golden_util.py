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


def load_files_case(name):
    """tests/golden/files_*.npz: a tape, the reference's options, the output files it wrote and its log lines about them."""
    z = np.load(os.path.join(GOLDEN, f"{name}.npz"))
    ntrks, tdelta, mode, tstart, flags = (int(x) for x in z["hdr"])
    maxvolts, bpi, ips = (float(x) for x in z["hdrf"])
    hdr = tbin.TbinHeader(ntrks=ntrks, tdelta_ns=tdelta, maxvolts=maxvolts, mode=mode, bpi=bpi, ips=ips, tstart_ns=tstart, flags=flags, trkorder=str(z["trkorder"]))
    names = [str(n) for n in z["names"]]
    return dict(hdr=hdr, rows=z["rows"], ref_opts=[str(x) for x in z["ref_opts"]], files={n: z[f"file{i}"].tobytes() for i, n in enumerate(names)},
                report=[str(x) for x in z["report"]])


def report_lines(text):
    """The log lines about output files and the end-of-run summary (the same filter tests/make_goldens.py applies to the reference's log);
    the wall-clock seconds of the "samples were processed" line are normalised to the reference's 0."""
    import re
    keep = []
    for l in text.splitlines():
        if (l.startswith('creating file "') or " was closed at time " in l or l.startswith("summary for file") or " samples were processed in " in l
                or (l.startswith("  created ") and "output file" in l) or l.startswith("  decoded ") or l.startswith("  the last block written")
                or " had errors, " in l or "blocks were unusable" in l or "good blocks had to try" in l or (l.startswith("  parmset ") and "was tried" in l)):
            keep.append(re.sub(r"processed in \d+ seconds \([0-9.]+ seconds/block\)", "processed in 0 seconds (0.000 seconds/block)", l.rstrip()))
    return keep
