import os, sys, json
os.environ["RTFE_DEBUG"] = "1"
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from readtape_amd import frontend, synth
def run(name, base, target, **kw):
    hdr = base.spec.header()
    k = max(1, int(target // base.rows.shape[0]))
    rows = torch.from_numpy(base.rows).cuda().repeat(k, 1).contiguous()
    fe = frontend.FrontEnd(frontend.FrontEndConfig.from_header(hdr, **kw))
    fe.set_timing(True)
    r = fe.scan(rows); ms = fe.kernel_ms()[0]; r.fetch()
    ws = r.bufs["ws"].cpu().numpy()
    dbg = ws[64:128].view(np.uint64)
    nt = max(int(dbg[3]), 1)
    d2 = ws[136:200].view(np.uint64)
    print(name, "ms", {k2: round(v, 2) for k2, v in ms.items() if v > 0.01}, "tiles", int(dbg[3]), "cyc/tile load", int(dbg[0] / nt), "screen", int(dbg[1] / nt), "walk", int(dbg[5] / nt), "final", int(dbg[6] / nt), "zc tracks", int(d2[4]), "zc ok", int(d2[5]), "zc phases (pass, verify+repair, events+store, tail barrier)", [int(d2[i] / nt) for i in range(4)])
run("PEz", synth.pe_tape(seed=71, nblocks=40, minlen=500, maxlen=4000, gap_samples=6000), 2e7, nparmsets=1, find_zeros=True)
if len(sys.argv) > 1 and sys.argv[1] == "gcr":
    gcr = synth.gcr_tape(seed=81, nblocks=20, minlen=1000, maxlen=4000, gap_samples=8000)
    run("GCR 1 set", gcr, 2e7, nparmsets=1)
    sets = [(bf, rise, mp, al, 0, 0.0) for bf in (1.2, 1.5) for rise, mp in ((0.14, 0.0), (0.2, 0.2)) for al in (0.3, 0.5)]
    run("GCR 8 sets", gcr, 2e7, parmsets=sets)
    run("PE peak 1 set", synth.pe_tape(seed=71, nblocks=40, minlen=500, maxlen=4000, gap_samples=6000), 2e7, nparmsets=1)
