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
    r = fe.scan(rows); r.fetch()
    ws = r.bufs["ws"].cpu().numpy()
    d2 = ws[136:200].view(np.uint64)
    dbg = ws[64:128].view(np.uint64)
    why = ws[200:264].view(np.uint64)
    print(name, "give-backs: list overflow", int(dbg[6]), "walker reasons [notfast/trust, unknown min, caps, end_ld]:", [int(x) for x in d2[4:8]], "why(par,seq..)", [int(x) for x in why[:2]], "widths", fe.widths)
pe = synth.pe_tape(seed=71, nblocks=40, minlen=500, maxlen=4000, gap_samples=6000)
run("PE", pe, 1e7, nparmsets=1)
gcr = synth.gcr_tape(seed=81, nblocks=20, minlen=1000, maxlen=4000, gap_samples=8000)
run("GCR", gcr, 1e7, nparmsets=1)
