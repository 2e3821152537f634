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
    r = fe.scan(rows); ms = fe.kernel_ms(); r.fetch()
    ws = r.bufs["ws"].cpu().numpy()
    dbg = ws[64:128].view(np.uint64)
    nt = max(int(dbg[3]), 1)
    print(name, "ms", {k2: round(v, 2) for k2, v in ms.items() if v > 0.01}, "tiles", int(dbg[3]), "cyc/tile load", int(dbg[0] / nt), "screen", int(dbg[1] / nt), "walk", int(dbg[5] / nt), "final", int(dbg[6] / nt), "events/tile", int(r.counts.sum()) / nt, "A cyc/tile", int(ws[128+8:128+8+64].view(np.uint64)[4])/nt, "B cyc/tile", int(ws[136:200].view(np.uint64)[5])/nt, "hits trk0/tile", int(ws[136:200].view(np.uint64)[6])/nt, "eval cyc/tile", int(ws[136:200].view(np.uint64)[7])/nt, "evals trk0/tile", int(ws[136:200].view(np.uint64)[3])/nt)
run("GCR", synth.gcr_tape(seed=81, nblocks=20, minlen=1000, maxlen=4000, gap_samples=8000), 2e7, nparmsets=1)
run("PE", synth.pe_tape(seed=71, nblocks=40, minlen=500, maxlen=4000, gap_samples=6000), 2e7, nparmsets=1)
run("PEz", synth.pe_tape(seed=71, nblocks=40, minlen=500, maxlen=4000, gap_samples=6000), 2e7, nparmsets=1, find_zeros=True)
