"""GPU box helper: how the tiles of an NRZI 8-set sweep fare in k_walk (parallel tile path vs sequential walk, give-backs)."""
import os, sys, json
os.environ["RTFE_DEBUG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from readtape_amd import frontend, synth
tape = synth.nrzi_tape(seed=91, nblocks=40, minlen=500, maxlen=4000, gap_samples=5000)
hdr = tape.spec.header()
rows = torch.from_numpy(tape.rows).cuda().repeat(4, 1).contiguous()
for n in (1, 4, 8):
    fe = frontend.FrontEnd(frontend.FrontEndConfig.from_header(hdr, nparmsets=n)); fe.set_timing(True)
    r = fe.scan(rows); ms = fe.kernel_ms(); r.fetch()
    ws = r.bufs["ws"].cpu().numpy()
    why = ws[200:264].view(np.uint64); dbg = ws[64:128].view(np.uint64); d2 = ws[136:200].view(np.uint64)
    print(json.dumps({"parmsets": n, "k_walk_ms": round(ms["k_walk"], 2), "tiles_parallel": int(why[0]), "tiles_sequential": int(why[1]), "why2..7": [int(x) for x in why[2:8]],
                      "resume_ms": round(ms["k_decode_resume"], 2)}))
