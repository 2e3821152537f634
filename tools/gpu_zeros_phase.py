"""GPU box: where a tile of k_zeros spends its cycles (RTFE_DEBUG=1 phase counters of zeros_tile_parallel)."""
import os, sys
os.environ["RTFE_DEBUG"] = "1"
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from readtape_amd import frontend, synth
base = synth.pe_tape(seed=71, nblocks=40, minlen=500, maxlen=4000, gap_samples=6000)
hdr = base.spec.header()
k = max(1, int(1e8 // base.rows.shape[0]))
rows = torch.from_numpy(base.rows).cuda().repeat(k, 1).contiguous()
fe = frontend.FrontEnd(frontend.FrontEndConfig.from_header(hdr, nparmsets=1, find_zeros=True))
fe.set_timing(True)
for _ in range(2):
    r = fe.scan(rows); ms = fe.kernel_ms(); r.fetch()
ws = r.bufs["ws"].cpu().numpy()
d2 = ws[136:200].view(np.uint64)
nt = max(int(d2[6]), 1)
print("rows", rows.shape[0], "ms", {k2: round(v, 2) for k2, v in ms.items() if v > 0.01}, "tiles", nt, "track-tiles", int(d2[4]), "parallel ok", int(d2[5]),
      "cycles/tile (pass, verify+repair, events+store, tail barrier)", [int(d2[i] / nt) for i in range(4)])
