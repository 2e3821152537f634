"""GPU box helper: throughput of the other BASELINE configurations' shapes (not the bench line): C3 (PE 1600 BPI, 640 ns,
-zeros), PE peak path, C4 (GCR 9042 BPI, 160 ns, 8-parameter-set sweep), NRZI default 8-set sweep.  Data resident in HBM,
per-kernel HIP-event times of one scan after a warm-up scan."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from readtape_amd import frontend, synth

def run(name, base, target_rows, **cfgkw):
    hdr = base.spec.header()
    k = max(1, int(target_rows // base.rows.shape[0]))
    rows = torch.from_numpy(base.rows).cuda().repeat(k, 1).contiguous()
    fe = frontend.FrontEnd(frontend.FrontEndConfig.from_header(hdr, **cfgkw))
    fe.set_timing(True)
    for _ in range(2):
        r = fe.scan(rows)
        ms = fe.kernel_ms()[0]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        r = fe.scan(rows)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    r.fetch()
    print(json.dumps({"config": name, "rows": int(rows.shape[0]), "parmsets": len(fe.cfg.parmsets), "ms_per_scan": round(dt * 1e3, 3),
                      "Msamples_per_s": round(rows.shape[0] / dt / 1e6, 1), "events": int(r.counts.sum()), "bursts": int(r.nbursts),
                      "flagged": int((r.bursts["flags"] & ~np.uint32(frontend.F_EXACT_START | frontend.F_STATE_AT_END)).astype(bool).sum()),
                      "kernel_ms": {k2: round(v, 3) for k2, v in ms.items()}}))
    del rows, fe

pe = synth.pe_tape(seed=71, nblocks=40, minlen=500, maxlen=4000, gap_samples=6000)
run("C3 shape: PE 1600 BPI 640 ns -zeros", pe, 5e7, nparmsets=1, find_zeros=True)
run("PE 1600 BPI 640 ns peak path, 1 set", pe, 5e7, nparmsets=1)
gcr = synth.gcr_tape(seed=81, nblocks=20, minlen=1000, maxlen=4000, gap_samples=8000)
sets = [(bf, rise, mp, al, 0, 0.0) for bf in (1.2, 1.5) for rise, mp in ((0.14, 0.0), (0.2, 0.2)) for al in (0.3, 0.5)]
run("C4 shape: GCR 9042 BPI 160 ns, 8-set sweep", gcr, 5e7, parmsets=sets)
run("GCR 9042 BPI 160 ns, 1 set", gcr, 5e7, nparmsets=1)
nrzi = synth.nrzi_tape(seed=91, nblocks=40, minlen=500, maxlen=4000, gap_samples=5000)
run("NRZI 800 BPI 1280 ns, default 8-set sweep (-m)", nrzi, 5e7, nparmsets=8)
