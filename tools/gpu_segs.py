"""GPU box helper: the segmented record walk - parity against the unsegmented walk on the bench tape,
how many bursts were cut, how many failed to join, kernel times."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from readtape_amd import frontend, synth

def scan(rows, hdr, seg):
    if seg is None: os.environ.pop("RTFE_SEG_TILES", None)
    else: os.environ["RTFE_SEG_TILES"] = str(seg)
    fe = frontend.FrontEnd(frontend.FrontEndConfig.from_header(hdr))
    fe.set_timing(True)
    r = fe.scan(rows); r = fe.scan(rows); ms = fe.kernel_ms(); r.fetch()
    ws = r.bufs["ws"].cpu().numpy()
    sc = ws[:64].view(np.int32)
    why = ws[64+8+64+64:64+8+64+64+64].view(np.uint64) if False else None
    return r, ms, {"nsegs": int(sc[6]), "seg_failed": int(sc[8]), "diff_bits": [int(x) for x in ws[200:264].view(np.uint64)], "status_bad": int(ws[136:200].view(np.uint64)[7]), "gain_ulps_1_4_64_more": [int(x) for x in ws[136:168].view(np.uint64)]}

def same(a, b):
    if a.nbursts != b.nbursts or not (a.counts == b.counts).all(): return "counts differ"
    if not (a.bursts["flags"] == b.bursts["flags"]).all(): return "flags differ"
    for i in range(a.nbursts):
        ea, eb = a.events_of(i), b.events_of(i) if hasattr(a, "events_of") else (None, None)
    return None

nblocks = int(sys.argv[1]) if len(sys.argv) > 1 else 400
maxlen = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
minlen = int(sys.argv[3]) if len(sys.argv) > 3 else 512
tape = synth.nrzi_tape(seed=5, nblocks=nblocks, minlen=minlen, maxlen=maxlen, marks_every=16, gap_samples=6000)
hdr = tape.spec.header()
rows = torch.from_numpy(tape.rows).cuda()
ref, ms0, st0 = scan(rows, hdr, 0)
for seg, warm in ((48, 8),):
    os.environ['RTFE_SEG_WARMUP'] = str(warm)
    r, ms, st = scan(rows, hdr, seg)
    ok = r.nbursts == ref.nbursts and (r.counts == ref.counts).all() and (r.bursts["flags"] == ref.bursts["flags"]).all()
    if ok:
        for b in range(r.nbursts):
            B = r.bursts[b]; cap = int(B["event_cap"]); base = int(B["event_base"])
            for t in range(hdr.ntrks):
                n = int(r.counts[b, 0, t])
                x = r._events[base + t * cap: base + t * cap + n]; y = ref._events[base + t * cap: base + t * cap + n]
                if x.tobytes() != y.tobytes(): ok = False; break
            if not ok: break
    print(json.dumps({"seg_tiles": seg, "warm": warm, "identical_to_unsegmented": bool(ok), **st, "rows": int(rows.shape[0]), "bursts": int(r.nbursts),
                      "k_walk_ms": round(ms["k_walk"], 3), "unsegmented_k_walk_ms": round(ms0["k_walk"], 3), "resume_ms": round(ms["k_decode_resume"], 3)}))
