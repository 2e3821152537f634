"""Debugging aid: the peak records k_peaks left in a scan's workspace, as {(tile, screen, head, 'main'|'spill'): (records, entries)}."""
import ctypes as C
import numpy as np

DIR = np.dtype([("blob", "<u4"), ("rec_rel", "<u2"), ("nrec", "<u2"), ("ent_rel", "<u2"), ("nent", "<u2"), ("ents8", "<u2"), ("pad", "<u2")])


def dump(fe, res, nrows):
    be = fe.backend
    be.sync()
    out = (C.c_int64 * 8)()
    fe.lib.rtfe_debug_layout.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
    fe.lib.rtfe_debug_layout(fe.h, nrows, out)
    dm_off, ds_off, pool_off, ntiles, nscreens = (int(out[i]) for i in range(5))
    ws = be.to_numpy(res.bufs["ws"], np.uint8)
    if hasattr(ws, "cpu"):
        ws = ws.cpu().numpy()
    ws = np.asarray(ws)
    ntrks = fe.cfg.ntrks
    n = ntiles * nscreens * ntrks
    dm = ws[dm_off: dm_off + n * 16].view(DIR).reshape(ntiles, nscreens, ntrks)
    ds = ws[ds_off: ds_off + n * 16].view(DIR).reshape(ntiles, nscreens, ntrks)
    lists = {}
    for name, d in (("main", dm), ("spill", ds)):
        for t in range(ntiles):
            for s in range(nscreens):
                for h in range(ntrks):
                    e = d[t, s, h]
                    if e["nrec"] >= 0xfffe:
                        lists[(t, s, h, name)] = int(e["nrec"])
                        continue
                    b = pool_off + int(e["blob"]) * 16
                    recs = ws[b + int(e["rec_rel"]) * 8: b + (int(e["rec_rel"]) + int(e["nrec"])) * 8].view("<u4").reshape(-1, 2).copy()
                    eb = b + int(e["ents8"]) * 8 + int(e["ent_rel"]) * 2
                    ents = ws[eb: eb + int(e["nent"]) * 2].view("<u2").copy()
                    lists[(t, s, h, name)] = (recs, ents)
    return lists


def compare(a, b, limit=10):
    msgs = []
    for k in sorted(set(a) | set(b)):
        x, y = a.get(k), b.get(k)
        if x is None or y is None:
            msgs.append(f"{k}: only in one"); continue
        if isinstance(x, int) or isinstance(y, int):
            if not (isinstance(x, int) and isinstance(y, int) and x == y):
                msgs.append(f"{k}: status {x if isinstance(x, int) else 'list'} vs {y if isinstance(y, int) else 'list'}")
            continue
        if x[0].shape != y[0].shape or not np.array_equal(x[0], y[0]) or not np.array_equal(x[1], y[1]):
            n = min(len(x[0]), len(y[0]))
            bad = [i for i in range(n) if not np.array_equal(x[0][i], y[0][i])]
            msgs.append(f"{k}: {len(x[0])} vs {len(y[0])} records, {len(x[1])} vs {len(y[1])} entries, first differing record {bad[:1]}: {x[0][bad[0]] if bad else ''} vs {y[0][bad[0]] if bad else ''}")
        if len(msgs) >= limit:
            break
    return msgs
