"""Reader for the 48-byte event-dump records written by oracle/ref_event_shim.c (reference build)
and oracle/oracle_main.c (-evt=).  TEST INFRASTRUCTURE."""
import numpy as np

REC = np.dtype([("kind", "<u4"), ("trk", "<u4"), ("peakcount", "<i4"), ("parmset", "<i4"),
                ("t_peak", "<f8"), ("timenow_ns", "<i8"),
                ("v_peak", "<f4"), ("agc_gain", "<f4"), ("v_avg_height", "<f4"), ("pad", "<u4")])
assert REC.itemsize == 48


def load(path):
    return np.fromfile(path, dtype=REC)


def compare(a, b, ignore_attempt_time=True, ignore_fields=()):
    """Returns a list of human-readable differences (empty = identical)."""
    out = []
    if a.size != b.size:
        out.append(f"record count {a.size} != {b.size}")
    n = min(a.size, b.size)
    a = a[:n].copy(); b = b[:n].copy()
    if ignore_attempt_time:
        # the reference's `timenow` before the very first sample is (float)tstart/1e9; it is a log
        # value only (src/readtape.c:1376) and not part of the front end's contract
        a["t_peak"][a["kind"] == 2] = 0
        b["t_peak"][b["kind"] == 2] = 0
    for name in REC.names:
        if name in ignore_fields:
            continue
        va, vb = a[name], b[name]
        if va.dtype.kind == "f":
            neq = va.view(f"u{va.dtype.itemsize}") != vb.view(f"u{vb.dtype.itemsize}")
        else:
            neq = va != vb
        idx = np.flatnonzero(neq)
        if idx.size:
            i = idx[0]
            out.append(f"field {name}: {idx.size} differ, first at rec {i}: {va[i]!r} vs {vb[i]!r} (kind {a['kind'][i]} trk {a['trk'][i]})")
    return out


if __name__ == "__main__":
    import sys
    a, b = load(sys.argv[1]), load(sys.argv[2])
    d = compare(a, b)
    print(f"{a.size} vs {b.size} records;", "IDENTICAL" if not d else "\n".join(d))
    sys.exit(1 if d else 0)
