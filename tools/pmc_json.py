"""Turns the rocprofv3 --pmc passes of tools/gpu_traffic.sh into profiles/pmc_<config>.json: HBM bytes per scan for every timed span
of rtfe_scan (the names rtfe_kernel_name() reports), counters corrected by factors calibrated in the same session on kernels of known
traffic (tools/pmc_calib.py).  bench.py reports the dominant span's fetch + write bytes as roofline.traffic.
usage: pmc_json.py <dir with calib_fetch/ calib_write/ pmc_fetch/ pmc_write/> <config> <rows> <round tag> > pmc_<config>.json"""
import glob, json, os, sqlite3, sys

root, config, rows, tag = sys.argv[1], sys.argv[2], int(float(sys.argv[3])), sys.argv[4]
CAL_BYTES = 1 << 30

# kernel -> the span of rtfe_scan that times it (readtape_amd/csrc/rtfe_api.hip KNAMES)
SPANS = [("k_scan_begin", "k_sift"), ("k_reset_floor", "k_sift"), ("k_adapt_floor", "k_decode"), ("k_dseg", "k_dseg"), ("k_dorder", "k_dchain"), ("k_dchain", "k_dchain"), ("k_sift_hard", "k_prep"), ("k_clear", "k_prep"), ("k_sift", "k_sift"), ("k_qpack", "k_bursts"), ("k_pscan", "k_prep"), ("k_prep", "k_prep"), ("k_bursts", "k_bursts"),
         ("k_zones", "k_bursts"), ("k_segplan", "k_gain_s"), ("k_gain_seg", "k_gain_s"), ("k_gain_join", "k_gain_s"), ("k_gain", "k_gain"), ("k_emit_seg", "k_emit"), ("k_emit", "k_emit"), ("k_publish", "k_emit"), ("k_decode", "k_decode"),
         ("k_zeros", "k_zeros"), ("k_quiet", "k_quiet")]


def span_of(kernel):
    base = kernel.split("(")[0].split("<")[0].split("::")[-1].strip()
    for prefix, span in SPANS:
        if base.startswith(prefix):
            return base, span
    return base, None


def counters(sub):
    out = {}
    for f in sorted(glob.glob(os.path.join(root, sub, "**", "*.db"), recursive=True)):
        db = sqlite3.connect(f)
        for kn, cn, n, tot in db.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name"):
            out[(kn, cn)] = (n, tot)
    return out


def calib(sub, counter, needle, known):
    best = None
    for (kn, cn), (n, tot) in counters(sub).items():
        if cn == counter and needle.lower() in kn.lower() and n >= 1:
            best = known / (tot / n * 1024.0)
    return best


fetch_f = calib("calib_fetch", "FETCH_SIZE", "bitwise_xor", CAL_BYTES) or calib("calib_fetch", "FETCH_SIZE", "BitwiseXor", CAL_BYTES)
write_f = calib("calib_write", "WRITE_SIZE", "FillFunctor", CAL_BYTES)
write_f2 = calib("calib_write", "WRITE_SIZE", "bitwise_xor", CAL_BYTES) or calib("calib_write", "WRITE_SIZE", "BitwiseXor", CAL_BYTES)
doc = {"round": tag, "config": config, "workload_rows": rows,
       "source": "tools/gpu_traffic.sh (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, no trace domains)",
       "calibration": {"bytes": CAL_BYTES, "fetch_factor": fetch_f, "write_factor_fill": write_f, "write_factor_xor": write_f2,
                       "note": "factor = known bytes / (counter x 1024); 16 B per lane coalesced streams (tools/pmc_calib.py), 1 GiB per dispatch"}}
ff = fetch_f or 2.0
wf = write_f or 1.0
per_kernel = {}
nscans = 0
for sub, counter, factor, key in (("pmc_fetch", "FETCH_SIZE", ff, "fetch_bytes"), ("pmc_write", "WRITE_SIZE", wf, "write_bytes")):
    c = counters(sub)
    # a scan launches exactly one of k_bursts (short quiet maps) and k_bursts_tail (a workgroup per round, round 4)
    scans = sum(n for (kn, cn), (n, tot) in c.items() if cn == counter and span_of(kn)[0] in ("k_bursts", "k_bursts_tail")) or 1
    nscans = scans
    for (kn, cn), (n, tot) in c.items():
        base, span = span_of(kn)
        if cn != counter or span is None or "rtfe" not in kn:
            continue
        e = per_kernel.setdefault(base, {"span": span, "dispatches_per_scan": n / scans, "fetch_bytes": 0, "write_bytes": 0})
        e[key] += int(tot * 1024.0 * factor / scans)
doc["scans_profiled"] = nscans
doc["kernels"] = per_kernel
for base, e in per_kernel.items():
    s = doc.setdefault(e["span"], {"fetch_bytes": 0, "write_bytes": 0, "kernels": []})
    s["fetch_bytes"] += e["fetch_bytes"]; s["write_bytes"] += e["write_bytes"]; s["kernels"].append(base)
if "k_gain" in doc:
    doc["k_gain"]["note"] = "k_gain runs twice per scan (heads, tails): both launches are counted here, none under k_gain_tail"
print(json.dumps(doc, indent=1))
