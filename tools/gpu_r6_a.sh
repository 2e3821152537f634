#!/bin/bash
# GPU box, round 6 run A: the whole -m gpu suite on the dieted k_sift_s + the in-scan floor probe, the driver's line with every other config, SQ counters of C2.
mkdir -p gpurun_out/r06a
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/r06a/pytest_gpu.txt
timeout 1500 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e > gpurun_out/r06a/bench.json 2> gpurun_out/r06a/bench.err; echo "bench rc $?"
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r06a/bench.json").read().strip().splitlines()[-1])
print({k: j[k] for k in ("value", "ms_per_step", "ms_per_step_serial", "timed_steps")}, j["roofline"]["kernel"], j["roofline"]["frac"], j["kernel_ms"])
for k, v in j.get("other_configs", {}).items(): print(k, {kk: v.get(kk) for kk in ("value", "ms_per_step", "ms_per_step_serial", "dominant_kernel", "dominant_kernel_ms", "frac", "flagged_bursts", "error")}, (v.get("last_scan_stats") or {}).get("redone"), (v.get("last_scan_stats") or {}).get("screen_floor_used"), v.get("kernel_ms"))
PY
timeout 600 bash tools/gpu_pmc.sh --no-other-configs --no-overlap > gpurun_out/r06a/sq_counters_c2_a.txt 2>&1
timeout 600 bash tools/gpu_pmc2.sh --no-other-configs --no-overlap > gpurun_out/r06a/sq_counters_c2_b.txt 2>&1
grep -h "k_sift_s" gpurun_out/r06a/sq_counters_c2_a.txt gpurun_out/r06a/sq_counters_c2_b.txt | head -40
rm -rf gpurun_out/pmc_sq gpurun_out/pmc_sq2
# the shader clock under k_sift_s: busy cycles of the graphics block over the kernel's duration (one pass, counters + kernel trace)
export TMPDIR=/tmp
out=$PWD/gpurun_out/pmc_clk; rm -rf $out; mkdir -p $out
(cd /tmp && rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --kernel-trace -d $out -o clk -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-other-configs --no-overlap > /dev/null 2> $out/err.log)
python - <<PY | tee gpurun_out/r06a/clock_c2.txt
import sqlite3, glob
for f in sorted(glob.glob("$out/**/*.db", recursive=True)):
    db = sqlite3.connect(f)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    try:
        rows = list(db.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"))
        for kn, cn, n, v in rows:
            if "k_sift_s" in kn or "k_prep" in kn or "k_gain_seg" in kn: print(kn.split("(")[0][:24], cn, n, f"{v:.5g}")
    except Exception as e: print("counters:", e)
    try:
        for kn, n, d in db.execute("select name, count(*), avg(end - start) from kernels group by name"):
            if "k_sift_s" in kn or "k_prep" in kn or "k_gain_seg" in kn: print(kn.split("(")[0][:24], "dispatches", n, "avg ns", f"{d:.6g}")
    except Exception as e: print("kernels:", e, [t for t in tabs if "kernel" in t][:8])
PY
tail -2 $out/err.log; rm -rf $out
