#!/bin/bash
# GPU box, round 5 mid-round: new parity tests, the driver's line with other_configs (incl. the noise lines), C2 traffic and SQ counters.
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -k "p1 or noise" 2>&1 | tail -3
timeout 1500 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc $?"
python - <<'PY'
import json
j = json.loads(open("gpurun_out/bench_default.json").read().strip().splitlines()[-1])
print({k: j[k] for k in ("value", "ms_per_step", "timed_steps", "timed_seconds")}, j["roofline"]["kernel"], j["roofline"]["frac"], j["roofline"]["whole_step"]["frac"], "traffic", j["roofline"]["traffic"], j["roofline"].get("traffic_all_kernels"))
for k, v in j.get("other_configs", {}).items(): print(k, {kk: v.get(kk) for kk in ("value", "ms_per_step", "dominant_kernel", "dominant_kernel_ms", "frac", "frac_rows_only", "whole_step_frac", "flagged_bursts", "last_scan_stats", "error")})
print("e2e", {k: j["e2e"].get(k) for k in ("value", "seconds", "tap_identical_to_cpu_port", "error")}, "cpu", j.get("cpu_baseline", {}).get("value"), j.get("cpu_baseline", {}).get("kind"))
PY
timeout 900 bash tools/gpu_traffic.sh r05 C2 > gpurun_out/traffic_c2.log 2>&1; echo "traffic C2 rc $?"; cp gpurun_out/traffic_r05_C2/pmc_C2.json gpurun_out/pmc_C2.json
timeout 600 bash tools/gpu_pmc.sh --no-other-configs > gpurun_out/r05_sq_a.txt 2>&1; grep "k_sift_s\|k_prep\|k_emit_seg\|k_gain_seg" gpurun_out/r05_sq_a.txt
rm -rf gpurun_out/pmc_sq gpurun_out/traffic_r05_C2/calib_* gpurun_out/traffic_r05_C2/pmc_fetch gpurun_out/traffic_r05_C2/pmc_write
