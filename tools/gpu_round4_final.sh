#!/bin/bash
# GPU box, end of round 4: the driver's line (with other_configs), rocprofv3 kernel summaries, PMC traffic (C2, C4), C5 whole and one rank's share of eight.
mkdir -p gpurun_out
timeout 1500 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc $?"
python - <<'PY'
import json
j = json.loads(open("gpurun_out/bench_default.json").read().strip().splitlines()[-1])
print({k: j[k] for k in ("value", "ms_per_step", "timed_steps", "timed_seconds")}, j["roofline"]["kernel"], j["roofline"]["frac"], j["roofline"]["whole_step"]["frac"], "traffic", j["roofline"]["traffic"])
for k, v in j.get("other_configs", {}).items(): print(k, {kk: v.get(kk) for kk in ("value", "ms_per_step", "dominant_kernel", "dominant_kernel_ms", "frac", "whole_step_frac", "flagged_bursts", "traffic", "error")})
print("e2e", {k: j["e2e"].get(k) for k in ("value", "seconds", "tap_identical_to_cpu_port", "error")}, "cpu", j.get("cpu_baseline", {}).get("value"), j.get("cpu_baseline", {}).get("kind"))
PY
timeout 600 bash tools/gpu_profile.sh r04_c4 --config C4 --steps 2 --warmup 1 > gpurun_out/profile_c4.log 2>&1; echo "profile C4 rc $?"; head -8 gpurun_out/profile_c4.log
timeout 600 bash tools/gpu_profile.sh r04 --steps 20 --warmup 5 --no-other-configs > gpurun_out/profile_c2.log 2>&1; echo "profile C2 rc $?"; head -6 gpurun_out/profile_c2.log
timeout 900 bash tools/gpu_traffic.sh r04 C4 > gpurun_out/traffic_c4.log 2>&1; echo "traffic C4 rc $?"
for r in 5.54e8 6.9e7; do
  timeout 300 python bench.py --config C5 --rows $r --steps 20 --warmup 5 --no-cpu-baseline --no-e2e > gpurun_out/bench_c5_rows_$r.json 2>/dev/null
  python -c "
import json; j=json.loads(open('gpurun_out/bench_c5_rows_$r.json').read().strip().splitlines()[-1]); print('C5 rows $r:', j['config']['rows_per_gpu'], j['value'], j['ms_per_step'])"
done
timeout 600 python bench.py --config C4 > gpurun_out/bench_c4.json 2> gpurun_out/bench_c4_full.err; echo "C4 full line rc $?"
timeout 600 python bench.py --config C3 > gpurun_out/bench_c3.json 2> gpurun_out/bench_c3_full.err; echo "C3 full line rc $?"
timeout 600 bash tools/gpu_profile.sh r04_g1 --config G1 --steps 2 --warmup 1 > gpurun_out/profile_g1.log 2>&1; echo "profile G1 rc $?"; head -8 gpurun_out/profile_g1.log
timeout 600 bash tools/gpu_profile.sh r04_c5 --config C5 --steps 5 --warmup 2 > gpurun_out/profile_c5.log 2>&1; echo "profile C5 rc $?"; head -6 gpurun_out/profile_c5.log
