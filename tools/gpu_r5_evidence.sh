#!/bin/bash
# GPU box, round 5 evidence: the whole -m gpu suite, the driver's line, rocprofv3 kernel summaries and PMC traffic of every configuration, SQ counters of C2.
mkdir -p gpurun_out/r05
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/r05/pytest_gpu.txt
timeout 1800 python bench.py --steps 20 --warmup 5 > gpurun_out/r05/bench_default.json 2> gpurun_out/r05/bench_default.err; echo "bench rc $?"
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r05/bench_default.json").read().strip().splitlines()[-1])
print({k: j[k] for k in ("value", "ms_per_step", "ms_per_step_serial", "timed_steps", "timed_seconds")}, j["roofline"]["kernel"], j["roofline"]["frac"], j["roofline"]["whole_step"]["frac"], "traffic", j["roofline"]["traffic"], j["roofline"].get("traffic_all_kernels"))
for k, v in j.get("other_configs", {}).items(): print(k, {kk: v.get(kk) for kk in ("value", "ms_per_step", "ms_per_step_serial", "dominant_kernel", "dominant_kernel_ms", "frac", "frac_rows_only", "flagged_bursts", "screen_floor_height", "error")})
print("e2e", {k: j["e2e"].get(k) for k in ("value", "seconds", "tap_identical_to_cpu_port", "error")}, "cpu", j.get("cpu_baseline", {}).get("value"), j.get("cpu_baseline", {}).get("kind"))
PY
for spec in "c2:--steps 20 --warmup 5 --no-other-configs --no-overlap" "c3:--config C3 --steps 4 --warmup 1 --no-overlap" "c4:--config C4 --steps 2 --warmup 1" "g1:--config G1 --steps 2 --warmup 1" "p1:--config P1 --steps 2 --warmup 1" "m8:--config M8 --steps 5 --warmup 2 --no-overlap" "m8f:--config M8f --steps 3 --warmup 1 --no-overlap" "n1:--config N1 --steps 5 --warmup 2 --no-overlap" "c5:--config C5 --steps 5 --warmup 2 --no-graphs"; do
  tag=${spec%%:*}; args=${spec#*:}
  timeout 900 bash tools/gpu_profile.sh r05_$tag $args > gpurun_out/r05/profile_$tag.log 2>&1; echo "profile $tag rc $?"; cp gpurun_out/prof_r05_$tag/summary.txt gpurun_out/r05/rocprof_summary_$tag.txt; cp gpurun_out/prof_r05_$tag/bench_under_rocprof.json gpurun_out/r05/bench_under_rocprof_$tag.json; rm -rf gpurun_out/prof_r05_$tag
done
for spec in "C2:--no-other-configs --no-overlap" "C3:--no-overlap" "C4:" "G1:" "P1:" "M8:--no-overlap"; do
  cfg=${spec%%:*}; args=${spec#*:}
  timeout 1200 bash tools/gpu_traffic.sh r05 $cfg $args > gpurun_out/r05/traffic_$cfg.log 2>&1; echo "traffic $cfg rc $?"; cp gpurun_out/traffic_r05_$cfg/pmc_$cfg.json gpurun_out/r05/pmc_$cfg.json; rm -rf gpurun_out/traffic_r05_$cfg
done
timeout 600 bash tools/gpu_pmc.sh --no-other-configs --no-overlap > gpurun_out/r05/sq_counters_c2_a.txt 2>&1
timeout 600 bash tools/gpu_pmc2.sh --no-other-configs --no-overlap > gpurun_out/r05/sq_counters_c2_b.txt 2>&1
rm -rf gpurun_out/pmc_sq gpurun_out/pmc_sq2
for r in 5.54e8 6.9e7; do timeout 300 python bench.py --config C5 --rows $r --steps 20 --warmup 5 --no-cpu-baseline --no-e2e > gpurun_out/r05/bench_c5_rows_$r.json 2>/dev/null; timeout 300 python bench.py --config C5 --rows $r --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --graphs > gpurun_out/r05/bench_c5_rows_${r}_graphs.json 2>/dev/null; done
ls gpurun_out/r05
