#!/bin/bash
# GPU box, round 6 evidence: the whole -m gpu suite, the driver's line, rocprofv3 kernel summaries and PMC traffic of every configuration, SQ counters of C2 and G1,
# the k_sift_s phase profile, a rank's share of C5.
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/r06/pytest_gpu.txt
timeout 2400 python bench.py --steps 20 --warmup 5 > gpurun_out/r06/bench_default.json 2> gpurun_out/r06/bench_default.err; echo "bench rc $?"
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r06/bench_default.json").read().strip().splitlines()[-1])
print({k: j[k] for k in ("value", "ms_per_step", "ms_per_step_serial", "timed_steps", "timed_seconds")}, j["roofline"]["kernel"], j["roofline"]["frac"], j["roofline"]["whole_step"]["frac"], "traffic", j["roofline"]["traffic"], j["roofline"].get("traffic_all_kernels"))
for k, v in j.get("other_configs", {}).items(): print(k, {kk: v.get(kk) for kk in ("value", "ms_per_step", "ms_per_step_serial", "dominant_kernel", "dominant_kernel_ms", "frac", "launches_per_step", "flagged_bursts", "error")}, (v.get("last_scan_stats") or {}).get("redone"), (v.get("last_scan_stats") or {}).get("screen_floor_used"))
print("e2e", {k: j["e2e"].get(k) for k in ("value", "seconds", "tap_identical_to_cpu_port", "error")}, "cpu", j.get("cpu_baseline", {}).get("value"), j.get("cpu_baseline", {}).get("kind"))
PY
RTFE_LIB_PATH=$PWD/readtape_amd/librtfe_prof.so timeout 300 python tools/gpu_sift_prof.py C2 2>&1 | tail -10 | tee gpurun_out/r06/sift_phases.txt
for spec in "c2:--steps 20 --warmup 5 --no-other-configs --no-overlap" "c2f:--config C2f --steps 10 --warmup 2 --no-overlap" "c3:--config C3 --steps 4 --warmup 1 --no-overlap" "c4:--config C4 --steps 2 --warmup 1" "g1:--config G1 --steps 2 --warmup 1" "p1:--config P1 --steps 2 --warmup 1" "m8:--config M8 --steps 5 --warmup 2 --no-overlap" "m8f:--config M8f --steps 5 --warmup 2 --no-overlap" "n1:--config N1 --steps 5 --warmup 2 --no-overlap" "n1f:--config N1f --steps 5 --warmup 2 --no-overlap" "n2:--config N2 --steps 2 --warmup 1" "c5:--config C5 --steps 5 --warmup 2 --no-graphs"; do
  tag=${spec%%:*}; args=${spec#*:}
  timeout 900 bash tools/gpu_profile.sh r06_$tag $args > gpurun_out/r06/profile_$tag.log 2>&1; echo "profile $tag rc $?"; cp gpurun_out/prof_r06_$tag/summary.txt gpurun_out/r06/rocprof_summary_$tag.txt; cp gpurun_out/prof_r06_$tag/bench_under_rocprof.json gpurun_out/r06/bench_under_rocprof_$tag.json; rm -rf gpurun_out/prof_r06_$tag
done
for spec in "C2:--no-other-configs --no-overlap" "C3:--no-overlap" "C4:" "C5:" "G1:" "P1:" "M8:--no-overlap" "M8f:--no-overlap" "N1f:--no-overlap" "N2:"; do
  cfg=${spec%%:*}; args=${spec#*:}
  timeout 1200 bash tools/gpu_traffic.sh r06 $cfg $args > gpurun_out/r06/traffic_$cfg.log 2>&1; echo "traffic $cfg rc $?"; cp gpurun_out/traffic_r06_$cfg/pmc_$cfg.json gpurun_out/r06/pmc_$cfg.json; rm -rf gpurun_out/traffic_r06_$cfg
done
timeout 600 bash tools/gpu_pmc.sh --no-other-configs --no-overlap > gpurun_out/r06/sq_counters_c2_a.txt 2>&1
timeout 600 bash tools/gpu_pmc2.sh --no-other-configs --no-overlap > gpurun_out/r06/sq_counters_c2_b.txt 2>&1
timeout 900 bash tools/gpu_pmc.sh --config G1 > gpurun_out/r06/sq_counters_g1_a.txt 2>&1
timeout 900 bash tools/gpu_pmc2.sh --config G1 > gpurun_out/r06/sq_counters_g1_b.txt 2>&1
rm -rf gpurun_out/pmc_sq gpurun_out/pmc_sq2
for r in 5.54e8 6.9e7; do timeout 300 python bench.py --config C5 --rows $r --steps 20 --warmup 5 --no-cpu-baseline --no-e2e > gpurun_out/r06/bench_c5_rows_$r.json 2>/dev/null; done
RTFE_LIB_PATH=$PWD/readtape_amd/librtfe_hprof.so RTFE_DEBUG=9 timeout 300 python bench.py --config N1      `# (python tools/build_variant.py hprof -DRTFE_HARD_PROF=1)` --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-other-configs 2>/dev/null | python -c "
import json, sys
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); p = j['config']['last_scan_stats']['phase_cycles']
print('k_sift_hard on N1 (RTFE_DEBUG=9), cycles of the workgroups\' first waves: fetch', p[0], 'walk', p[2], 'trips', p[3], 'workgroups', p[4], '-> per trip', p[0] // max(p[3], 1), '+', p[2] // max(p[3], 1))" | tee gpurun_out/r06/sift_hard_phases.txt
for s in 12000 12100; do timeout 900 python tools/fuzz_shapes.py --gpu $s 100 > gpurun_out/r06/fuzz_$s.log 2>&1; echo "fuzz $s ok $(grep -c '^ok' gpurun_out/r06/fuzz_$s.log) fail $(grep -c '^FAIL ' gpurun_out/r06/fuzz_$s.log)"; done | tee gpurun_out/r06/fuzz_shapes.txt
ls gpurun_out/r06
