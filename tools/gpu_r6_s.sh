#!/bin/bash
# round 6, run S: k_sift_hard reads the candidate's rows in aligned 16-byte pieces
mkdir -p gpurun_out/r06s
for lab in n1 n1f; do
  cfgn=$(echo $lab | sed 's/n1f/N1f/; s/^n1$/N1/')
  timeout 600 python bench.py --no-cpu-baseline --no-e2e --no-other-configs --config $cfgn --steps 5 --warmup 2 > gpurun_out/r06s/$lab.json 2>/dev/null
  python -c "
import json; j=json.loads(open('gpurun_out/r06s/$lab.json').read().strip().splitlines()[-1]); print('$lab', j['value'], j['ms_per_step'], j['ms_per_step_serial'], {k: v for k, v in j['kernel_ms'].items() if v > 0.02})"
done
RTFE_DEBUG=9 timeout 300 python bench.py --config N1 --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-other-configs 2>/dev/null | python -c "
import json, sys
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); p = j['config']['last_scan_stats']['phase_cycles']
print('k_sift_hard on N1 (RTFE_DEBUG=9), cycles of the workgroups\' first waves: fetch', p[0], 'walk', p[2], 'trips', p[3], 'workgroups', p[4], '-> per trip', p[0] // max(p[3], 1), '+', p[2] // max(p[3], 1))" | tee gpurun_out/r06s/sift_hard_phases.txt
timeout 600 python tools/fuzz_shapes.py --gpu 13000 100 > gpurun_out/r06s/fuzz.log 2>&1; echo "fuzz ok $(grep -c '^ok' gpurun_out/r06s/fuzz.log) fail $(grep -c '^FAIL ' gpurun_out/r06s/fuzz.log)"
STRESS_SHAPES=1 timeout 800 bash tools/gpu_stress.sh 3100 1 60
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 600 bash tools/gpu_profile.sh r06s_n1 --config N1 --steps 5 --warmup 2 --no-overlap > gpurun_out/r06s/profile_n1.log 2>&1; cp gpurun_out/prof_r06s_n1/summary.txt gpurun_out/r06s/rocprof_summary_n1.txt; cp gpurun_out/prof_r06s_n1/bench_under_rocprof.json gpurun_out/r06s/bench_under_rocprof_n1.json; head -8 gpurun_out/r06s/rocprof_summary_n1.txt
