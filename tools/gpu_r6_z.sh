#!/bin/bash
# round 6, run Z: k_sift_hard at six waves a SIMD (79 registers: one instantiation of the walk, compile-time counters)
mkdir -p gpurun_out/r06z
one() { local label=$1; shift
   env "$@" timeout 900 python bench.py --no-cpu-baseline --no-e2e --no-other-configs $EXTRA > gpurun_out/r06z/$label.json 2> gpurun_out/r06z/$label.err
   python -c "
import json; j=json.loads(open('gpurun_out/r06z/$label.json').read().strip().splitlines()[-1]); print('$label', j['value'], j['ms_per_step'], j['ms_per_step_serial'], {k: v for k, v in j['kernel_ms'].items() if v > 0.02})"
}
EXTRA="--config N1 --steps 5 --warmup 2" one n1 A=1
EXTRA="--config N1f --steps 5 --warmup 2" one n1f A=1
EXTRA="--config M8 --steps 5 --warmup 2" one m8 A=1
EXTRA="--steps 20 --warmup 5" one c2 A=1
timeout 600 python tools/fuzz_shapes.py --gpu 50000 100 > gpurun_out/r06z/fuzz.log 2>&1; echo "fuzz ok $(grep -c '^ok' gpurun_out/r06z/fuzz.log) fail $(grep -c '^FAIL ' gpurun_out/r06z/fuzz.log)"
STRESS_SHAPES=1 timeout 800 bash tools/gpu_stress.sh 3300 1 60
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/r06z/pytest_gpu.txt
for tag in n1 n1f; do cfgn=$(echo $tag | sed 's/n1f/N1f/; s/^n1$/N1/')
  timeout 600 bash tools/gpu_profile.sh r06z_$tag --config $cfgn --steps 5 --warmup 2 --no-overlap > gpurun_out/r06z/profile_$tag.log 2>&1; cp gpurun_out/prof_r06z_$tag/summary.txt gpurun_out/r06z/rocprof_summary_$tag.txt; cp gpurun_out/prof_r06z_$tag/bench_under_rocprof.json gpurun_out/r06z/bench_under_rocprof_$tag.json; rm -rf gpurun_out/prof_r06z_$tag; head -6 gpurun_out/r06z/rocprof_summary_$tag.txt
done
timeout 1200 python bench.py > gpurun_out/r06z/bench_default.json 2> gpurun_out/r06z/bench_default.err; echo "bench rc $?"; python -c "
import json; j=json.loads(open('gpurun_out/r06z/bench_default.json').read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['cpu_baseline']['value'], j['e2e']['value']); print({k: (v.get('ms_per_step'), v.get('error')) for k, v in j['other_configs'].items()})"
