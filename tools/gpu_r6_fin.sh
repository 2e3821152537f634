#!/bin/bash
# round 6, last run: k_emit_seg at eight waves a SIMD - the GPU suite, the driver's line, the kernel summaries of C2 and M8
mkdir -p gpurun_out/r06fin
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/r06fin/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 1200 python bench.py --steps 20 --warmup 5 > gpurun_out/r06fin/bench_default.json 2> gpurun_out/r06fin/bench_default.err; echo "bench rc $?"; python -c "
import json; j=json.loads(open('gpurun_out/r06fin/bench_default.json').read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['ms_per_step_serial'], j['roofline']['frac'], j['cpu_baseline']['value'], j['e2e']['value'], j['kernel_ms']); print({k: (v.get('ms_per_step'), v.get('error')) for k, v in j['other_configs'].items()})"
for spec in "c2:--steps 20 --warmup 5 --no-other-configs --no-overlap" "m8:--config M8 --steps 5 --warmup 2 --no-overlap" "c5:--config C5 --steps 5 --warmup 2"; do
  tag=${spec%%:*}; args=${spec#*:}
  timeout 900 bash tools/gpu_profile.sh r06fin_$tag $args > gpurun_out/r06fin/profile_$tag.log 2>&1; echo "profile $tag rc $?"; cp gpurun_out/prof_r06fin_$tag/summary.txt gpurun_out/r06fin/rocprof_summary_$tag.txt; cp gpurun_out/prof_r06fin_$tag/bench_under_rocprof.json gpurun_out/r06fin/bench_under_rocprof_$tag.json; rm -rf gpurun_out/prof_r06fin_$tag; head -7 gpurun_out/r06fin/rocprof_summary_$tag.txt
done
timeout 600 python tools/fuzz_shapes.py --gpu 60000 100 > gpurun_out/r06fin/fuzz.log 2>&1; echo "fuzz ok $(grep -c '^ok' gpurun_out/r06fin/fuzz.log) fail $(grep -c '^FAIL ' gpurun_out/r06fin/fuzz.log)"
