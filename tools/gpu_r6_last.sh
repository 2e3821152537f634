#!/bin/bash
# round 6: the last commit's library under the shape fuzzer and the shaped stress sweep once more
mkdir -p gpurun_out/r06last
for s in 70000 70100 70200 100900; do
  timeout 700 python tools/fuzz_shapes.py --gpu $s 100 > gpurun_out/r06last/fuzz_$s.log 2>&1; echo "fuzz $s rc $? ok $(grep -c '^ok' gpurun_out/r06last/fuzz_$s.log) fail $(grep -c '^FAIL ' gpurun_out/r06last/fuzz_$s.log)"
  grep -A3 '^FAIL ' gpurun_out/r06last/fuzz_$s.log | head -12
done
STRESS_SHAPES=1 timeout 1500 bash tools/gpu_stress.sh 3400 2 60
timeout 700 bash tools/gpu_stress.sh 3410 1 60
