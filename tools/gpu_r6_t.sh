#!/bin/bash
# round 6, run T: the final library under the shape fuzzer and the shaped stress sweep, at length
mkdir -p gpurun_out/r06t
for s in 20000 20100 20200 20300 20400 20500 20600 20700; do
  timeout 700 python tools/fuzz_shapes.py --gpu $s 100 > gpurun_out/r06t/fuzz_$s.log 2>&1; echo "fuzz $s rc $? ok $(grep -c '^ok' gpurun_out/r06t/fuzz_$s.log) fail $(grep -c '^FAIL ' gpurun_out/r06t/fuzz_$s.log)"
  grep -A3 '^FAIL ' gpurun_out/r06t/fuzz_$s.log | head -12
done
STRESS_SHAPES=1 timeout 2400 bash tools/gpu_stress.sh 3200 4 60
timeout 1500 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
