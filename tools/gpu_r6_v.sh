#!/bin/bash
# round 6, run V: the shape fuzzer with shapes over the blocks' first peaks (the chains' start-up)
mkdir -p gpurun_out/r06v
for s in 100100 100200 100300 100400 100500 100600; do
  timeout 700 python tools/fuzz_shapes.py --gpu $s 100 > gpurun_out/r06v/fuzz_$s.log 2>&1; echo "fuzz $s rc $? ok $(grep -c '^ok' gpurun_out/r06v/fuzz_$s.log) fail $(grep -c '^FAIL ' gpurun_out/r06v/fuzz_$s.log)"
  grep -A3 '^FAIL ' gpurun_out/r06v/fuzz_$s.log | head -12
done
timeout 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -2
