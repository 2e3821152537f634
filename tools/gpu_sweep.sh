#!/bin/bash
# GPU box: the randomized parity sweep (tests/stress_gpu.py) with several seeds, bounded; logs as it goes
mkdir -p gpurun_out
for seed in "$@"; do
  timeout 700 python tests/stress_gpu.py $seed 150 > gpurun_out/stress_$seed.log 2>&1
  echo "seed $seed rc $? $(grep -c . gpurun_out/stress_$seed.log) lines; last: $(tail -1 gpurun_out/stress_$seed.log | cut -c1-200)"
done
