#!/bin/bash
# GPU box: random tapes x option combinations on every path (peak, sample, dense) against the oracle / each other.  usage: gpu_stress.sh <seed0> <nseeds> <tapes per seed>
mkdir -p gpurun_out
s0=${1:-930}; n=${2:-4}; k=${3:-60}
for ((s=s0; s<s0+n; s++)); do
  timeout 700 python tests/stress_gpu.py $s $k > gpurun_out/stress_$s.log 2>&1; echo "seed $s rc $? ok $(grep -c '^ok' gpurun_out/stress_$s.log) fail $(grep -c '^FAIL' gpurun_out/stress_$s.log)"
  grep -A4 '^FAIL' gpurun_out/stress_$s.log | head -20
done
