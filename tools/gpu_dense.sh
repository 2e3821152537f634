#!/bin/bash
mkdir -p gpurun_out
run() { PROBE_COPIES=$1 timeout -s INT 100 python -X faulthandler tools/gpu_dense_probe.py 5e6 $2 $3 2>&1 | grep -E "^scan 2" | tr '\n' ' '; }
for cfgs in "8 gcr" "5 gcr" "1 gcr" "8 pe" "1 pe"; do
  for c in 2 7 14 56; do
    echo -n "sets/kind $cfgs copies $c: dense "; RTFE_DENSE_PATH=1 run $c $cfgs; echo -n " | old "; RTFE_DENSE_PATH=0 run $c $cfgs; echo
  done
done
