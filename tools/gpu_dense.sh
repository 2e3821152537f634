#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
bash tools/gpu_round4_final.sh
timeout 700 bash tools/gpu_traffic.sh r04 C4 > gpurun_out/traffic_C4.log 2>&1; echo "traffic C4 rc $?"
