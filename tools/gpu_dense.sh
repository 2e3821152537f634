#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
bash tools/gpu_stress.sh 974 1 100
bash tools/gpu_stress.sh 990 2 60
