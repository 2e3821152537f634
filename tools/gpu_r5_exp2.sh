#!/bin/bash
# GPU box, round 5 experiment 2: fixed margin blocks - parity, C2 serial and pipelined, M8
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -3
for i in 1 2; do bash tools/gpu_try.sh "A=1" --steps 20 --warmup 5 --no-other-configs; done
bash tools/gpu_try.sh "A=1" --steps 20 --warmup 5 --no-other-configs --pipeline
bash tools/gpu_try.sh "A=1" --config M8 --steps 5 --warmup 2
bash tools/gpu_try.sh "A=1" --config C5 --steps 10 --warmup 2
