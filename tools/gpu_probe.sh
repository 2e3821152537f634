#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/gpu_tests.log 2>&1; echo "tests rc $?"; tail -4 gpurun_out/gpu_tests.log
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/quick_bench.json 2> gpurun_out/quick_bench.err; echo "bench rc $?"; tail -c 1500 gpurun_out/quick_bench.json
