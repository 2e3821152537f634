#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_abi.py tests/test_gpu_fullsize.py -m gpu -x -q -k "c_client or bench_tape or c4_full or c5_shape" > gpurun_out/gpu_tests_new.log 2>&1; echo "new gpu tests rc $?"; tail -12 gpurun_out/gpu_tests_new.log
