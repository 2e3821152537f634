#!/bin/bash
# quick bounded GPU check: golden parity through the C ABI, then a short bench (every step has its own timeout; logs go to gpurun_out/)
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "test_golden_tapes or test_fresh_nrzi_tapes or test_fresh_pe_tape" > gpurun_out/quick_tests.log 2>&1; echo "tests rc $?"; tail -5 gpurun_out/quick_tests.log
timeout 600 python bench.py --steps ${STEPS:-5} --warmup 2 --no-cpu-baseline > gpurun_out/quick_bench.json 2> gpurun_out/quick_bench.err; echo "bench rc $?"; tail -c 2500 gpurun_out/quick_bench.json; tail -3 gpurun_out/quick_bench.err
