#!/bin/bash
# GPU box, one call: tests, the bench lines of every config, the rocprofv3 summaries - every step under its own timeout, logs as it goes.
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ingest.py -x -q > gpurun_out/ingest_tests.log 2>&1; echo "ingest rc $?"; tail -5 gpurun_out/ingest_tests.log
timeout 900 python bench.py --config C4 --steps 3 --warmup 1 > gpurun_out/bench_c4.json 2> gpurun_out/bench_c4.err; echo "C4 rc $?"; tail -c 3000 gpurun_out/bench_c4.json; tail -3 gpurun_out/bench_c4.err
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; echo "gpu tests rc $?"; tail -5 gpurun_out/gpu_tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1; echo "smoke rc $?"; tail -2 gpurun_out/smoke.log
timeout 1200 bash tools/gpu_profile.sh r02 > gpurun_out/profile.log 2>&1; echo "profile rc $?"; tail -30 gpurun_out/profile.log
