#!/bin/bash
# GPU box, one call: the full -m gpu suite, smoke, the bench lines of every config, the rocprofv3 summaries.
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; echo "gpu tests rc $?"; tail -3 gpurun_out/gpu_tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1; echo "smoke rc $?"; tail -1 gpurun_out/smoke.log
timeout 600 python bench.py > gpurun_out/bench_c2.json 2> gpurun_out/bench_c2.err; echo "C2 rc $?"; tail -c 3000 gpurun_out/bench_c2.json
timeout 900 python bench.py --config C3 --steps 3 --warmup 1 > gpurun_out/bench_c3.json 2> gpurun_out/bench_c3.err; echo "C3 rc $?"; tail -c 1500 gpurun_out/bench_c3.json | cut -c1-1500
timeout 900 python bench.py --config C4 --steps 3 --warmup 1 > gpurun_out/bench_c4.json 2> gpurun_out/bench_c4.err; echo "C4 rc $?"; tail -c 1500 gpurun_out/bench_c4.json | cut -c1-1200
timeout 900 python bench.py --config C5 --steps 5 --warmup 1 --no-e2e --no-cpu-baseline > gpurun_out/bench_c5.json 2> gpurun_out/bench_c5.err; echo "C5 (1 GPU) rc $?"; tail -c 1200 gpurun_out/bench_c5.json
timeout 1200 bash tools/gpu_profile.sh r02 > gpurun_out/profile.log 2>&1; echo "profile rc $?"; tail -28 gpurun_out/profile.log
