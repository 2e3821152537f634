#!/bin/bash
# GPU box, one call: the full -m gpu suite, smoke, the driver's bench line, the rocprofv3 summary and the PMC traffic for C2.
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; echo "gpu tests rc $?"; tail -3 gpurun_out/gpu_tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1; echo "smoke rc $?"; tail -1 gpurun_out/smoke.log
timeout 900 bash tools/gpu_profile.sh r03 > gpurun_out/profile.log 2>&1; echo "profile rc $?"; head -24 gpurun_out/profile.log
timeout 900 bash tools/gpu_traffic.sh r03 C2 > gpurun_out/traffic.log 2>&1; echo "traffic rc $?"; mkdir -p profiles; cp gpurun_out/traffic_r03_C2/pmc_C2.json profiles/pmc_C2.json 2>/dev/null
timeout 600 python bench.py > gpurun_out/bench_c2.json 2> gpurun_out/bench_c2.err; echo "C2 rc $?"; tail -c 1500 gpurun_out/bench_c2.json
