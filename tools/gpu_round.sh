#!/bin/bash
# GPU box, one call: the full -m gpu suite, smoke, the driver's bench line, the rocprofv3 summaries for C2 and C3.
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; echo "gpu tests rc $?"; tail -3 gpurun_out/gpu_tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1; echo "smoke rc $?"; tail -1 gpurun_out/smoke.log
timeout 600 python bench.py > gpurun_out/bench_c2.json 2> gpurun_out/bench_c2.err; echo "C2 rc $?"; tail -c 600 gpurun_out/bench_c2.json
timeout 1200 bash tools/gpu_profile.sh r02 > gpurun_out/profile.log 2>&1; echo "profile rc $?"; head -12 gpurun_out/profile.log
timeout 1500 bash tools/gpu_profile.sh r02_c3 --config C3 --steps 2 --warmup 1 > gpurun_out/profile_c3.log 2>&1; echo "profile C3 rc $?"; head -8 gpurun_out/profile_c3.log
for c in C3 C4 C5; do
  timeout 1200 python bench.py --config $c > gpurun_out/bench_$c.json 2> gpurun_out/bench_$c.err; echo "$c rc $?"; tail -c 300 gpurun_out/bench_$c.json
done
