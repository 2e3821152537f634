#!/bin/bash
# GPU box, one call: the full -m gpu suite, smoke, the rocprofv3 kernel summary and the PMC traffic for C2 (and C3), the driver's bench
# line, the other configurations.  Everything lands in gpurun_out/; the builder copies the summaries to profiles/r03_*.
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/gpu_tests.log 2>&1; echo "gpu tests rc $?"; tail -3 gpurun_out/gpu_tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1; echo "smoke rc $?"; tail -1 gpurun_out/smoke.log
timeout 900 bash tools/gpu_profile.sh r03 --steps 20 --warmup 5 > gpurun_out/profile.log 2>&1; echo "profile rc $?"; head -24 gpurun_out/profile.log
timeout 900 bash tools/gpu_profile.sh r03_c3 --config C3 --steps 5 --warmup 2 > gpurun_out/profile_c3.log 2>&1; echo "profile C3 rc $?"; head -12 gpurun_out/profile_c3.log
timeout 900 bash tools/gpu_traffic.sh r03 C2 > gpurun_out/traffic.log 2>&1; echo "traffic C2 rc $?"; cp gpurun_out/traffic_r03_C2/pmc_C2.json profiles/pmc_C2.json 2>/dev/null
timeout 1500 bash tools/gpu_traffic.sh r03 C3 > gpurun_out/traffic_c3.log 2>&1; echo "traffic C3 rc $?"; cp gpurun_out/traffic_r03_C3/pmc_C3.json profiles/pmc_C3.json 2>/dev/null
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_c2.json 2> gpurun_out/bench_c2.err; echo "C2 rc $?"; tail -c 2500 gpurun_out/bench_c2.json
for c in C3 C4 C5; do
  timeout 1500 python bench.py --config $c > gpurun_out/bench_$c.json 2> gpurun_out/bench_$c.err; echo; echo "$c rc $?"; tail -c 1400 gpurun_out/bench_$c.json
done
