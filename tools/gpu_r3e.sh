#!/bin/bash
mkdir -p gpurun_out
for v in "RTFE_VERBOSE=1" "RTFE_CUT=1" "RTFE_CUT=2" "RTFE_CUT=3" "RTFE_CUT=4" "RTFE_SIFT_GENERIC=1"; do
  env RTFE_PEAK_PATH=1 RTFE_PEAK_STOP=1 $v timeout 600 python bench.py --no-cpu-baseline --no-e2e --steps 5 --warmup 2 > gpurun_out/r3d.json 2> gpurun_out/r3d.err; echo "[$v] rc $?"; grep "rtfe:" gpurun_out/r3d.err | head -8; cat gpurun_out/r3d.json | python -c "import sys,json; j=json.loads(sys.stdin.read()); print({k:v for k,v in j['kernel_ms'].items() if v>0.01})"
done
