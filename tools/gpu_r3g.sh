#!/bin/bash
mkdir -p gpurun_out
RTFE_PEAK_PATH=1 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden_tapes or fresh_nrzi or large_tape or peak_record_path" > gpurun_out/r3b_tests.log 2>&1; echo "tests rc $?"; tail -3 gpurun_out/r3b_tests.log
for v in "" "RTFE_GAIN_FAST=0"; do
  env RTFE_PEAK_PATH=1 $v timeout 600 python bench.py --no-cpu-baseline --no-e2e --steps 5 --warmup 2 > gpurun_out/r3b_new.json 2> gpurun_out/r3b_new.err; echo "[$v] rc $?"; cat gpurun_out/r3b_new.json | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], {k:v for k,v in j['kernel_ms'].items() if v>0.01}, j['roofline']['frac'], j['config']['flagged_bursts'], j['config']['events_per_gpu'])"
done
timeout 600 python tools/gpu_sift_phase.py 1e8 2>&1 | tail -4
