#!/bin/bash
mkdir -p gpurun_out
env RTFE_PEAK_PATH=1 timeout 600 python bench.py --no-cpu-baseline --no-e2e --steps 5 --warmup 2 > gpurun_out/r3b_new.json 2> gpurun_out/r3b_new.err; echo "rc $?"; cat gpurun_out/r3b_new.json | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], {k:v for k,v in j['kernel_ms'].items() if v>0.01}, j['roofline']['frac'], j['config']['flagged_bursts'], j['config']['events_per_gpu'])"
RTFE_PEAK_PATH=1 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "golden_tapes or fresh_nrzi or large_tape or peak_record" > gpurun_out/r3i_tests2.log 2>&1; echo "peak-path tests rc $?"; tail -3 gpurun_out/r3i_tests2.log
