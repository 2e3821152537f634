#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; echo "gpu tests rc $?"; tail -3 gpurun_out/gpu_tests.log
run() { PROBE_COPIES=$1 timeout -s INT 200 python -X faulthandler tools/gpu_dense_probe.py 5e6 $2 $3 2>&1 | grep -E "^scan 2|k_dchain|^bursts" | tr '\n' ' '; }
echo -n "gcr 1 set copies 208: "; run 208 1 gcr; echo
timeout 600 python bench.py --config C4 --no-cpu-baseline --no-e2e 2> /dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C4:', j['value'], j['ms_per_step'], {k:v for k,v in j['kernel_ms'].items() if v>1}, j['config']['last_scan_stats'])"
bash tools/gpu_stress.sh 960 4 60
