#!/bin/bash
mkdir -p gpurun_out
run() { PROBE_COPIES=$1 timeout -s INT 200 python -X faulthandler tools/gpu_dense_probe.py 5e6 $2 $3 2>&1 | grep -E "^rows|^scan 2|k_dchain" | tr '\n' ' '; }
echo -n "gcr 1 set 1e9: "; run 208 1 gcr; echo
echo -n "pe 1 set 1e9: "; run 208 1 pe; echo
echo -n "gcr 5 default 1e9: "; run 208 5 gcr; echo
echo -n "old gcr 1 set 1e9: "; RTFE_DENSE_PATH=0 run 208 1 gcr; echo
echo -n "old pe 1 set 1e9: "; RTFE_DENSE_PATH=0 run 208 1 pe; echo
timeout 600 python bench.py --config C4 --no-cpu-baseline --no-e2e 2> /dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C4:', j['value'], j['ms_per_step'], {k:v for k,v in j['kernel_ms'].items() if v>1}, j['config']['launches_per_step'], j['roofline']['frac'], j['roofline']['whole_step']['frac'])"
