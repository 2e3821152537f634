#!/bin/bash
run() { PROBE_COPIES=$1 timeout -s INT 200 python -X faulthandler tools/gpu_dense_probe.py 5e6 $2 $3 2>&1 | grep -E "^scan 2|k_dchain" | tr '\n' ' '; }
echo -n "gcr 1 set copies 208: "; run 208 1 gcr; echo
echo -n "pe 1 set copies 208: "; run 208 1 pe; echo
timeout 600 python bench.py --config C4 --no-cpu-baseline --no-e2e 2> /dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C4:', j['value'], j['ms_per_step'], {k:v for k,v in j['kernel_ms'].items() if v>1})"
