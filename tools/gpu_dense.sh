#!/bin/bash
mkdir -p gpurun_out
for w in "" "RTFE_DS_WARM=60" "RTFE_DS_WARM=48"; do
env $w timeout 600 python bench.py --config C4 --no-cpu-baseline --no-e2e --window-rows 536870912 2> /dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C4 $w:', j['value'], j['ms_per_step'], {k:v for k,v in j['kernel_ms'].items() if v>1}, j['config']['launches_per_step'], j['config']['last_scan_stats'])"
done
