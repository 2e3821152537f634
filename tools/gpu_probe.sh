#!/bin/bash
mkdir -p gpurun_out
for pp in 0 1; do echo "RTFE_PEAK_PATH=$pp"; RTFE_PEAK_PATH=$pp timeout 500 python tools/gpu_configs.py 2>&1 | grep -E "^\{" | python3 -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print(j['config'], '| rows', j['rows'], '| ms', j['ms_per_scan'], '| Msamples/s', j['Msamples_per_s'], '| flagged', j['flagged'], '|', {k: v for k, v in j['kernel_ms'].items() if v > 0.05})
"; done
