#!/bin/bash
# GPU box: where M8's k_gain_tail comes from - the segments' warm-up and length
mkdir -p gpurun_out
one() { local label=$1; shift
   env "$@" timeout 600 python bench.py --config M8 --steps 5 --warmup 2 --no-overlap --no-cpu-baseline --no-e2e --no-other-configs --min-seconds 0.3 > gpurun_out/m8_$label.json 2> gpurun_out/m8_$label.err
   python - <<PY
import json
try:
    j = json.loads(open("gpurun_out/m8_$label.json").read().strip().splitlines()[-1])
    st = j["config"]["last_scan_stats"]
    print("$label ms", j["ms_per_step"], {k: v for k, v in j["kernel_ms"].items() if v > 0.05}, "seq", st["sequential"], "par", st["parallel"])
except Exception as e:
    print("$label FAILED", e); print(open("gpurun_out/m8_$label.err").read()[-800:])
PY
}
one base A=1
one warm300 RTFE_SEG_WARM=300
one warm600 RTFE_SEG_WARM=600
one recs128 RTFE_SEG_RECS=128
one recs512 RTFE_SEG_RECS=512
one fast0 RTFE_GAIN_FAST=0
