#!/bin/bash
mkdir -p gpurun_out/r06i
one() { local label=$1; shift
   env "$@" timeout 900 python bench.py --no-cpu-baseline --no-e2e --no-other-configs $EXTRA > gpurun_out/r06i/$label.json 2> gpurun_out/r06i/$label.err
   python - <<PY
import json
try:
    j = json.loads(open("gpurun_out/r06i/$label.json").read().strip().splitlines()[-1])
    print("$label value", j["value"], "ms", j["ms_per_step"], "serial", j["ms_per_step_serial"], "frac", j["roofline"]["frac"], "flagged", j["config"]["flagged_bursts"], "events", j["config"]["events_total"], {k: v for k, v in j["kernel_ms"].items() if v > 0.02})
except Exception as e:
    print("$label FAILED", e); print(open("gpurun_out/r06i/$label.err").read()[-600:])
PY
}
EXTRA="--config M8 --steps 5 --warmup 2" one m8 A=1
EXTRA="--config M8 --steps 5 --warmup 2" one m8_nodedup RTFE_DENSE_DEDUP=0
EXTRA="--config M8f --steps 5 --warmup 2" one m8f A=1
EXTRA="--steps 20 --warmup 5" one c2 A=1
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
