#!/bin/bash
mkdir -p gpurun_out
one() { local label=$1; shift
   timeout 600 python bench.py "$@" --no-cpu-baseline --no-e2e --no-other-configs > gpurun_out/ev_$label.json 2> gpurun_out/ev_$label.err
   python - <<PY
import json
try:
    j = json.loads(open("gpurun_out/ev_$label.json").read().strip().splitlines()[-1])
    print("$label ms", j["ms_per_step"], "serial", j["ms_per_step_serial"], "value", j["value"])
except Exception as e:
    print("$label FAILED", e); print(open("gpurun_out/ev_$label.err").read()[-1500:])
PY
}
one c2 --steps 20 --warmup 5
one c2_noev --steps 20 --warmup 5 --no-kernel-events
one c2_b --steps 20 --warmup 5
one c2_noev_b --steps 20 --warmup 5 --no-kernel-events
one m8 --config M8 --steps 5 --warmup 2
one m8_noev --config M8 --steps 5 --warmup 2 --no-kernel-events
one c3 --config C3 --steps 8 --warmup 2
one c3_noev --config C3 --steps 8 --warmup 2 --no-kernel-events
