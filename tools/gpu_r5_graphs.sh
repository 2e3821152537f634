#!/bin/bash
# GPU box: rtfe_set_graphs against direct launches
mkdir -p gpurun_out
one() { local label=$1; shift
   timeout 600 python bench.py "$@" --no-cpu-baseline --no-e2e --no-other-configs > gpurun_out/gr_$label.json 2> gpurun_out/gr_$label.err
   python - <<PY
import json
try:
    j = json.loads(open("gpurun_out/gr_$label.json").read().strip().splitlines()[-1])
    print("$label ms", j["ms_per_step"], "serial", j["ms_per_step_serial"], "value", j["value"], "graphs", bool(j.get("graphs")))
except Exception as e:
    print("$label FAILED", e); print(open("gpurun_out/gr_$label.err").read()[-1500:])
PY
}
one c2 --steps 20 --warmup 5
one c2_g --steps 20 --warmup 5 --graphs
one c2_serial --steps 20 --warmup 5 --no-overlap
one c2_serial_g --steps 20 --warmup 5 --no-overlap --graphs
one c5s --config C5 --rows 6.9e7 --steps 20 --warmup 5
one c5s_g --config C5 --rows 6.9e7 --steps 20 --warmup 5 --graphs
one g1 --config G1 --steps 3 --warmup 1 --min-seconds 1
one g1_g --config G1 --steps 3 --warmup 1 --min-seconds 1 --graphs
one c3_g --config C3 --steps 8 --warmup 2 --graphs
one m8_g --config M8 --steps 5 --warmup 2 --graphs


