#!/bin/bash
mkdir -p gpurun_out
for g in 4 5 6 8 16; do
for c in G1 P1 C4; do RTFE_DCHAIN_WGS=$g timeout 300 python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-other-configs > gpurun_out/bench_$c.json 2>gpurun_out/bench_$c.err
python - <<PY
import json
j = json.loads(open("gpurun_out/bench_$c.json").read().strip().splitlines()[-1])
print("wgs $g $c", j["ms_per_step"], j["kernel_ms"]["k_dchain"])
PY
done; done
