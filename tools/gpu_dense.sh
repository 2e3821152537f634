#!/bin/bash
mkdir -p gpurun_out
for u in 2 3; do for c in C4 G1 P1; do RTFE_DS_UP=$u timeout 300 python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-other-configs > gpurun_out/bench_$c.json 2>gpurun_out/bench_$c.err
python - <<PY
import json
j = json.loads(open("gpurun_out/bench_$c.json").read().strip().splitlines()[-1])
print("ds_up $u $c", j["ms_per_step"], j["kernel_ms"]["k_dseg"], j["kernel_ms"]["k_dchain"])
PY
done; done
bash tools/gpu_stress.sh 1001 1 100
