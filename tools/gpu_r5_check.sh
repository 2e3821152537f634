#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "graph_replayed or dense" 2>&1 | tail -2
for spec in "c5_n1:--config C5 --steps 10 --warmup 3" "c5_n1_ng:--config C5 --steps 10 --warmup 3 --no-graphs" "c5_s8:--config C5 --rows 6.9e7 --steps 20 --warmup 5" "c5_s8_ng:--config C5 --rows 6.9e7 --steps 20 --warmup 5 --no-graphs" "c5_tiny:--config C5 --rows 2e6 --steps 20 --warmup 5 --no-graphs" "c5_tiny_g:--config C5 --rows 2e6 --steps 20 --warmup 5" "g1:--config G1 --steps 2 --warmup 1 --min-seconds 1" "c4:--config C4 --steps 1 --warmup 1 --min-seconds 1"; do
  tag=${spec%%:*}; args=${spec#*:}
  timeout 600 python bench.py $args --no-cpu-baseline --no-e2e --no-other-configs > gpurun_out/ck_$tag.json 2> gpurun_out/ck_$tag.err
  python - <<PY
import json
try:
    j = json.loads(open("gpurun_out/ck_$tag.json").read().strip().splitlines()[-1])
    print("$tag ms", j["ms_per_step"], "serial", j["ms_per_step_serial"], "graphs", bool(j.get("graphs")), {k: v for k, v in j["kernel_ms"].items() if v > 0.05})
except Exception as e:
    print("$tag FAILED", e); print(open("gpurun_out/ck_$tag.err").read()[-800:])
PY
done
