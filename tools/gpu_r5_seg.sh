#!/bin/bash
# GPU box: the segments' length (RTFE_SEG_RECS) on a short scan (a rank's share of C5 at N = 8), on C5, on C2 and on M8
mkdir -p gpurun_out
for recs in 256 128 64; do
for spec in "c5s:--config C5 --rows 6.9e7 --steps 20 --warmup 5" "c5:--config C5 --steps 10 --warmup 3" "c2:--steps 20 --warmup 5" "m8:--config M8 --steps 5 --warmup 2"; do
  tag=${spec%%:*}; args=${spec#*:}
  RTFE_SEG_RECS=$recs timeout 600 python bench.py $args --no-cpu-baseline --no-e2e --no-other-configs > gpurun_out/sg_${tag}_$recs.json 2> gpurun_out/sg_${tag}_$recs.err
  python - <<PY
import json
try:
    j = json.loads(open("gpurun_out/sg_${tag}_$recs.json").read().strip().splitlines()[-1])
    print("recs $recs $tag ms", j["ms_per_step"], "serial", j["ms_per_step_serial"], {k: v for k, v in j["kernel_ms"].items() if k in ("k_gain", "k_gain_s", "k_gain_tail", "k_emit")})
except Exception as e:
    print("$tag $recs FAILED", e); print(open("gpurun_out/sg_${tag}_$recs.err").read()[-500:])
PY
done; done
