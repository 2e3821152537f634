#!/bin/bash
# GPU box: rtfe_set_graphs against direct launches, with and without the per-kernel events (a stream of the bench's own: the legacy default stream is not capturable)
mkdir -p gpurun_out
one() { local label=$1; shift
   timeout 600 python bench.py "$@" --no-cpu-baseline --no-e2e --no-other-configs > gpurun_out/gr_$label.json 2> gpurun_out/gr_$label.err
   python - <<PY
import json
try:
    j = json.loads(open("gpurun_out/gr_$label.json").read().strip().splitlines()[-1])
    print("$label ms", j["ms_per_step"], "serial(with events)", j["ms_per_step_serial"], "graphs", bool(j.get("graphs")))
except Exception as e:
    print("$label FAILED", e); print(open("gpurun_out/gr_$label.err").read()[-1500:])
PY
}
for cfg in "c2s:--steps 20 --warmup 5 --no-overlap" "c5s8:--config C5 --rows 6.9e7 --steps 20 --warmup 5" "c5n1:--config C5 --steps 10 --warmup 3" "tiny:--config C5 --rows 2e6 --steps 20 --warmup 5"; do
   tag=${cfg%%:*}; args=${cfg#*:}
   one ${tag}_events $args --no-graphs
   one ${tag}_noevents $args --no-graphs --no-kernel-events
   one ${tag}_graphs $args --graphs
done
