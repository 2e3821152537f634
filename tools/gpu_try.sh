#!/bin/bash
# GPU box: one bench configuration with environment overrides, summary line only.  usage: tools/gpu_try.sh "ENV=.. ENV=.." <bench args>
envs=$1; shift
env $envs timeout 900 python bench.py --no-cpu-baseline --no-e2e "$@" > /tmp/try.json 2> /tmp/try.err || tail -5 /tmp/try.err
python - <<'PY'
import json
try:
    j = json.loads(open("/tmp/try.json").read().strip().splitlines()[-1])
    print(j["config"]["workload"][:40], "|", j["value"], "Msamples/s", j["ms_per_step"], "ms (serial", j.get("ms_per_step_serial"), ")", {k: v for k, v in j["kernel_ms"].items() if v > 0.01}, "frac", j["roofline"]["frac"], j["roofline"]["kernel"], "flagged", j["config"]["flagged_bursts"], "events", j["config"]["events_per_gpu"])
except Exception as e:
    print("no line:", e)
PY
