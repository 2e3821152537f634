#!/bin/bash
# round 6, run Q: k_sift_hard without its tables
mkdir -p gpurun_out/r06q
one() { local label=$1; shift
   env "$@" timeout 900 python bench.py --no-cpu-baseline --no-e2e --no-other-configs $EXTRA > gpurun_out/r06q/$label.json 2> gpurun_out/r06q/$label.err
   python - <<PY
import json
try:
    j = json.loads(open("gpurun_out/r06q/$label.json").read().strip().splitlines()[-1])
    print("$label value", j["value"], "ms", j["ms_per_step"], "serial", j["ms_per_step_serial"], "frac", j["roofline"]["frac"], "flagged", j["config"]["flagged_bursts"], "events", j["config"]["events_total"], "seq", j["config"]["last_scan_stats"]["sequential"], {k: v for k, v in j["kernel_ms"].items() if v > 0.02}, j["config"]["last_scan_stats"]["phase_cycles"][:6])
except Exception as e:
    print("$label FAILED", e); print(open("gpurun_out/r06q/$label.err").read()[-600:])
PY
}
EXTRA="--config N1 --steps 5 --warmup 2" one n1 A=1
EXTRA="--config N1 --steps 3 --warmup 1" one n1_dbg9 RTFE_DEBUG=9
EXTRA="--steps 20 --warmup 5" one c2 A=1
EXTRA="--config M8 --steps 5 --warmup 2" one m8 A=1
EXTRA="--config N1f --steps 5 --warmup 2" one n1f A=1
timeout 600 python tools/fuzz_shapes.py --gpu 9000 100 > gpurun_out/r06q/fuzz.log 2>&1; echo "fuzz rc $? ok $(grep -c '^ok' gpurun_out/r06q/fuzz.log) fail $(grep -c '^FAIL ' gpurun_out/r06q/fuzz.log)"
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
