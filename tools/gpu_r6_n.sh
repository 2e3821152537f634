#!/bin/bash
# round 6, run N: chains a wave of k_gain's tails (64 / 16 / 8), segment length (256 / 128)
mkdir -p gpurun_out/r06n
one() { local label=$1; shift
   env "$@" timeout 900 python bench.py --no-cpu-baseline --no-e2e --no-other-configs $EXTRA > gpurun_out/r06n/$label.json 2> gpurun_out/r06n/$label.err
   python - <<PY
import json
try:
    j = json.loads(open("gpurun_out/r06n/$label.json").read().strip().splitlines()[-1])
    print("$label value", j["value"], "ms", j["ms_per_step"], "serial", j["ms_per_step_serial"], "frac", j["roofline"]["frac"], "flagged", j["config"]["flagged_bursts"], "events", j["config"]["events_total"], "seq", j["config"]["last_scan_stats"]["sequential"], {k: v for k, v in j["kernel_ms"].items() if v > 0.02}, j["config"]["last_scan_stats"]["phase_cycles"][:6])
except Exception as e:
    print("$label FAILED", e); print(open("gpurun_out/r06n/$label.err").read()[-600:])
PY
}
EXTRA="--steps 20 --warmup 5" one c2 A=1
EXTRA="--config N1 --steps 5 --warmup 2" one n1_16 A=1
EXTRA="--config N1 --steps 5 --warmup 2" one n1_64 RTFE_TAIL_LANES=64
EXTRA="--config N1 --steps 5 --warmup 2" one n1_8 RTFE_TAIL_LANES=8
EXTRA="--config N1 --steps 3 --warmup 1" one n1_16_dbg4 RTFE_DEBUG=4
EXTRA="--config N1 --steps 5 --warmup 2" one n1_16_s128 RTFE_SEG_RECS=128
EXTRA="--config N1 --steps 5 --warmup 2" one n1_8_s128 RTFE_SEG_RECS=128 RTFE_TAIL_LANES=8
EXTRA="--steps 20 --warmup 5" one c2_s128 RTFE_SEG_RECS=128
EXTRA="--config M8 --steps 5 --warmup 2" one m8 A=1
EXTRA="--config M8 --steps 5 --warmup 2" one m8_64 RTFE_TAIL_LANES=64
EXTRA="--config N1f --steps 5 --warmup 2" one n1f A=1
timeout 600 python tools/fuzz_shapes.py --gpu 6000 100 > gpurun_out/r06n/fuzz.log 2>&1; echo "fuzz rc $? ok $(grep -c '^ok' gpurun_out/r06n/fuzz.log) fail $(grep -c '^FAIL ' gpurun_out/r06n/fuzz.log)"
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
