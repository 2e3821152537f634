#!/bin/bash
# round 6, run R: kCrWeak records clear up to their LAST row (their successor - the true minimum one sample on - begins right behind it)
mkdir -p gpurun_out/r06r
one() { local label=$1; shift
   env "$@" timeout 900 python bench.py --no-cpu-baseline --no-e2e --no-other-configs $EXTRA > gpurun_out/r06r/$label.json 2> gpurun_out/r06r/$label.err
   python - <<PY
import json
try:
    j = json.loads(open("gpurun_out/r06r/$label.json").read().strip().splitlines()[-1])
    print("$label value", j["value"], "ms", j["ms_per_step"], "serial", j["ms_per_step_serial"], "frac", j["roofline"]["frac"], "flagged", j["config"]["flagged_bursts"], "events", j["config"]["events_total"], "seq", j["config"]["last_scan_stats"]["sequential"], {k: v for k, v in j["kernel_ms"].items() if v > 0.02})
except Exception as e:
    print("$label FAILED", e); print(open("gpurun_out/r06r/$label.err").read()[-600:])
PY
}
EXTRA="--config N1 --steps 5 --warmup 2" one n1 A=1
EXTRA="--config N1f --steps 5 --warmup 2" one n1f A=1
EXTRA="--steps 20 --warmup 5" one c2 A=1
EXTRA="--config M8 --steps 5 --warmup 2" one m8 A=1
EXTRA="--config C5 --steps 5 --warmup 2" one c5 A=1
for s in 11000 11100 11200; do
  timeout 900 python tools/fuzz_shapes.py --gpu $s 100 > gpurun_out/r06r/fuzz_$s.log 2>&1; echo "fuzz $s rc $? ok $(grep -c '^ok' gpurun_out/r06r/fuzz_$s.log) fail $(grep -c '^FAIL ' gpurun_out/r06r/fuzz_$s.log)"
  grep -A3 '^FAIL ' gpurun_out/r06r/fuzz_$s.log | head -12
done
STRESS_SHAPES=1 timeout 1500 bash tools/gpu_stress.sh 3000 2 60
timeout 1500 bash tools/gpu_stress.sh 3010 1 60
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
