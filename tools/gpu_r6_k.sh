#!/bin/bash
# round 6, run K: short explicit records fire on the lean step (kCrWeak + kCrClear); k_prep at seven waves a SIMD against the plain build
mkdir -p gpurun_out/r06k
one() { local label=$1; shift
   env "$@" timeout 900 python bench.py --no-cpu-baseline --no-e2e --no-other-configs $EXTRA > gpurun_out/r06k/$label.json 2> gpurun_out/r06k/$label.err
   python - <<PY
import json
try:
    j = json.loads(open("gpurun_out/r06k/$label.json").read().strip().splitlines()[-1])
    print("$label value", j["value"], "ms", j["ms_per_step"], "serial", j["ms_per_step_serial"], "frac", j["roofline"]["frac"], "flagged", j["config"]["flagged_bursts"], "events", j["config"]["events_total"], "seq", j["config"]["last_scan_stats"]["sequential"], {k: v for k, v in j["kernel_ms"].items() if v > 0.02})
except Exception as e:
    print("$label FAILED", e); print(open("gpurun_out/r06k/$label.err").read()[-600:])
PY
}
EXTRA="--steps 20 --warmup 5" one c2 A=1
EXTRA="--steps 20 --warmup 5" one c2_prep0 RTFE_LIB_PATH=readtape_amd/librtfe_prep0.so
EXTRA="--steps 20 --warmup 5" one c2_b A=1
EXTRA="--steps 20 --warmup 5" one c2_prep0_b RTFE_LIB_PATH=readtape_amd/librtfe_prep0.so
EXTRA="--config N1 --steps 5 --warmup 2" one n1 A=1
EXTRA="--config N1f --steps 5 --warmup 2" one n1f A=1
EXTRA="--config M8 --steps 5 --warmup 2" one m8 A=1
EXTRA="--config C5 --steps 5 --warmup 2" one c5 A=1
for s in 3000 3100 3200 3300; do
  timeout 900 python tools/fuzz_shapes.py --gpu $s 100 > gpurun_out/r06k/fuzz_$s.log 2>&1; echo "fuzz $s rc $? ok $(grep -c '^ok' gpurun_out/r06k/fuzz_$s.log) fail $(grep -c '^FAIL ' gpurun_out/r06k/fuzz_$s.log)"
  grep -A3 '^FAIL ' gpurun_out/r06k/fuzz_$s.log | head -12
done
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 1500 bash tools/gpu_stress.sh 2700 2 60
