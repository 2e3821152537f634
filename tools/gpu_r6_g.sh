#!/bin/bash
mkdir -p gpurun_out/r06g
one() { local label=$1; shift
   env "$@" timeout 900 python bench.py --no-cpu-baseline --no-e2e --no-other-configs $EXTRA > gpurun_out/r06g/$label.json 2> gpurun_out/r06g/$label.err
   python - <<PY
import json
try:
    j = json.loads(open("gpurun_out/r06g/$label.json").read().strip().splitlines()[-1])
    print("$label value", j["value"], "ms", j["ms_per_step"], "serial", j["ms_per_step_serial"], "frac", j["roofline"]["frac"], "flagged", j["config"]["flagged_bursts"], {k: v for k, v in j["kernel_ms"].items() if v > 0.02})
except Exception as e:
    print("$label FAILED", e); print(open("gpurun_out/r06g/$label.err").read()[-600:])
PY
}
EXTRA="--config N1 --steps 10 --warmup 2" one n1 A=1
EXTRA="--config N1f --steps 5 --warmup 2" one n1f A=1
EXTRA="--config M8 --steps 5 --warmup 2" one m8 A=1
timeout 600 bash tools/gpu_profile.sh r06g_n1 --config N1 --steps 5 --warmup 2 --no-overlap 2>&1 | head -14
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_ingest.py -x -q -k "nrzi or peak or golden or c2 or C2 or nois or rare or floor" 2>&1 | tail -3
