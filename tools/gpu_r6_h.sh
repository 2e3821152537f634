#!/bin/bash
mkdir -p gpurun_out/r06h
one() { local label=$1; shift
   env "$@" timeout 900 python bench.py --no-cpu-baseline --no-e2e --no-other-configs $EXTRA > gpurun_out/r06h/$label.json 2> gpurun_out/r06h/$label.err
   python - <<PY
import json
try:
    j = json.loads(open("gpurun_out/r06h/$label.json").read().strip().splitlines()[-1])
    print("$label value", j["value"], "ms", j["ms_per_step"], "serial", j["ms_per_step_serial"], "frac", j["roofline"]["frac"], "flagged", j["config"]["flagged_bursts"], {k: v for k, v in j["kernel_ms"].items() if v > 0.02})
except Exception as e:
    print("$label FAILED", e); print(open("gpurun_out/r06h/$label.err").read()[-600:])
PY
}
EXTRA="--config C3 --steps 4 --warmup 1 --no-overlap" one c3_wps4 A=1
for w in 5 6 8; do EXTRA="--config C3 --steps 4 --warmup 1 --no-overlap" one c3_wps$w RTFE_LIB_PATH=$PWD/readtape_amd/librtfe_zp$w.so; done
timeout 600 bash tools/gpu_pmc.sh --config C3 --no-overlap 2>&1 | grep "k_zeros"
rm -rf gpurun_out/pmc_sq
