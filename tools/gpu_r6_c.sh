#!/bin/bash
mkdir -p gpurun_out/r06c
for v in prof; do RTFE_LIB_PATH=$PWD/readtape_amd/librtfe_$v.so timeout 300 python tools/gpu_sift_prof.py C2 2>&1 | tail -10 | tee gpurun_out/r06c/sift_$v.txt; done
one() { local label=$1; shift
   env "$@" timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-other-configs $EXTRA > gpurun_out/r06c/$label.json 2> gpurun_out/r06c/$label.err
   python - <<PY
import json
try:
    j = json.loads(open("gpurun_out/r06c/$label.json").read().strip().splitlines()[-1])
    print("$label ms", j["ms_per_step"], "serial", j["ms_per_step_serial"], "sift", j["kernel_ms"]["k_sift"], "frac", j["roofline"]["frac"], {k: v for k, v in j["kernel_ms"].items() if v > 0.02})
except Exception as e:
    print("$label FAILED", e); print(open("gpurun_out/r06c/$label.err").read()[-800:])
PY
}
EXTRA="" one c2 A=1
EXTRA="" one c2_b A=1
EXTRA="--config C5" one c5 A=1
EXTRA="--config M8" one m8 A=1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -k "nrzi or peak or golden or c2 or C2" 2>&1 | tail -3
