#!/bin/bash
# GPU box, round 6 run B: where k_sift_s's cycles go (the profiling build), its sensitivity to resident workgroups, the first-scan lines with the lean probe.
mkdir -p gpurun_out/r06b
RTFE_LIB_PATH=$PWD/readtape_amd/librtfe_prof.so timeout 300 python tools/gpu_sift_prof.py C2 2>&1 | tail -12 | tee gpurun_out/r06b/sift_prof_c2.txt
one() { local label=$1; shift
   env "$@" timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-other-configs $EXTRA > gpurun_out/r06b/$label.json 2> gpurun_out/r06b/$label.err
   python - <<PY
import json
try:
    j = json.loads(open("gpurun_out/r06b/$label.json").read().strip().splitlines()[-1])
    print("$label ms", j["ms_per_step"], "serial", j["ms_per_step_serial"], "sift", j["kernel_ms"]["k_sift"], "frac", j["roofline"]["frac"], "floor", (j["config"]["last_scan_stats"] or {}).get("screen_floor_used"), "redone", (j["config"]["last_scan_stats"] or {}).get("redone"))
except Exception as e:
    print("$label FAILED", e); print(open("gpurun_out/r06b/$label.err").read()[-800:])
PY
}
EXTRA="" one c2 A=1
EXTRA="" one c2_wgs4 RTFE_SIFT_WGS=4
EXTRA="" one c2_wgs3 RTFE_SIFT_WGS=3
EXTRA="--config C2f" one c2f A=1
EXTRA="--config M8f" one m8f A=1
EXTRA="--config N1f" one n1f A=1
EXTRA="--config M8" one m8 A=1
timeout 900 python -m pytest tests/test_gpu_ingest.py tests/test_gpu_parity.py -x -q -k "ingest or floor or noisy or golden_tapes" 2>&1 | tail -3
