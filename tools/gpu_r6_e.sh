#!/bin/bash
# GPU box, round 6 run E: the whole -m gpu suite on the final k_sift_s; rounds per wave and tile; C2 overlapped against resident workgroups; C4 in one launch.
mkdir -p gpurun_out/r06e
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r06e/pytest_gpu.txt
timeout 300 python tools/gpu_sift_phase.py 1e8 2>&1 | tail -5 | tee gpurun_out/r06e/sift_rounds.txt
one() { local label=$1; shift
   env "$@" timeout 900 python bench.py --no-cpu-baseline --no-e2e --no-other-configs $EXTRA > gpurun_out/r06e/$label.json 2> gpurun_out/r06e/$label.err
   python - <<PY
import json
try:
    j = json.loads(open("gpurun_out/r06e/$label.json").read().strip().splitlines()[-1])
    print("$label value", j["value"], "ms", j["ms_per_step"], "serial", j["ms_per_step_serial"], "frac", j["roofline"]["frac"], "launches", j["config"]["launches_per_step"], "flagged", j["config"]["flagged_bursts"], {k: v for k, v in j["kernel_ms"].items() if v > 0.02})
except Exception as e:
    print("$label FAILED", e); print(open("gpurun_out/r06e/$label.err").read()[-1200:])
PY
}
EXTRA="--steps 20 --warmup 5" one c2_wgs6 A=1
EXTRA="--steps 20 --warmup 5" one c2_wgs5 RTFE_SIFT_WGS=5
EXTRA="--steps 20 --warmup 5" one c2_wgs4 RTFE_SIFT_WGS=4
EXTRA="--config C4 --steps 1 --warmup 1" one c4_three A=1
EXTRA="--config C4 --steps 1 --warmup 1 --window-rows 1.1e9" one c4_one_cap09 RT_BENCH_EVENT_CAP=0.09
EXTRA="--config C4 --steps 1 --warmup 1 --window-rows 5.4e8" one c4_two_cap09 RT_BENCH_EVENT_CAP=0.09
