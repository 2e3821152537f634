#!/bin/bash
# GPU box, round 6 run F: C4 in one launch (arena sized by a bursts hint), three scan contexts for the overlapped lines, SQ counters after the diet
mkdir -p gpurun_out/r06f
one() { local label=$1; shift
   env "$@" timeout 900 python bench.py --no-cpu-baseline --no-e2e --no-other-configs $EXTRA > gpurun_out/r06f/$label.json 2> gpurun_out/r06f/$label.err
   python - <<PY
import json
try:
    j = json.loads(open("gpurun_out/r06f/$label.json").read().strip().splitlines()[-1])
    print("$label value", j["value"], "ms", j["ms_per_step"], "serial", j["ms_per_step_serial"], "frac", j["roofline"]["frac"], "launches", j["config"]["launches_per_step"], "flagged", j["config"]["flagged_bursts"], {k: v for k, v in j["kernel_ms"].items() if v > 0.02})
except Exception as e:
    print("$label FAILED", e); print(open("gpurun_out/r06f/$label.err").read()[-600:])
PY
}
EXTRA="--config C4 --steps 1 --warmup 1 --window-rows 1.1e9" one c4_one RT_BENCH_BURSTS_PER_ROW=1e-4
EXTRA="--steps 20 --warmup 5" one c2_ctx2 A=1
EXTRA="--steps 20 --warmup 5" one c2_ctx3 RT_BENCH_CONTEXTS=3
EXTRA="--steps 20 --warmup 5" one c2_ctx4 RT_BENCH_CONTEXTS=4
EXTRA="--config M8 --steps 5 --warmup 2" one m8_ctx3 RT_BENCH_CONTEXTS=3
timeout 600 bash tools/gpu_pmc.sh --no-other-configs --no-overlap > gpurun_out/r06f/sq_counters_c2_a.txt 2>&1
grep -h "k_sift_s" gpurun_out/r06f/sq_counters_c2_a.txt | head -20
rm -rf gpurun_out/pmc_sq
