#!/bin/bash
mkdir -p gpurun_out
one() { local label=$1; shift
   env "$@" timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-other-configs > gpurun_out/sf_$label.json 2> gpurun_out/sf_$label.err
   python - <<PY
import json
try:
    j = json.loads(open("gpurun_out/sf_$label.json").read().strip().splitlines()[-1])
    print("$label ms", j["ms_per_step"], "serial", j["ms_per_step_serial"], "sift", j["kernel_ms"]["k_sift"], "frac", j["roofline"]["frac"])
except Exception as e:
    print("$label FAILED", e); print(open("gpurun_out/sf_$label.err").read()[-800:])
PY
}
one plain A=1
one general RTFE_SIFT_PLAIN=0
one plain2 A=1
one general2 RTFE_SIFT_PLAIN=0
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "golden_tapes or rare_paths or sample_rates or peak_record" 2>&1 | tail -2
