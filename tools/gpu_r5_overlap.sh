#!/bin/bash
# GPU box: C2 with two / three scan contexts while k_sift_s leaves room on the CUs for the other context's kernels
mkdir -p gpurun_out
one() { local label=$1; shift
   env "$@" timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-other-configs > gpurun_out/ov_$label.json 2> gpurun_out/ov_$label.err
   python - <<PY
import json
try:
    j = json.loads(open("gpurun_out/ov_$label.json").read().strip().splitlines()[-1])
    print("$label ms", j["ms_per_step"], "serial", j["ms_per_step_serial"], "sift", j["kernel_ms"]["k_sift"], "prep", j["kernel_ms"]["k_prep"], "emit", j["kernel_ms"]["k_emit"])
except Exception as e:
    print("$label FAILED", e); print(open("gpurun_out/ov_$label.err").read()[-800:])
PY
}
one base A=1
one wgs4 RTFE_SIFT_WGS=4
one wgs3 RTFE_SIFT_WGS=3
one wgs4_c3 RTFE_SIFT_WGS=4 RT_BENCH_CONTEXTS=3
one wgs3_c3 RTFE_SIFT_WGS=3 RT_BENCH_CONTEXTS=3
one base_c3 RT_BENCH_CONTEXTS=3
one wgs2_c3 RTFE_SIFT_WGS=2 RT_BENCH_CONTEXTS=3
