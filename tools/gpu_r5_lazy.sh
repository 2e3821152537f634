#!/bin/bash
# GPU box: k_dseg eager (round 4) against lazy (the sub-segment lanes classify the rows they meet), local candidate thresholds on / off,
# register budgets, tile sizes and workgroup sizes of the lazy kernel.  usage: gpu_r5_lazy.sh "<configs>" <variant:env,env ...> ...
mkdir -p gpurun_out
cp readtape_amd/librtfe.so /tmp/librtfe_default.so
one() {  # label, config, lib, env...
   local label=$1 c=$2 lib=$3; shift 3
   cp readtape_amd/variants/librtfe_$lib.so readtape_amd/librtfe.so
   env "$@" timeout 400 python bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-other-configs --min-seconds 0.5 > gpurun_out/lz_${label}_${c}.json 2> gpurun_out/lz_${label}_${c}.err
   python - <<PY
import json
try:
    j = json.loads(open("gpurun_out/lz_${label}_${c}.json").read().strip().splitlines()[-1])
    st = j["config"].get("last_scan_stats", {})
    print("${label} ${c} ms", j["ms_per_step"], "dseg", j["kernel_ms"].get("k_dseg"), "dchain", j["kernel_ms"].get("k_dchain"), "lit", st.get("parallel"), "rec", st.get("sequential"), "redone", st.get("redone"))
except Exception as e:
    print("${label} ${c} FAILED", e); print(open("gpurun_out/lz_${label}_${c}.err").read()[-800:])
PY
}
configs=$1; shift
for c in $configs; do
   for spec in "$@"; do
      lib=${spec%%:*}; envs=${spec#*:}; [ "$envs" == "$spec" ] && envs=""
      one "${lib}_$(echo $envs | tr ',= ' '___')" $c $lib $(echo $envs | tr ',' ' ')
   done
done
cp /tmp/librtfe_default.so readtape_amd/librtfe.so
