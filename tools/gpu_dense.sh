#!/bin/bash
mkdir -p gpurun_out
for lo in 0.3333 0.45 0.6; do for hi in 1.25 1.1; do for c in G1 P1; do RTFE_DS_BAND_LO=$lo RTFE_DS_BAND_HI=$hi timeout 300 python bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-other-configs > gpurun_out/bench_$c.json 2>gpurun_out/bench_$c.err
python - <<PY
import json
j = json.loads(open("gpurun_out/bench_$c.json").read().strip().splitlines()[-1])
st = j["config"].get("last_scan_stats", {})
print("band $lo $hi $c", j["ms_per_step"], j["kernel_ms"]["k_dseg"], j["kernel_ms"]["k_dchain"], "lit", st.get("parallel"), "rec", st.get("sequential"), st.get("gave_up", [])[:3])
PY
done; done; done
