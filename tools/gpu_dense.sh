#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; echo "gpu tests rc $?"; tail -4 gpurun_out/gpu_tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 1200 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc $?"
python - <<'PY'
import json
j = json.loads(open("gpurun_out/bench_default.json").read().strip().splitlines()[-1])
print({k: j[k] for k in ("value", "ms_per_step", "timed_steps", "timed_seconds")}, j["roofline"]["kernel"], j["roofline"]["frac"], j["roofline"]["whole_step"]["frac"])
for k, v in j.get("other_configs", {}).items(): print(k, {kk: v.get(kk) for kk in ("value", "ms_per_step", "dominant_kernel", "dominant_kernel_ms", "frac", "whole_step_frac", "flagged_bursts", "error")})
print("e2e", {k: j["e2e"].get(k) for k in ("value", "seconds", "tap_identical_to_cpu_port", "error")})
PY
timeout 600 python bench.py --config C4 --no-cpu-baseline > gpurun_out/bench_c4.json 2> gpurun_out/bench_c4.err; echo "C4 rc $?"; tail -c 1500 gpurun_out/bench_c4.json
