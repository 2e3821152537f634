#!/bin/bash
# scratch: the C4 full line with its stderr, then the default line
mkdir -p gpurun_out
timeout 700 python bench.py --config C4 > gpurun_out/bench_c4.json 2> gpurun_out/bench_c4_full.err; echo "C4 full line rc $?"; tail -5 gpurun_out/bench_c4_full.err
timeout 1500 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc $?"; tail -3 gpurun_out/bench_default.err
python - <<'PY'
import json
for f in ("gpurun_out/bench_c4.json", "gpurun_out/bench_default.json"):
    try: j = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "unreadable", e); continue
    print(f, {k: j[k] for k in ("value", "ms_per_step", "timed_steps", "timed_seconds")}, j["roofline"]["kernel"], j["roofline"]["frac"], j["roofline"]["whole_step"]["frac"], j["kernel_ms"])
    for k, v in j.get("other_configs", {}).items(): print(k, {kk: v.get(kk) for kk in ("value", "ms_per_step", "dominant_kernel", "dominant_kernel_ms", "frac", "whole_step_frac", "error")})
    print("e2e", {k: j.get("e2e", {}).get(k) for k in ("value", "seconds", "tap_identical_to_cpu_port", "error")}, "cpu", j.get("cpu_baseline"))
PY
