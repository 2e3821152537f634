#!/bin/bash
# scratch: default line with M8, full C4 / C3 lines, the NRZI -m path sweep, G1 / P1 kernel breakdowns
mkdir -p gpurun_out
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc $?"
timeout 700 python bench.py --config C4 > gpurun_out/bench_c4.json 2> gpurun_out/bench_c4_full.err; echo "C4 full line rc $?"; tail -3 gpurun_out/bench_c4_full.err
timeout 600 python bench.py --config C3 > gpurun_out/bench_c3.json 2> gpurun_out/bench_c3_full.err; echo "C3 full line rc $?"
for c in G1 P1 M8; do timeout 300 python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline --no-e2e > gpurun_out/bench_$c.json 2>/dev/null; echo "$c rc $?"; done
python - <<'PY'
import json
for f in ("bench_default", "bench_c4", "bench_c3", "bench_G1", "bench_P1", "bench_M8"):
    try: j = json.loads(open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "unreadable", e); continue
    print(f, {k: j[k] for k in ("value", "ms_per_step", "timed_steps", "timed_seconds")}, j["roofline"]["kernel"], j["roofline"]["frac"], j["roofline"]["whole_step"]["frac"], {k: v for k, v in j["kernel_ms"].items() if v > 0.05})
    for k, v in j.get("other_configs", {}).items(): print("   ", k, {kk: v.get(kk) for kk in ("value", "ms_per_step", "dominant_kernel", "dominant_kernel_ms", "frac", "whole_step_frac", "error")})
    if "e2e" in j: print("    e2e", {k: j.get("e2e", {}).get(k) for k in ("value", "seconds", "tap_identical_to_cpu_port", "error")}, "cpu", {k: j.get("cpu_baseline", {}).get(k) for k in ("value", "kind", "tap_identical_to_reference")})
PY
timeout 600 python tools/gpu_sweep_paths.py 2>&1 | grep "^sets" > gpurun_out/sweep_paths.txt; cat gpurun_out/sweep_paths.txt
