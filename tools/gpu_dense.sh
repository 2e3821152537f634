#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc $?"; tail -3 gpurun_out/bench_default.err
python - <<'PY'
import json
j = json.loads(open("gpurun_out/bench_default.json").read().strip().splitlines()[-1])
print({k: j[k] for k in ("value", "ms_per_step", "timed_steps", "timed_seconds")}, j["roofline"]["kernel"], j["roofline"]["frac"], j["roofline"]["whole_step"]["frac"])
print({k: round(v, 3) for k, v in j["kernel_ms"].items() if v > 0.003})
for k, v in j.get("other_configs", {}).items(): print(k, v)
print("e2e", {k: j["e2e"].get(k) for k in ("value", "seconds", "tap_identical_to_cpu_port", "host_replay_events_per_s_per_thread", "error")}, "cpu", j.get("cpu_baseline", {}).get("value"))
PY
