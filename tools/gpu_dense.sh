#!/bin/bash
mkdir -p gpurun_out
for c in G1 P1 C4 C3 C2 C5; do timeout 300 python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-other-configs > gpurun_out/bench_$c.json 2>gpurun_out/bench_$c.err; echo "$c rc $?"; done
python - <<'PY'
import json
for f in ("bench_G1", "bench_P1", "bench_C4", "bench_C3", "bench_C2", "bench_C5"):
    try: j = json.loads(open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "unreadable", e); continue
    print(f, {k: j[k] for k in ("value", "ms_per_step", "timed_steps")}, {k: v for k, v in j["kernel_ms"].items() if v > 0.05})
PY
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
