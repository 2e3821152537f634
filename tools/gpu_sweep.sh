#!/bin/bash
# GPU box: parity first, then the bench at a few tile sizes, then the rocprof kernel trace.
mkdir -p gpurun_out
(python -m pytest tests -m gpu -x -q 2>&1 | tail -8) > gpurun_out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
for t in 256 512 1024 2048; do
  RTFE_TILE_ROWS=$t python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_tile$t.json
done
cat gpurun_out/pytest_gpu.log; tail -2 gpurun_out/smoke.log
for t in 256 512 1024 2048; do python - <<PY
import json
j=json.load(open("gpurun_out/bench_tile$t.json"))
print($t, j["value"], j["kernel_ms"], j["roofline"]["frac"])
PY
done
