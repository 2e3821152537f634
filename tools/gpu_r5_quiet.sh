#!/bin/bash
mkdir -p gpurun_out
for spec in "c3:A=1:--config C3 --steps 8 --warmup 2" "c3s:A=1:--config C3 --steps 8 --warmup 2 --no-overlap"; do
  tag=${spec%%:*}; rest=${spec#*:}; envs=${rest%%:*}; args=${rest#*:}
  env $envs timeout 600 python bench.py $args --no-cpu-baseline --no-e2e --no-other-configs > gpurun_out/q_$tag.json 2> gpurun_out/q_$tag.err
  python - <<PY
import json
try:
    j = json.loads(open("gpurun_out/q_$tag.json").read().strip().splitlines()[-1])
    print("$tag ms", j["ms_per_step"], "serial", j["ms_per_step_serial"], {k: v for k, v in j["kernel_ms"].items() if v > 0.05}, j["config"]["bursts"], j["config"]["events_per_gpu"])
except Exception as e:
    print("$tag FAILED", e); print(open("gpurun_out/q_$tag.err").read()[-800:])
PY
done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_ingest.py -x -q -k "zeros or c3 or diff or density or nobpi or seam or shards or fragments or ingest or sample_path" 2>&1 | tail -2
