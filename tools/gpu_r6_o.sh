#!/bin/bash
# round 6, run O: the stress sweep with the fuzzer's shapes over 60 % of its tapes (every option, every path), the shape fuzzer, C2 / N1 on the final code
mkdir -p gpurun_out/r06o
export STRESS_SHAPES=1
timeout 2400 bash tools/gpu_stress.sh 2800 4 60
unset STRESS_SHAPES
for s in 8000 8100; do
  timeout 900 python tools/fuzz_shapes.py --gpu $s 100 > gpurun_out/r06o/fuzz_$s.log 2>&1; echo "fuzz $s rc $? ok $(grep -c '^ok' gpurun_out/r06o/fuzz_$s.log) fail $(grep -c '^FAIL ' gpurun_out/r06o/fuzz_$s.log)"
done
python bench.py --no-cpu-baseline --no-e2e --no-other-configs --steps 20 --warmup 5 > gpurun_out/r06o/c2.json 2>/dev/null; python -c "
import json; j=json.loads(open('gpurun_out/r06o/c2.json').read().strip().splitlines()[-1]); print('c2', j['value'], j['ms_per_step'], j['roofline']['frac'], j['kernel_ms'])"
python bench.py --no-cpu-baseline --no-e2e --no-other-configs --config N1 --steps 5 --warmup 2 > gpurun_out/r06o/n1.json 2>/dev/null; python -c "
import json; j=json.loads(open('gpurun_out/r06o/n1.json').read().strip().splitlines()[-1]); print('n1', j['value'], j['ms_per_step'], j['kernel_ms'])"
