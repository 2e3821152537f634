#!/bin/bash
# GPU box: k_screen time with phases cut off (RTFE_DEBUG bits 4/8/16/32: stop after load / screen / starts / list)
for d in 5 9 17 33 1 0; do
  RTFE_DEBUG=$d python tools/gpu_phase.py 1e8 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('debug=$d', {k: round(v,2) for k,v in d['kernel_ms'].items()})"
done
