#!/bin/bash
# GPU box: k_screen time with phases cut off (RTFE_CUT: stop after 1 load / 2 screen / 3 stale minima / 4 run starts / 5 run table+scan)
for d in 1 2 3 4 5 0; do
  RTFE_CUT=$d RTFE_DEBUG=0 python tools/gpu_phase.py 1e8 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cut=$d k_screen ms', round(d['kernel_ms']['k_screen'],3))"
done
