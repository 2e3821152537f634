#!/bin/bash
# GPU box: the dense path's three bench lines (G1, P1, C4) and its parity tests
for c in G1 P1 C4; do bash tools/gpu_try.sh "A=1" --config $c --steps 2 --warmup 1; done
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "dense" 2>&1 | tail -2
