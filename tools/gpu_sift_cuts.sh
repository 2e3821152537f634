#!/bin/bash
# GPU box: k_sift's time with its phases cut off one after the other (RTFE_CUT) and the per-phase cycle counters (RTFE_DEBUG=3)
for c in 0 1 2 3 4 5 6; do echo -n "cut $c: "; RTFE_CUT=$c RTFE_PEAK_STOP=1 timeout 300 python tools/gpu_sift_phase.py 1e8 2>&1 | grep "^rows" | sed 's/.*k_sift.: \([0-9.]*\).*/k_sift \1 ms/'; done
RTFE_DEBUG=3 timeout 300 python tools/gpu_sift_phase.py 1e8 2>&1 | grep "wave-cycles\|record bytes"
