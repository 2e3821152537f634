#!/bin/bash
# GPU box: SQ counters of the peak path's kernels on the C2 line
export RTFE_PEAK_PATH=1
bash tools/gpu_pmc.sh 2>&1 | grep -v "^$" | tail -60
