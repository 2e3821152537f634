#!/bin/bash
# GPU box, round 5 experiment 1: k_sift_s with the lists' copy-out deferred to the next tile step against the round-4 order; phase cuts; SQ counters.
mkdir -p gpurun_out
for d in 0 1 0 1; do echo -n "defer $d: "; bash tools/gpu_try.sh "RTFE_SIFT_DEFER=$d" --steps 20 --warmup 5 --no-other-configs; done
for d in 0 1; do for c in 1 2 3 4 5 6; do echo -n "defer $d cut $c: "; RTFE_DEBUG=0 RTFE_SIFT_DEFER=$d RTFE_CUT=$c RTFE_PEAK_STOP=1 timeout 300 python tools/gpu_sift_phase.py 1e8 2>&1 | grep "^rows" | sed 's/.*k_sift.: \([0-9.]*\).*/k_sift \1 ms/'; done; done
timeout 600 bash tools/gpu_pmc.sh --no-other-configs > gpurun_out/r5_sq_a.txt 2>&1; grep "k_sift_s" gpurun_out/r5_sq_a.txt
timeout 600 bash tools/gpu_pmc2.sh --no-other-configs > gpurun_out/r5_sq_b.txt 2>&1; grep "k_sift_s" gpurun_out/r5_sq_b.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3
