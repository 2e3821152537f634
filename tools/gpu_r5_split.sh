#!/bin/bash
# GPU box: k_sift_s with the last head split among the pair waves - parity of the peak path, then C2 / C5 / M8c / N1c timing lines.
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "nrzi or peak or sift" 2>&1 | tail -4
for i in 1 2; do bash tools/gpu_try.sh "A=1" --steps 20 --warmup 5 --no-other-configs --no-overlap; done
bash tools/gpu_try.sh "A=1" --steps 20 --warmup 5 --no-other-configs
for e in "RTFE_SIFT_WGS=4" "RTFE_SIFT_DEFER=0"; do echo "$e:"; bash tools/gpu_try.sh "$e" --steps 20 --warmup 5 --no-other-configs --no-overlap; done
