#!/bin/bash
# GPU box: rocprofv3 kernel summaries and PMC traffic for C4 (dense path) and C2, C5 refreshed; single-set GCR / PE lines.
mkdir -p gpurun_out
timeout 600 bash tools/gpu_profile.sh r04_c4 --config C4 --steps 2 --warmup 1 > gpurun_out/profile_c4.log 2>&1; echo "profile C4 rc $?"; head -12 gpurun_out/profile_c4.log
timeout 600 bash tools/gpu_profile.sh r04 --steps 20 --warmup 5 --no-other-configs > gpurun_out/profile_c2.log 2>&1; echo "profile C2 rc $?"; head -22 gpurun_out/profile_c2.log
timeout 900 bash tools/gpu_traffic.sh r04 C4 > gpurun_out/traffic_c4.log 2>&1; echo "traffic C4 rc $?"; tail -3 gpurun_out/traffic_c4.log
timeout 900 bash tools/gpu_traffic.sh r04 C2 > gpurun_out/traffic_c2.log 2>&1; echo "traffic C2 rc $?"; tail -3 gpurun_out/traffic_c2.log
