#!/bin/bash
# GPU box: rocprofv3 kernel summaries and PMC traffic for C4 and C5 (round 3 had none on file), refreshed for C2.
mkdir -p gpurun_out
timeout 600 bash tools/gpu_profile.sh r04_c4 --config C4 --steps 2 --warmup 1 > gpurun_out/profile_c4.log 2>&1; echo "profile C4 rc $?"; head -14 gpurun_out/profile_c4.log
timeout 600 bash tools/gpu_profile.sh r04_c5 --config C5 --steps 5 --warmup 2 > gpurun_out/profile_c5.log 2>&1; echo "profile C5 rc $?"; head -24 gpurun_out/profile_c5.log
timeout 900 bash tools/gpu_traffic.sh r04 C4 > gpurun_out/traffic_c4.log 2>&1; echo "traffic C4 rc $?"; tail -5 gpurun_out/traffic_c4.log
timeout 900 bash tools/gpu_traffic.sh r04 C5 > gpurun_out/traffic_c5.log 2>&1; echo "traffic C5 rc $?"; tail -5 gpurun_out/traffic_c5.log
