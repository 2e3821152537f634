#!/bin/bash
mkdir -p gpurun_out
for dp in 1 0; do
export RTFE_DENSE_PATH=$dp
echo "==== RTFE_DENSE_PATH=$dp"
echo "== gcr 8 sets (C4)"; PROBE_COPIES=14 timeout -s INT 100 python -X faulthandler tools/gpu_dense_probe.py 5e6 8 2>&1 | grep -v amdgpu.ids | tail -3 | head -2
echo "== gcr 1 set"; PROBE_COPIES=14 timeout -s INT 100 python -X faulthandler tools/gpu_dense_probe.py 5e6 1 2>&1 | grep -v amdgpu.ids | tail -3 | head -2
echo "== gcr 5 default sets"; PROBE_COPIES=14 timeout -s INT 100 python -X faulthandler tools/gpu_dense_probe.py 5e6 5 2>&1 | grep -v amdgpu.ids | tail -3 | head -2
echo "== pe 1 set"; PROBE_COPIES=14 timeout -s INT 100 python -X faulthandler tools/gpu_dense_probe.py 5e6 1 pe 2>&1 | grep -v amdgpu.ids | tail -3 | head -2
echo "== pe 8 default sets"; PROBE_COPIES=14 timeout -s INT 100 python -X faulthandler tools/gpu_dense_probe.py 5e6 8 pe 2>&1 | grep -v amdgpu.ids | tail -3 | head -2
echo "== gcr 1 set, 56 copies"; PROBE_COPIES=56 timeout -s INT 100 python -X faulthandler tools/gpu_dense_probe.py 5e6 1 2>&1 | grep -v amdgpu.ids | tail -3 | head -2
done
