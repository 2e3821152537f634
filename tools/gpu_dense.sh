#!/bin/bash
mkdir -p gpurun_out
echo "== copies 14, 8 sets, dchain phases"; RTFE_DEBUG=8 PROBE_COPIES=14 timeout -s INT 100 python -X faulthandler tools/gpu_dense_probe.py 5e6 8 2>&1 | grep -v amdgpu.ids | tail -3
echo "== copies 14, 1 set"; RTFE_DEBUG=8 PROBE_COPIES=14 timeout -s INT 100 python -X faulthandler tools/gpu_dense_probe.py 5e6 1 2>&1 | grep -v amdgpu.ids | tail -3
