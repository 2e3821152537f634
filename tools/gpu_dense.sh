#!/bin/bash
mkdir -p gpurun_out
export PROBE_COPIES=200
echo "== RTFE_DEBUG=8 gcr"; RTFE_DEBUG=8 timeout 300 python tools/gpu_dense_probe.py 5e6 1 gcr 2>&1 | tail -2
echo "== RTFE_DEBUG=8 pe"; RTFE_DEBUG=8 timeout 300 python tools/gpu_dense_probe.py 5e6 1 pe 2>&1 | tail -2
