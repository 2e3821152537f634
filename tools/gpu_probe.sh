#!/bin/bash
mkdir -p gpurun_out
for c in nrzi9 nrzi9_m pe pe_m gcr gcr_m nrzi9_skew nrzi9_invert nrzi7_order nrzi9_nobpi nrzi9_deskew_long; do
  timeout 120 python tools/gpu_cmp_paths.py $c 2>&1 | grep -v amdgpu.ids | tail -2
done
