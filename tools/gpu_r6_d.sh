#!/bin/bash
mkdir -p gpurun_out/r06d
for v in prof2 exp1 exp2 exp3; do echo "== $v"; RTFE_LIB_PATH=$PWD/readtape_amd/librtfe_$v.so timeout 300 python tools/gpu_sift_prof.py C2 2>&1 | tail -9 | tee gpurun_out/r06d/sift_$v.txt; done
