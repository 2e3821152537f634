#!/bin/bash
cp readtape_amd/librtfe.so /tmp/librtfe_default.so
for spec in "$@"; do
   lib=${spec%%:*}; rest=${spec#*:}; np=${rest%%:*}; envs=${rest#*:}; [ "$envs" == "$rest" ] && envs="A=1"
   cp readtape_amd/variants/librtfe_$lib.so readtape_amd/librtfe.so
   echo "== $lib nparm $np $envs"
   env $(echo $envs | tr ',' ' ') timeout 300 python tools/gpu_dseg_phase.py 2e8 $np gcr 2>&1 | tail -1
done
cp /tmp/librtfe_default.so readtape_amd/librtfe.so
