#!/bin/bash
# round 6, run ES: k_emit_seg - rounds loaded ahead against resident waves (variant libraries)
mkdir -p gpurun_out/r06es2
one() { local label=$1; shift
   env "$@" timeout 900 python bench.py --no-cpu-baseline --no-e2e --no-other-configs $EXTRA > gpurun_out/r06es2/$label.json 2> gpurun_out/r06es2/$label.err
   python -c "
import json; j=json.loads(open('gpurun_out/r06es2/$label.json').read().strip().splitlines()[-1]); print('$label', j['value'], j['ms_per_step'], j['ms_per_step_serial'], {k: v for k, v in j['kernel_ms'].items() if v > 0.02})"
}
for rep in 1 2; do
for v in base es_a2w7 es_a1w8; do
  if [ $v = base ]; then L=; else L="RTFE_LIB_PATH=$PWD/readtape_amd/librtfe_$v.so"; fi
  EXTRA="--steps 20 --warmup 5" one c2_${v}_$rep A=1 $L
  EXTRA="--config M8 --steps 5 --warmup 2" one m8_${v}_$rep A=1 $L
done; done
