#!/bin/bash
# round 6, run X: k_prep's workgroups per CU now that it holds seven waves a SIMD again; the segments' warm-up
mkdir -p gpurun_out/r06x
one() { local label=$1; shift
   env "$@" timeout 600 python bench.py --no-cpu-baseline --no-e2e --no-other-configs --steps 20 --warmup 5 > gpurun_out/r06x/$label.json 2> gpurun_out/r06x/$label.err
   python -c "
import json; j=json.loads(open('gpurun_out/r06x/$label.json').read().strip().splitlines()[-1]); print('$label', j['value'], j['ms_per_step'], j['ms_per_step_serial'], {k: v for k, v in j['kernel_ms'].items() if v > 0.02})"
}
one base A=1
one prep16 RTFE_PREP_WGS=16
one prep24 RTFE_PREP_WGS=24
one prep48 RTFE_PREP_WGS=48
one prep64 RTFE_PREP_WGS=64
one base2 A=1
one warm32 RTFE_SEG_WARM=32
one warm48 RTFE_SEG_WARM=48
one seg512 RTFE_SEG_RECS=512
