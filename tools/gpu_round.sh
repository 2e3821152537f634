#!/bin/bash
# GPU box: k_lwalk sweep of segment size / warm-up (C2 bench line without the CPU legs)
mkdir -p gpurun_out
run() { env "$@" timeout 600 python bench.py --steps 5 --warmup 2 --no-e2e --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['kernel_ms']; print('$*', j['value'], 'walk', k['k_walk'], 'resume', k['k_decode_resume'], 'screen', k['k_screen'], 'flagged', j['config']['flagged_bursts'])"; }
run RTFE_LWALK=1
run RTFE_LWALK=0
for st in 8 16 32; do for wm in 2 4 8; do if [ $wm -le $st ]; then run RTFE_SEG_TILES=$st RTFE_SEG_WARMUP=$wm; fi; done; done
RTFE_DEBUG=1 timeout 300 python tools/gpu_segs.py 2>&1 | tail -12
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "segmented or rare_paths or golden_tapes or long_blocks" > gpurun_out/gpu_tests_walk.log 2>&1; echo "tests rc $?"; tail -3 gpurun_out/gpu_tests_walk.log
