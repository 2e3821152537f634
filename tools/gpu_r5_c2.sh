#!/bin/bash
# GPU box: quick C2 (and optional extra configs) timing lines.  usage: tools/gpu_r5_c2.sh [cfg ...]
for i in 1 2; do bash tools/gpu_try.sh "A=1" --steps 20 --warmup 5 --no-other-configs; done
for c in "$@"; do bash tools/gpu_try.sh "A=1" --config $c --steps 5 --warmup 2; done
