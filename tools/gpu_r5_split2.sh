#!/bin/bash
# GPU box: k_sift_s with and without the split of the last head (rebuilt on the box), serial and overlapped C2 lines
for v in 0 1; do
  RTFE_EXTRA_HIPFLAGS="-DRTFE_SFS_SPLIT=$v" python -c "from readtape_amd import build; build.build_frontend(force=True)"
  echo "split=$v"
  bash tools/gpu_try.sh "A=1" --steps 20 --warmup 5 --no-other-configs --no-overlap
  bash tools/gpu_try.sh "A=1" --steps 20 --warmup 5 --no-other-configs
  bash tools/gpu_try.sh "A=1" --steps 20 --warmup 5 --no-other-configs
done
