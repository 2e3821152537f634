#!/bin/bash
for f in 1 0; do
  echo "== RTFE_GAIN_FAST=$f"
  RTFE_GAIN_FAST=$f python tools/gpu_gcr_diff.py 2>&1 | grep "differing lists" | cut -c1-150
done
