#!/bin/bash
# GPU box: SQ counters of k_sift cut down phase by phase (RTFE_CUT): dynamic instruction counts per phase by difference
export RTFE_PEAK_PATH=1 RTFE_PEAK_STOP=1
for c in 1 2 3 4 0; do
  echo "=== RTFE_CUT=$c"
  RTFE_CUT=$c bash tools/gpu_pmc.sh 2>&1 | grep "k_sift" | awk '{print $3, $4, $5, $6}' | tr '\n' ';'
  echo
done
