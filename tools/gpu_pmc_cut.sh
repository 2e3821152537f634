#!/bin/bash
# GPU box: VALU/SALU/LDS instruction counts of k_screen with phases cut off (RTFE_CUT)
export TMPDIR=/tmp
for c in 1 2 3 4 5 0; do
  out=$PWD/gpurun_out/pmc_cut$c; mkdir -p $out
  (cd /tmp && RTFE_CUT=$c rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $out -o sq -- python $GRAFT_REPO_ROOT/tools/gpu_phase.py 2e7 > /dev/null 2> $out/err.log)
  python - <<PY
import sqlite3, glob
for f in sorted(glob.glob("$out/**/*.db", recursive=True)):
    db = sqlite3.connect(f)
    r = {cn: v for kn, cn, n, v in db.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name") if "k_screen" in kn}
    print("cut=$c", {k: f"{v:.3g}" for k, v in r.items()})
PY
done
