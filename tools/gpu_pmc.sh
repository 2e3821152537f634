#!/bin/bash
# GPU box: SQ counters for the bench kernels (one pass, 8 SQ slots)
export TMPDIR=/tmp
out=$PWD/gpurun_out/pmc_sq
mkdir -p $out
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $out -o sq -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e "$@" > /dev/null 2> $out/err.log
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM -d $out/b -o sq -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e "$@" > /dev/null 2>> $out/err.log
cd $GRAFT_REPO_ROOT
python - <<PY
import sqlite3, glob
for f in sorted(glob.glob("$out/**/*.db", recursive=True)):
    db = sqlite3.connect(f)
    for kn, cn, n, v in db.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"):
        if "rtfe" in kn: print(kn.split("(")[0][:20], cn, n, f"{v:.4g}")
PY
tail -3 $out/err.log
