#!/bin/bash
# GPU box: a second set of SQ counters (occupancy, in-flight memory instructions, branches, lane utilisation); args = bench.py args
export TMPDIR=/tmp
out=$PWD/gpurun_out/pmc_sq2
rm -rf $out; mkdir -p $out
cd /tmp
rocprofv3 --pmc SQ_INSTS_BRANCH SQ_IFETCH SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_THREAD_CYCLES_VALU SQ_LEVEL_WAVES SQ_BUSY_CYCLES -d $out -o sq -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-e2e "$@" > /dev/null 2> $out/err.log
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_SMEM SQ_INST_CYCLES_SMEM SQ_CYCLES -d $out/b -o sq -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-e2e "$@" > /dev/null 2>> $out/err.log
cd $GRAFT_REPO_ROOT
python - <<PY
import sqlite3, glob
for f in sorted(glob.glob("$out/**/*.db", recursive=True)):
    db = sqlite3.connect(f)
    for kn, cn, n, v in db.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"):
        if "rtfe" in kn: print(kn.split("(")[0][:20], cn, n, f"{v:.4g}")
PY
tail -3 $out/err.log
