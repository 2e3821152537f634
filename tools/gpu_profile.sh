#!/bin/bash
# GPU box: rocprofv3 kernel trace + stats of the default bench command, then PMC passes (separate runs).
# usage: tools/gpu_profile.sh <tag>
tag=${1:-r01}
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp
rocprofv3 --kernel-trace --stats -d $out/trace -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-e2e > $out/bench_under_rocprof.json 2> $out/rocprof_stderr.log
rocprofv3 --pmc FETCH_SIZE -d $out/pmc_fetch -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > /dev/null 2>> $out/rocprof_stderr.log
rocprofv3 --pmc WRITE_SIZE -d $out/pmc_write -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > /dev/null 2>> $out/rocprof_stderr.log
cd $GRAFT_REPO_ROOT
find $out -name "*.csv" | head -20
python tools/prof_summary.py $out > $out/summary.txt 2>&1
cat $out/summary.txt
