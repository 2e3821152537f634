#!/bin/bash
# GPU box: rocprofv3 kernel trace + stats of a bench command, then PMC passes (separate runs).
# usage: tools/gpu_profile.sh <tag> [bench.py arguments, default: the driver's C2 line without the CPU legs]
tag=${1:-r01}; shift
args="$*"; [ -z "$args" ] && args="--steps 5 --warmup 2"
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp
rocprofv3 --kernel-trace --stats -d $out/trace -o bench -- python $GRAFT_REPO_ROOT/bench.py $args --no-cpu-baseline --no-e2e > $out/bench_under_rocprof.json 2> $out/rocprof_stderr.log
rocprofv3 --pmc FETCH_SIZE -d $out/pmc_fetch -o bench -- python $GRAFT_REPO_ROOT/bench.py $args --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > /dev/null 2>> $out/rocprof_stderr.log
rocprofv3 --pmc WRITE_SIZE -d $out/pmc_write -o bench -- python $GRAFT_REPO_ROOT/bench.py $args --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > /dev/null 2>> $out/rocprof_stderr.log
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $out "$args" > $out/summary.txt 2>&1
cat $out/summary.txt
