#!/bin/bash
# GPU box: rocprofv3 kernel trace + stats of a bench command (the PMC passes are tools/gpu_traffic.sh: never in one run with a trace).
# usage: tools/gpu_profile.sh <tag> [bench.py arguments, default: the driver's C2 line without the CPU legs]
tag=${1:-r03}; shift
args="$*"; [ -z "$args" ] && args="--steps 5 --warmup 2"
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
cd /tmp
rocprofv3 --kernel-trace --stats -d $out/trace -o bench -- python $GRAFT_REPO_ROOT/bench.py $args --min-seconds 0 --no-cpu-baseline --no-e2e > $out/bench_under_rocprof.json 2> $out/rocprof_stderr.log
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $out "$args" > $out/summary.txt 2>&1
cat $out/summary.txt
find $out -name "*.db" -delete
