#!/bin/bash
export RTFE_PEAK_PATH=1 TMPDIR=/tmp
out=$PWD/gpurun_out/prof_r3h
mkdir -p $out
cd /tmp
rocprofv3 --kernel-trace --stats -d $out/trace -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-e2e > $out/bench.json 2> $out/err.log
cd $GRAFT_REPO_ROOT
find $out -name "*kernel_stats*" | head; f=$(find $out -name "*kernel_stats.csv" | head -1); head -20 "$f" | cut -c1-200
python tools/gpu_case_stats.py gcr 2>&1 | tail -4
