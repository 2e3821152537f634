#!/bin/bash
# GPU box: HBM traffic per kernel of a bench.py configuration from the PMC counters - separate --pmc passes (never with trace domains),
# calibrated in the same session on kernels of known traffic.  Writes gpurun_out/traffic_<tag>/pmc_<config>.json (copy to profiles/).
# usage: tools/gpu_traffic.sh <round tag> <config, default C2> [more bench.py arguments]
tag=${1:-r03}; config=${2:-C2}; shift; shift
export TMPDIR=/tmp
out=$PWD/gpurun_out/traffic_${tag}_$config
rm -rf $out; mkdir -p $out
cd /tmp
rocprofv3 --pmc FETCH_SIZE -d $out/calib_fetch -o c -- python $GRAFT_REPO_ROOT/tools/pmc_calib.py > $out/calib.log 2> $out/err.log
rocprofv3 --pmc WRITE_SIZE -d $out/calib_write -o c -- python $GRAFT_REPO_ROOT/tools/pmc_calib.py >> $out/calib.log 2>> $out/err.log
rocprofv3 --pmc FETCH_SIZE -d $out/pmc_fetch -o b -- python $GRAFT_REPO_ROOT/bench.py --config $config --steps 2 --warmup 1 --min-seconds 0 --no-cpu-baseline --no-e2e "$@" > $out/bench_fetch.json 2>> $out/err.log
rocprofv3 --pmc WRITE_SIZE -d $out/pmc_write -o b -- python $GRAFT_REPO_ROOT/bench.py --config $config --steps 2 --warmup 1 --min-seconds 0 --no-cpu-baseline --no-e2e "$@" > $out/bench_write.json 2>> $out/err.log
cd $GRAFT_REPO_ROOT
rows=$(python -c "import json,sys; print(json.loads(open('$out/bench_fetch.json').read().strip().splitlines()[-1])['config']['rows_per_gpu'])")
python tools/pmc_json.py $out $config $rows $tag > $out/pmc_$config.json
cat $out/pmc_$config.json | head -80
find $out -name "*.db" -delete
