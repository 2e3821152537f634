#!/bin/bash
# GPU box, one call: the new tests first, the default bench line (with e2e), then C3 / C4, every step under its own timeout and
# with its log written as it goes.
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ingest.py -x -q > gpurun_out/ingest_tests.log 2>&1; echo "ingest rc $?"; tail -5 gpurun_out/ingest_tests.log
timeout 600 python bench.py --steps 5 --warmup 2 > gpurun_out/bench_c2.json 2> gpurun_out/bench_c2.err; echo "C2 rc $?"; tail -c 3000 gpurun_out/bench_c2.json; tail -3 gpurun_out/bench_c2.err
timeout 900 python bench.py --config C3 --steps 3 --warmup 1 > gpurun_out/bench_c3.json 2> gpurun_out/bench_c3.err; echo "C3 rc $?"; tail -c 3000 gpurun_out/bench_c3.json; tail -3 gpurun_out/bench_c3.err
timeout 900 python bench.py --config C4 --steps 3 --warmup 1 > gpurun_out/bench_c4.json 2> gpurun_out/bench_c4.err; echo "C4 rc $?"; tail -c 3000 gpurun_out/bench_c4.json; tail -3 gpurun_out/bench_c4.err
