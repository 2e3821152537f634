#!/bin/bash
# round 6, run Y: the shape fuzzer end to end on the GPU - front end, host decoders, .tap - against the oracle's transitions and .tap
mkdir -p gpurun_out/r06y
for s in 40000 40100 100800; do
  timeout 1000 python tools/fuzz_shapes.py --gpu --e2e $s 100 > gpurun_out/r06y/fuzz_$s.log 2>&1; echo "fuzz e2e $s rc $? ok $(grep -c '^ok' gpurun_out/r06y/fuzz_$s.log) fail $(grep -c '^FAIL ' gpurun_out/r06y/fuzz_$s.log)"
  grep -A3 '^FAIL ' gpurun_out/r06y/fuzz_$s.log | head -12
done
