#!/bin/bash
mkdir -p gpurun_out/r06i
for i in 1 2 3 4 5; do timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --min-seconds 0.2 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=j['e2e']; print('e2e', e.get('value'), e.get('seconds'), 'windows', e.get('windows'), 'replay ev/s/thread', e.get('host_replay_events_per_s_per_thread'), 'identical', e.get('tap_identical_to_cpu_port'), e.get('error'))"; done
timeout 900 python -m pytest tests/test_gpu_ingest.py -x -q 2>&1 | tail -2
