#!/bin/bash
# GPU box: the end-to-end line (file -> .tap) of the C2 sample under different host-side settings.
for s in "RT_E2E_READ_THREADS=4 RT_E2E_REPLAY_THREADS=32 RT_E2E_REPLAY_SPLIT=4" "RT_E2E_READ_THREADS=8 RT_E2E_REPLAY_THREADS=32 RT_E2E_REPLAY_SPLIT=4" "RT_E2E_READ_THREADS=8 RT_E2E_REPLAY_THREADS=64 RT_E2E_REPLAY_SPLIT=8" "RT_E2E_READ_THREADS=16 RT_E2E_REPLAY_THREADS=96 RT_E2E_REPLAY_SPLIT=8 RT_E2E_WINDOW_ROWS=4194304" "RT_E2E_READ_THREADS=16 RT_E2E_REPLAY_THREADS=128 RT_E2E_REPLAY_SPLIT=16" "RT_E2E_READ_THREADS=8 RT_E2E_REPLAY_THREADS=64 RT_E2E_REPLAY_SPLIT=8 RT_E2E_WINDOW_ROWS=16777216"; do
  env $s python bench.py --steps 2 --warmup 1 --min-seconds 0 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=j['e2e']; print('$s ->', e.get('value'), 'Msamples/s', e.get('seconds'), 'read', e.get('file_read_seconds_overlapped'), 'wait', e.get('scan_wait_seconds'), 'replay sum', e.get('host_replay_seconds_summed'), e.get('tap_identical_to_cpu_port'), e.get('error'))"
done
