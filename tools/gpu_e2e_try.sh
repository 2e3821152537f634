#!/bin/bash
# GPU box: the e2e object of bench.py under environment overrides.  usage: tools/gpu_e2e_try.sh "ENV=.. ENV=.."
env $1 timeout 900 python bench.py --no-cpu-baseline --steps 3 --warmup 2 > /tmp/e.json 2> /tmp/e.err || tail -3 /tmp/e.err
python - <<'PY'
import json
j = json.loads(open("/tmp/e.json").read().strip().splitlines()[-1])["e2e"]
print({k: j.get(k) for k in ("value", "seconds", "windows", "host_replay_seconds_summed", "host_replay_events_per_s_per_thread", "host_replay_threads", "host_read_threads", "file_read_seconds_overlapped", "scan_wait_seconds", "tap_identical_to_cpu_port", "error")})
PY
