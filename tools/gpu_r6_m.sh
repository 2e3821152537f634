#!/bin/bash
# round 6, run M: k_prep's work list in chunks; where k_gain's tails spend their time on the noisy tape (RTFE_DEBUG=4)
mkdir -p gpurun_out/r06m
one() { local label=$1; shift
   env "$@" timeout 900 python bench.py --no-cpu-baseline --no-e2e --no-other-configs $EXTRA > gpurun_out/r06m/$label.json 2> gpurun_out/r06m/$label.err
   python - <<PY
import json
try:
    j = json.loads(open("gpurun_out/r06m/$label.json").read().strip().splitlines()[-1])
    print("$label value", j["value"], "ms", j["ms_per_step"], "serial", j["ms_per_step_serial"], "frac", j["roofline"]["frac"], "flagged", j["config"]["flagged_bursts"], "events", j["config"]["events_total"], "seq", j["config"]["last_scan_stats"]["sequential"], {k: v for k, v in j["kernel_ms"].items() if v > 0.02}, j["config"]["last_scan_stats"]["phase_cycles"])
except Exception as e:
    print("$label FAILED", e); print(open("gpurun_out/r06m/$label.err").read()[-600:])
PY
}
EXTRA="--steps 20 --warmup 5" one c2 A=1
EXTRA="--config N1 --steps 5 --warmup 2" one n1 A=1
EXTRA="--config N1 --steps 3 --warmup 1" one n1_dbg4 RTFE_DEBUG=4
EXTRA="--config N1f --steps 5 --warmup 2" one n1f A=1
EXTRA="--config M8 --steps 5 --warmup 2" one m8 A=1
for s in 5000 5100; do
  timeout 900 python tools/fuzz_shapes.py --gpu $s 100 > gpurun_out/r06m/fuzz_$s.log 2>&1; echo "fuzz $s rc $? ok $(grep -c '^ok' gpurun_out/r06m/fuzz_$s.log) fail $(grep -c '^FAIL ' gpurun_out/r06m/fuzz_$s.log)"
  grep -A3 '^FAIL ' gpurun_out/r06m/fuzz_$s.log | head -12
done
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r06m/prof_n1 -- python $GRAFT_REPO_ROOT/bench.py --config N1 --steps 5 --warmup 2 --no-overlap --no-cpu-baseline --no-e2e --no-other-configs > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
python - <<'PY'
import glob, csv
for f in glob.glob("gpurun_out/r06m/prof_n1/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:14]: print(r["Name"][:40], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"])
PY
