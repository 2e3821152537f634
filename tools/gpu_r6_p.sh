#!/bin/bash
# round 6, run P: k_sift_hard's walk by the group (RTFE_DEBUG=9: fetch / tables / walk cycles of each workgroup's first wave)
mkdir -p gpurun_out/r06p
one() { local label=$1; shift
   env "$@" timeout 900 python bench.py --no-cpu-baseline --no-e2e --no-other-configs $EXTRA > gpurun_out/r06p/$label.json 2> gpurun_out/r06p/$label.err
   python - <<PY
import json
try:
    j = json.loads(open("gpurun_out/r06p/$label.json").read().strip().splitlines()[-1])
    print("$label value", j["value"], "ms", j["ms_per_step"], "serial", j["ms_per_step_serial"], "frac", j["roofline"]["frac"], "flagged", j["config"]["flagged_bursts"], "events", j["config"]["events_total"], "seq", j["config"]["last_scan_stats"]["sequential"], {k: v for k, v in j["kernel_ms"].items() if v > 0.02}, j["config"]["last_scan_stats"]["phase_cycles"][:6])
except Exception as e:
    print("$label FAILED", e); print(open("gpurun_out/r06p/$label.err").read()[-600:])
PY
}
EXTRA="--config N1 --steps 5 --warmup 2" one n1 A=1
EXTRA="--config N1 --steps 3 --warmup 1" one n1_dbg9 RTFE_DEBUG=9
EXTRA="--steps 20 --warmup 5" one c2 A=1
EXTRA="--config M8 --steps 5 --warmup 2" one m8 A=1
EXTRA="--config N1f --steps 5 --warmup 2" one n1f A=1
timeout 600 python tools/fuzz_shapes.py --gpu 9000 100 > gpurun_out/r06p/fuzz.log 2>&1; echo "fuzz rc $? ok $(grep -c '^ok' gpurun_out/r06p/fuzz.log) fail $(grep -c '^FAIL ' gpurun_out/r06p/fuzz.log)"
STRESS_SHAPES=1 timeout 900 bash tools/gpu_stress.sh 2900 1 60
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r06p/prof_n1 -- python $GRAFT_REPO_ROOT/bench.py --config N1 --steps 5 --warmup 2 --no-overlap --no-cpu-baseline --no-e2e --no-other-configs > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
find gpurun_out/r06p/prof_n1 -name "*kernel_stats.csv" | head -2
python - <<'PY'
import glob, csv
for f in glob.glob("gpurun_out/r06p/prof_n1/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:14]: print(r["Name"][:40], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"])
PY
