#!/bin/bash
# round 6, run U: the last check of the tree as it is committed - the whole -m gpu suite, smoke(), the driver's line
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/r06u_pytest.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 1200 python bench.py > gpurun_out/r06u_bench.json 2> gpurun_out/r06u_bench.err; echo "bench rc $?"; python -c "
import json; j=json.loads(open('gpurun_out/r06u_bench.json').read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['cpu_baseline']['value'], j['e2e']['value']); print({k: (v.get('ms_per_step'), v.get('error')) for k, v in j['other_configs'].items()})"
