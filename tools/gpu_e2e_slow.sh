#!/bin/bash
# GPU box: repeats the traced streaming decode and prints the trace of the slowest run (where does a stall sit?)
for i in 1 2 3 4 5; do timeout 600 python tools/gpu_e2e_trace.py 16 23 24 96 12 > /tmp/tr$i.txt 2>&1; grep "^{" /tmp/tr$i.txt | python -c "import json,sys; j=json.loads(sys.stdin.read()); print($i, j['seconds'])"; done > /tmp/secs.txt
cat /tmp/secs.txt
w=$(sort -k2 -n /tmp/secs.txt | tail -1 | cut -d' ' -f1)
grep -v "^f_\|^l_\|wait_ctx\|gpu_scan" /tmp/tr$w.txt | tail -72
