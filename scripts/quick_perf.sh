#!/bin/bash
# GPU box: short A/B of the kernels that have generated row loops (not the BASELINE-size record).
set -e
run() { python3 bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.readline()); print(r['config']['kernel'], '|', r['config']['queries'],'x',r['config']['subjects_per_gpu'],'x',r['config']['length_bp'], '|', r['value'], 'GCUPS', '|', (r['roofline']['issued'] or {}).get('frac'))"; }
run --config 2 --nq 2000
run --config 4 --nq 1000
run --config 5
run --config 2 --nq 500 --ns 128000 --length 300
run --config 2 --nq 200 --ns 20000 --length 4000
run --config 4 --nq 200 --ns 64000 --length 1000
run --config 4 --nq 1000 --ns 256000 --length 250
