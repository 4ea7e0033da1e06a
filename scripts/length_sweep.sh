#!/bin/bash
# GPU box: kernel throughput by read length (full-word lengths and the BASELINE ones), Myers / BitPAl 2,-3,-5 / banded k=8.
# 3000 queries x 1M subjects (Myers, banded), 500 x 1M (BitPAl); output: one line per (algorithm, length).
out=${1:-gpurun_out/length_sweep.txt}; : > $out
line() { python3 -c "
import json,sys; r=json.loads(sys.stdin.read()); i=r['roofline']['issued']; print('$1', $2, r['config']['kernel'], r['roofline']['kernel_gcups'], 'GCUPS, issued', i['frac'] if i else None)"; }
for L in 32 64 96 128 150 160 192 256 320 384 512 640 768 800 1000 1024; do
  timeout -k 10 240 python3 bench.py --config 2 --length $L --nq 3000 --steps 2 --no-cpu-baseline --no-total --no-clock-probe 2>/dev/null | line myers $L >> $out; done
for L in 64 100 150 200 256 352 500; do
  timeout -k 10 240 python3 bench.py --config 4 --length $L --nq 500 --steps 2 --no-cpu-baseline --no-total --no-clock-probe 2>/dev/null | line bitpal $L >> $out; done
for L in 100 150 250 500; do
  timeout -k 10 240 python3 bench.py --config 3 --length $L --nq 3000 --steps 2 --banded-mix random --banded-variants '' --no-cpu-baseline --no-total --no-clock-probe 2>/dev/null | line banded_random $L >> $out
  timeout -k 10 240 python3 bench.py --config 3 --length $L --nq 1000 --steps 2 --banded-mix survivors --banded-variants '' --no-cpu-baseline --no-total --no-clock-probe 2>/dev/null | line banded_survivors $L >> $out; done
cat $out
