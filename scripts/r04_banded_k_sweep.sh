#!/bin/bash
# Round 4: the banded kernels over the whole threshold range, 10k x 1M x 150 bp, kernel ms per pass — every pair surviving (the full
# band of every row is computed: the row body's rate) and uniform random pairs (how soon the waves stop).
out=${1:-gpurun_out/r04/banded_k_sweep.txt}; mkdir -p $(dirname $out); : > $out
one() { local k=$1 mix=$2
  r=$(timeout -k 10 200 python bench.py --config 3 --k $k --banded-mix $mix --banded-variants '' --steps 2 --warmup 1 --no-cpu-baseline --no-total --no-clock-probe --no-other-configs 2>/dev/null |
      python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['roofline']['kernel_ms'], 'ms', r['config']['kernel'], 'checksum', r['checksum'])" 2>/dev/null)
  echo "k=$k $mix: $r" | tee -a $out; }
for k in 1 2 4 6 8 9 12 13 14 15 16 18 20 24 28 31; do
  one $k survivors
  one $k random
done
