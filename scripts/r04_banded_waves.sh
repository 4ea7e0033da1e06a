#!/bin/bash
# Round 4: the banded kernels against the waves per SIMD of their persistent grid (BGSA_PERSISTENT_PER_CU = workgroups of four waves
# per CU; the default is what the occupancy query allows: 6 for k <= 12, 7 for the funnel-shift forms).  10k x 1M x 150 bp, every pair
# surviving, kernel ms.  The pair row ran SLOWER with eight waves than with six in the microbenchmark (profiles/r04_ubench_banded_pair.txt).
out=${1:-gpurun_out/r04/banded_waves.txt}; mkdir -p $(dirname $out); : > $out
one() { local k=$1 w=$2
  r=$(BGSA_PERSISTENT_PER_CU=$w timeout -k 10 200 python bench.py --config 3 --k $k --banded-mix survivors --banded-variants '' --steps 2 --warmup 1 --no-cpu-baseline --no-total --no-other-configs 2>/dev/null |
      python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['roofline']['kernel_ms'], 'ms', (r.get('clock') or {}).get('sustained_mhz'), 'MHz', r['config']['kernel'])" 2>/dev/null)
  echo "k=$k waves/SIMD<=$w: $r" | tee -a $out; }
for k in 8 13 31; do for w in 3 4 5 6 7 8; do one $k $w; done; done
