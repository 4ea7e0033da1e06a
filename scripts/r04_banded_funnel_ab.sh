#!/bin/bash
# Round 4: thresholds 13 .. 31 — round 2's one-group funnel-shift loops (banded_asm_kernel, BGSA_BANDED_IMPL=a) against the
# two-group loop of the one-word-window kernel around the same rows (banded_cut_kernel<2, ., funnel32 / funnel64>: woven dispatch,
# solid-survivor pushes, task counter, first words re-read per query).  10k x 1M x 150 bp, same box, kernel ms, checksums compared.
out=${1:-gpurun_out/r04/banded_funnel_ab.txt}
one() { local label=$1 k=$2 mix=$3; shift 3
  r=$(env "$@" timeout -k 10 300 python bench.py --config 3 --k $k --banded-mix $mix --banded-variants '' --steps 3 --warmup 1 --no-cpu-baseline --no-total --no-clock-probe 2>/dev/null |
      python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['roofline']['kernel_ms'], r['config']['kernel'], r['checksum'])" 2>/dev/null)
  echo "k=$k $mix $label: $r" | tee -a $out; }
for k in 13 15 16 24 31; do
  for mix in survivors random; do
    one "one group per wave (round 2's loop)" $k $mix BGSA_BANDED_IMPL=a
    one "two groups per wave (now)          " $k $mix BGSA_X=1
  done
done
