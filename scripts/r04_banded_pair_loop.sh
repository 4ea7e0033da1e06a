#!/bin/bash
# Round 4: the 64-bit pair rows (k = 16 .. 31) in the new loop with ONE group per wave (woven dispatch, task counter, first words
# re-read per query) against round 2's loop and against two groups.  10k x 1M x 150 bp, every pair surviving, same box.
out=${1:-gpurun_out/r04/banded_pair_loop.txt}
one() { local label=$1 k=$2; shift 2
  r=$(env "$@" timeout -k 10 300 python bench.py --config 3 --k $k --banded-mix survivors --banded-variants '' --steps 3 --warmup 1 --no-cpu-baseline --no-total --no-clock-probe 2>/dev/null |
      python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['roofline']['kernel_ms'], r['checksum'])" 2>/dev/null)
  echo "k=$k $label: $r" | tee -a $out; }
for k in 16 31; do
  one "round 2's loop (default)            " $k BGSA_X=1
  one "new loop, one group (PAIR_LOOP=1)   " $k BGSA_BANDED_PAIR_LOOP=1
  one "new loop, one group, static grid    " $k BGSA_BANDED_PAIR_LOOP=1 BGSA_BANDED_DYNAMIC=0
  one "new loop, two groups (PAIR_LOOP=2)  " $k BGSA_BANDED_PAIR_LOOP=2
done
