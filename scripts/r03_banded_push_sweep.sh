#!/bin/bash
# Regroup policy of the banded kernel (BGSA_BANDED_PUSH_ROW = rows after row k from which a test may hand few survivors
# to the dense pass, BGSA_BANDED_PUSH_MAX = how many lanes count as few): kernel ms per subject mix.  Any value gives the
# same scores; this picks the defaults.
out=gpurun_out/${1:-r03}; mkdir -p $out
ROWS=${ROWS:-24 32 40 48}; MAXES=${MAXES:-4 8 16}; MIXES=${MIXES:-random dense1pct}
for row in $ROWS; do for mx in $MAXES; do
  line="push_row=k+$row push_max=$mx"
  for mix in $MIXES; do
    ms=$(BGSA_BANDED_PUSH_ROW=$row BGSA_BANDED_PUSH_MAX=$mx timeout -k 10 200 python bench.py --config 3 --banded-mix $mix --banded-variants '' --steps 3 --warmup 1 --no-cpu-baseline --no-total --no-clock-probe 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['roofline']['kernel_ms'])" 2>/dev/null)
    line="$line  $mix=${ms:-fail}"
  done
  echo "$line" | tee -a $out/banded_push_sweep.txt
done; done
