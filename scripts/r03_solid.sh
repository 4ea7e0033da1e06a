#!/bin/bash
# Banded regroup policy, second pass: from which row (BGSA_BANDED_PUSH_SOLID, rows after k) a wave holding a solid survivor
# (at most limit - BGSA_BANDED_SOLID_MARGIN errors) may hand its few alive lanes to the dense pass.  Kernel ms per mix.
out=gpurun_out/${1:-r03}; mkdir -p $out
b3() { timeout -k 10 250 python bench.py --config 3 --banded-mix $1 --banded-variants '' --steps 3 --warmup 1 --no-cpu-baseline --no-total --no-clock-probe 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['roofline']['kernel_ms'])" 2>/dev/null; }
for cfg in "40 2" "32 4" "32 5" "24 5" "32 6" "24 6" "16 6"; do
  set -- $cfg
  line="push_solid=k+$1 margin=$2:"
  for mix in random dense1pct planted; do line="$line $mix=$(BGSA_BANDED_PUSH_SOLID=$1 BGSA_BANDED_SOLID_MARGIN=$2 b3 $mix)"; done
  echo "$line" | tee -a $out/solid.txt
done
