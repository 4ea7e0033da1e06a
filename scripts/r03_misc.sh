#!/bin/bash
# Round 3 one-off measurements on one MI355X (kernel ms from bench.py's HIP events):
#   banded: dispatch woven under the last row vs behind it (bgsa_amd/libbgsa_hip_noweave.so), the solid-survivor push rule;
#   BitPAl 10/-9/-15 (packed-carry column blocks) at 150 and 1,000 bp.
out=gpurun_out/${1:-r03}; mkdir -p $out
ms() { python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['roofline']['kernel_ms'], r['config']['kernel'], r['value'])" 2>/dev/null; }
b3() { timeout -k 10 250 python bench.py --config 3 --banded-mix $1 --banded-variants '' --steps 3 --warmup 1 --no-cpu-baseline --no-total --no-clock-probe 2>/dev/null | ms; }
for mix in random dense1pct survivors; do
  echo "banded $mix noweave: $(BGSA_HIP_LIB=$PWD/bgsa_amd/libbgsa_hip_noweave.so b3 $mix)" | tee -a $out/misc.txt
  echo "banded $mix weave  : $(b3 $mix)" | tee -a $out/misc.txt
done
for solid in 24 32 40 48; do
  line="push_solid=k+$solid (push_row k+48, max 8):"
  for mix in random dense1pct planted; do line="$line $mix=$(BGSA_BANDED_PUSH_SOLID=$solid b3 $mix | cut -d' ' -f1)"; done
  echo "$line" | tee -a $out/misc.txt
done
for mx in 4 16; do
  line="push_solid=k+32 max $mx:"
  for mix in random dense1pct; do line="$line $mix=$(BGSA_BANDED_PUSH_MAX=$mx b3 $mix | cut -d' ' -f1)"; done
  echo "$line" | tee -a $out/misc.txt
done
echo "bitpal 10,-9,-15 2k x 1M x 150: $(timeout -k 10 400 python bench.py --config 4 --scores=10,-9,-15 --nq 2000 --steps 1 --warmup 1 --no-cpu-baseline --no-total --no-clock-probe 2>$out/misc_bp150.err | ms)" | tee -a $out/misc.txt
echo "bitpal 10,-9,-15 200 x 64k x 1000: $(timeout -k 10 400 python bench.py --config 4 --scores=10,-9,-15 --nq 200 --ns 64000 --length 1000 --steps 1 --warmup 1 --no-cpu-baseline --no-total --no-clock-probe 2>$out/misc_bp1000.err | ms)" | tee -a $out/misc.txt
