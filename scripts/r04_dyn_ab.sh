#!/bin/bash
# Round 4: the task counter (persistent grid, dynamic handout) for kernels it used to cost a wave per SIMD — after the
# loop-invariant values of the task loop were laundered out of the VGPRs (BitPAl 5 words 103 -> 95, banded cut 89 -> 75).
# Same box, kernel ms from bench.py's HIP events, checksums compared.
out=${1:-gpurun_out/r04/dyn_ab.txt}
one() { local label=$1 cfg=$2; shift 2
  r=$(env "$@" timeout -k 10 400 python bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-total $EXTRA 2>/dev/null |
      python -c "
import sys,json
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
v=r.get('banded_variants') or {}
print(r['roofline']['kernel_ms'], r['config']['kernel'], r['checksum'], (r.get('clock') or {}).get('sustained_mhz'), ' '.join(f'{k}={x[\"kernel_ms\"]}' for k,x in v.items()))" 2>/dev/null)
  echo "cfg$cfg $label: $r" | tee -a $out; }
for rep in 1 2; do
  EXTRA="" one "BitPAl static grid (BGSA_DYNAMIC_TASKS=0)" 4 BGSA_DYNAMIC_TASKS=0
  EXTRA="" one "BitPAl task counter (default)           " 4 BGSA_X=1
  EXTRA="" one "banded static grid (default so far)     " 3 BGSA_X=1
  EXTRA="" one "banded task counter (BGSA_BANDED_DYNAMIC=1)" 3 BGSA_BANDED_DYNAMIC=1
done
