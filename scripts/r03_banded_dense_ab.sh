#!/bin/bash
# The funnel-shift banded kernels (k >= 13, and BGSA_BANDED_IMPL=a at k = 8) before and after their dense pass went from
# sixteen character registers to two: previous library against this one, same box.
out=gpurun_out/${1:-r03}; mkdir -p $out
one() { local label=$1 e=$2; shift 2
  r=$(env $e timeout -k 10 300 python bench.py "$@" --no-cpu-baseline --no-total --no-clock-probe --banded-variants '' 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['roofline']['kernel_ms'], r['value'], r['config']['kernel'], r['checksum'])" 2>/dev/null)
  echo "$label: ${r:-fail}" | tee -a $out/banded_dense_ab.txt; }
PREV=BGSA_HIP_LIB=$PWD/bgsa_amd/_prev/libbgsa_hip_prev.so
for rep in 1 2; do
for k in 13 16 31; do for mix in random survivors dense1pct; do
  one "k=$k $mix previous" "$PREV" --config 3 --k $k --steps 3 --banded-mix $mix
  one "k=$k $mix now     " "X=1" --config 3 --k $k --steps 3 --banded-mix $mix
done; done
for mix in random survivors dense1pct; do
  one "k=8 funnel loop $mix previous" "$PREV BGSA_BANDED_IMPL=a" --config 3 --steps 3 --banded-mix $mix
  one "k=8 funnel loop $mix now     " "BGSA_BANDED_IMPL=a" --config 3 --steps 3 --banded-mix $mix
done; done
