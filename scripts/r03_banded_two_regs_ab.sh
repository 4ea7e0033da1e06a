#!/bin/bash
# previous build (four registers per class and group, 91 VGPRs) against this one, same box
out=gpurun_out/r03q; mkdir -p $out
one() { local label=$1 e=$2; shift 2
  r=$(env $e timeout -k 10 300 python bench.py "$@" --no-cpu-baseline --no-total --no-clock-probe --banded-variants '' 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['roofline']['kernel_ms'], r['value'], r['checksum'])" 2>/dev/null)
  echo "$label: ${r:-fail}" | tee -a $out/banded_two_regs_ab.txt; }
PREV=BGSA_HIP_LIB=$PWD/bgsa_amd/_prev/libbgsa_hip_prev.so
for rep in 1 2; do
for k in 8 12 4; do for mix in random survivors dense1pct; do
  one "k=$k $mix previous" "$PREV" --config 3 --k $k --steps 5 --banded-mix $mix
  one "k=$k $mix two regs" "X=1" --config 3 --k $k --steps 5 --banded-mix $mix
done; done; done
