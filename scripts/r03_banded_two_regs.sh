#!/bin/bash
# The one-word-window banded kernel with two registers per class and group (71 VGPRs, seven waves per SIMD): the four
# subject mixes of config 3, plus k = 12 (three cuts per advance) and the occupancy it is worth (LDS padding).
out=gpurun_out/${1:-r03}; mkdir -p $out
one() { # one <label> <env> <bench args...>
  local label=$1 e=$2; shift 2
  r=$(env $e timeout -k 10 300 python bench.py "$@" --no-cpu-baseline --no-total --no-clock-probe --banded-variants '' 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['roofline']['kernel_ms'], r['value'], r['config']['kernel'], r['checksum'])" 2>/dev/null)
  echo "$label: ${r:-fail}" | tee -a $out/banded_two_regs.txt
}
for rep in 1 2; do
for mix in planted random dense1pct survivors; do
  one "k=8 $mix" "X=1" --config 3 --steps 5 --banded-mix $mix
done; done
one "k=12 random" "X=1" --config 3 --k 12 --steps 5 --banded-mix random
one "k=12 survivors" "X=1" --config 3 --k 12 --steps 5 --banded-mix survivors
for pad in 20000 24000 30000; do   # 160 KB / pad: 7, 6, 5 workgroups per CU
  one "k=8 random, LDS pad $pad" "BGSA_BANDED_LDS_PAD=$pad" --config 3 --steps 5 --banded-mix random
  one "k=8 survivors, LDS pad $pad" "BGSA_BANDED_LDS_PAD=$pad" --config 3 --steps 5 --banded-mix survivors
done
