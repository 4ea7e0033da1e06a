#!/bin/bash
# Round 4: the funnel-shift rows with the error count left to the events (rows_ir.banded_body_coll / banded_body64_coll: 10 / 19
# VALU per row instead of 12 / 22; D0 in a fixed register pair, v_lshrrev_b64 / v_lshl_add_u64) against the library of the commit
# before (bgsa_amd/_prev/libbgsa_hip_prev.so).  10k x 1M x 150 bp, same box, kernel ms, checksums compared.
out=${1:-gpurun_out/r04/banded_collector_ab.txt}
PREV=$PWD/bgsa_amd/_prev/libbgsa_hip_prev.so
one() { local label=$1 k=$2 mix=$3; shift 3
  r=$(env "$@" timeout -k 10 300 python bench.py --config 3 --k $k --banded-mix $mix --banded-variants '' --steps 3 --warmup 1 --no-cpu-baseline --no-total --no-clock-probe --no-other-configs 2>/dev/null |
      python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['roofline']['kernel_ms'], r['config']['kernel'], r['checksum'])" 2>/dev/null)
  echo "k=$k $mix $label: $r" | tee -a $out; }
for k in ${KS:-13 15 16 24 31}; do
  for mix in ${MIXES:-survivors random}; do
    one "12 / 22 per row (before)" $k $mix BGSA_HIP_LIB=$PREV
    one "10 / 19 per row (now)   " $k $mix BGSA_X=1
  done
done
