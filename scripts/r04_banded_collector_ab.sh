#!/bin/bash
# Round 4: changes to the funnel-shift rows of thresholds 13 .. 31 against the library of the commit before
# (bgsa_amd/_prev/libbgsa_hip_prev.so: git archive <commit> bgsa_amd/csrc include | tar -x -C <dir>; make -C <dir>/bgsa_amd/csrc).
# Used for (i) commit 4151226: the error count left to the events (10 / 19 VALU per row: collector, v_lshrrev_b64, v_lshl_add_u64) —
# slower, removed; (ii) the tree: D0 >> 1 of the pair row as one v_lshrrev_b64 (21 VALU per row).  10k x 1M x 150 bp, same box,
# kernel ms, checksums compared.  profiles/r04_banded_collector_ab.txt.
out=${1:-gpurun_out/r04/banded_collector_ab.txt}; mkdir -p $(dirname $out)
PREV=$PWD/bgsa_amd/_prev/libbgsa_hip_prev.so
one() { local label=$1 k=$2 mix=$3; shift 3
  r=$(env "$@" timeout -k 10 300 python bench.py --config 3 --k $k --banded-mix $mix --banded-variants '' --steps 3 --warmup 1 --no-cpu-baseline --no-total --no-clock-probe --no-other-configs 2>/dev/null |
      python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['roofline']['kernel_ms'], r['config']['kernel'], r['checksum'])" 2>/dev/null)
  echo "k=$k $mix $label: $r" | tee -a $out; }
for k in ${KS:-13 15 16 24 31}; do
  for mix in ${MIXES:-survivors random}; do
    one "before" $k $mix BGSA_HIP_LIB=$PREV
    one "now   " $k $mix BGSA_X=1
  done
done
