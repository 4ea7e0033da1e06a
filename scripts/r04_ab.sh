#!/bin/bash
# Round 4 A/B on one MI355X, same box, same process order: the library before a change (bgsa_amd/_prev/libbgsa_hip_prev.so,
# built from the commit before it) against the tree's library, kernel ms from bench.py's HIP events, checksums compared.
#   bash scripts/r04_ab.sh <out file> <config> [more bench.py arguments]
out=$1; shift
cfg=$1; shift
PREV=$PWD/bgsa_amd/_prev/libbgsa_hip_prev.so
one() { # one <label> <env...>
  local label=$1; shift
  local r
  r=$(env "$@" timeout -k 10 400 python bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-total --no-other-configs --banded-variants '' $EXTRA 2>/dev/null |
      python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['roofline']['kernel_ms'], r['value'], r['config']['kernel'], r['checksum'], (r.get('clock') or {}).get('sustained_mhz'))" 2>/dev/null)
  echo "cfg$cfg $EXTRA $label: $r" | tee -a $out
}
EXTRA="$*"
one "previous" BGSA_HIP_LIB=$PREV
one "now     " BGSA_X=1
one "previous" BGSA_HIP_LIB=$PREV
one "now     " BGSA_X=1
