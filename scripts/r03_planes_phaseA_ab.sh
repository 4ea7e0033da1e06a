#!/bin/bash
# the long-subject row body (code planes, two waves per SIMD) with phase A one word behind itself: previous library against this one
out=gpurun_out/${1:-r03}; mkdir -p $out
one() { local label=$1 e=$2; shift 2
  r=$(env $e timeout -k 10 300 python bench.py "$@" --no-cpu-baseline --no-total --no-clock-probe 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['roofline']['kernel_ms'], r['value'], r['config']['kernel'], r['checksum'])" 2>/dev/null)
  echo "$label: ${r:-fail}" | tee -a $out/planes_phaseA_ab.txt; }
PREV=BGSA_HIP_LIB=$PWD/bgsa_amd/_prev/libbgsa_hip_prev.so
for rep in 1 2; do
  one "1000 bp previous" "$PREV" --config 5 --steps 2
  one "1000 bp now     " "X=1" --config 5 --steps 2
  one " 900 bp previous" "$PREV" --config 5 --length 900 --steps 2
  one " 900 bp now     " "X=1" --config 5 --length 900 --steps 2
done
