#!/bin/bash
# Round 4, BitPAl row body A/B on one MI355X (config 4, 10k x 1M x 150 bp; kernel ms from bench.py's HIP events, checksums equal):
# the library before the change, each of the two rewrites alone (built with BGSA_GEN_BITPAL_INLINE_LE=0 / BGSA_GEN_BITPAL_ONE_CHAIN=0),
# both, and both with the persistent grid at eight workgroups per CU as before (BGSA_PERSISTENT_PER_CU=8).
out=${1:-gpurun_out/r04d/bitpal_ab.txt}
P=$PWD/bgsa_amd/_prev
one() { local label=$1; shift
  r=$(env "$@" timeout -k 10 400 python bench.py --config 4 --steps 3 --warmup 1 --no-cpu-baseline --no-total 2>/dev/null |
      python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['roofline']['kernel_ms'], r['value'], r['checksum'], (r.get('clock') or {}).get('sustained_mhz'), r['roofline']['issued']['generator_count']['valu_per_row'])" 2>/dev/null)
  echo "$label: $r" | tee -a $out; }
for rep in 1 2; do
  one "previous (69 VALU, 13 chains)      " BGSA_HIP_LIB=$P/libbgsa_hip_prev.so
  one "u<=D plane inlined only (68, 13)   " BGSA_HIP_LIB=$P/libbgsa_hip_le_only.so
  one "one chain per class only (69, 9)   " BGSA_HIP_LIB=$P/libbgsa_hip_chain_only.so
  one "both (68, 9)                       " BGSA_X=1
  one "both, 8 workgroups per CU          " BGSA_PERSISTENT_PER_CU=8
done
