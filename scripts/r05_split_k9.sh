#!/bin/bash
# GPU box, round 5: K = 8 against K = 9 on the counter grids (config 5 and 930 bp); libraries: scripts/build_variant.sh k9 BGSA_GEN_MYERS_SPLIT=9 etc.
set -e
cd "$(dirname "$0")/.."
run() { python3 bench.py --config 5 --steps 2 --warmup 1 --no-cpu-baseline --no-total "$@" 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.readline()); print('  ', r['config']['kernel'], '|', r['roofline']['kernel_ms'], 'ms |', r['value'], 'GCUPS | MHz', (r.get('clock') or {}).get('sustained_mhz'), '| checksum', r['checksum'])"; }
for i in 1 2 3; do
echo "K=8 counter"; run
echo "K=9 counter"; BGSA_HIP_LIB=$PWD/bgsa_amd/_ab/libbgsa_hip_k9.so run
done
echo "930 bp K=8"; run --nq 1000 --ns 524288 --length 930
echo "930 bp K=9"; BGSA_HIP_LIB=$PWD/bgsa_amd/_ab/libbgsa_hip_k9.so run --nq 1000 --ns 524288 --length 930
echo "930 bp K=10"; BGSA_HIP_LIB=$PWD/bgsa_amd/_ab/libbgsa_hip_k10.so run --nq 1000 --ns 524288 --length 930
