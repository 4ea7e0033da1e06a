#!/bin/bash
# GPU box, round 5: config 5 with the persistent workgroups of the counter kernel started with a per-workgroup delay (bgsa_common.h:
# dephase_persistent_workgroup; library: scripts/build_variant.sh dephase EXTRA=-DBGSA_MYERS_DEPHASE=1 on a tree with that three-line #ifdef in myers_global_asm_kernel; not kept) against the default.
set -e
cd "$(dirname "$0")/.."
run() { python3 bench.py --config 5 --steps 2 --warmup 1 --no-cpu-baseline --no-total "$@" 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.readline()); print('  ', r['config']['kernel'], '|', r['roofline']['kernel_ms'], 'ms |', r['value'], 'GCUPS | checksum', r['checksum'])"; }
for i in 1 2 3; do
echo "default"; run
echo "dephased start"; BGSA_HIP_LIB=$PWD/bgsa_amd/_ab/libbgsa_hip_dephase.so run
done
