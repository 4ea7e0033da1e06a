#!/bin/bash
# GPU box, round 5: the block size K of the chains-in-turns rows (30 / 32 words), finer than r05_split_ab.sh: K = 6 .. 10 on the final
# bodies (schedule_ilp 2,24), static grids for all (the counter instantiation only fits 256 VGPRs up to K = 8), config 5.
set -e
cd "$(dirname "$0")/.."
run() { python3 bench.py --config 5 --steps 2 --warmup 1 --no-cpu-baseline --no-total "$@" 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.readline()); print('  ', r['config']['kernel'], '|', r['roofline']['kernel_ms'], 'ms |', r['value'], 'GCUPS | MHz', (r.get('clock') or {}).get('sustained_mhz'), '| checksum', r['checksum'])"; }
lib() { echo $PWD/bgsa_amd/_ab/libbgsa_hip_$1.so; }
for i in 1 2; do
echo "== round $i (static grids)"
echo "K=8 (default library)"; BGSA_DYNAMIC_TASKS=0 run
for k in 6 7 9 10; do echo "K=$k"; BGSA_HIP_LIB=$(lib k$k) BGSA_DYNAMIC_TASKS=0 run; done
done
echo "== counter grids where they fit"
echo "K=8"; run
for k in 6 7; do echo "K=$k"; BGSA_HIP_LIB=$(lib k$k) run; done
