#!/bin/bash
# GPU box, round 5: the query tile of the counter launches (queries per Peq load) — 32 / 16 / 8 until round 5, up to 128 (32 for the 30-
# and 32-word Myers kernels) since: time and checksum of configs 2, 4, 5 with the old and the new tiles.
#     scripts/r05_tile_ab.sh > gpurun_out/r05_tile_ab.txt
set -e
cd "$(dirname "$0")/.."
run() { python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-total --no-other-configs "$@" 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.readline()); print('  ', r['config']['kernel'], '|', r['roofline']['kernel_ms'], 'ms |', r['value'], 'GCUPS | tile', (r['roofline']['traffic_model'] or {}).get('query_tile'), '| model bytes / algorithmic', (r['roofline']['traffic_model'] or {}).get('ratio_to_algorithmic'), '| checksum', r['checksum'])"; }
for i in 1 2; do
echo "== round $i"
echo "config 2, tile <= 32";   BGSA_QUERY_TILE_MAX=32 run --config 2
echo "config 2, tile <= 128";  run --config 2
echo "config 5, tile <= 8";    BGSA_MYERS_LONG_TILE=8 run --config 5
echo "config 5, tile <= 32";   run --config 5
done
echo "config 4, tile <= 16";   BGSA_QUERY_TILE_MAX=16 run --config 4 --nq 4000
echo "config 4, tile <= 128";  run --config 4 --nq 4000
