#!/bin/bash
# GPU box: the N > 1 line for configs 3 and 4 (two ranks on one card over gloo, reduced sizes): shape and content check only.
cd "$(dirname "$0")/.."
export BGSA_BENCH_SAME_GPU=1 BGSA_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
for c in 3 4; do
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 2958$c bench.py --gpus 2 --config $c --steps 2 --warmup 1 --nq 300 --ns 64000 2>/dev/null | grep '^{' | python3 -c "
import json,sys
r=json.loads(sys.stdin.readline()); print('config', $c, r['scaling'], r['value'], 'kernel_only', r['kernel_only']['gcups'], 'gather_ok', r['gather_ok'], r['gather']['content_check']['segments_ok'], r['config']['kernel'], r['config']['subjects_total'])"
done
