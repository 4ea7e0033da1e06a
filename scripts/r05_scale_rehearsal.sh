#!/bin/bash
# GPU box, round 5: the line `bench.py --gpus 4` prints, rehearsed on the ONE card there is — four ranks under torch.distributed.run
# exactly as the driver starts them, all on device 0 (BGSA_BENCH_SAME_GPU=1), collectives through gloo (RCCL refuses two ranks on one
# device; tiles then travel through host memory).  A rehearsal of the code path and of the line's shape at the BASELINE sizes — one
# 10k x 1M bucket cut by plan_shards, the streamed gather inside the timed region — NOT a measurement: four ranks share one GPU and
# the transport is not xGMI.
#     scripts/r05_scale_rehearsal.sh [ranks] > gpurun_out/r05_scale_rehearsal.json
cd "$(dirname "$0")/.."
ranks=${1:-4}
export BGSA_BENCH_SAME_GPU=1 BGSA_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 BGSA_BENCH_TIMEOUT=1100 BGSA_BENCH_GATHER_TIMEOUT=600 \
       BGSA_BENCH_STRONG_TIMEOUT=400 BGSA_BENCH_WEAK_TIMEOUT=300
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $ranks --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus $ranks \
    --steps 2 --warmup 1 2> gpurun_out/r05_scale_rehearsal.err | grep '^{'
