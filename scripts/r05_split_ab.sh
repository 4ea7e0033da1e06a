#!/bin/bash
# GPU box, round 5: Myers at 30 / 32 words (897..1024 bp, BASELINE config 5) — the code planes (nine instructions per word, the
# default until round 5) against resident Peq planes with the two carry chains in turns over blocks of K words
# (rows_ir.myers_body(split=K): eight instructions per word).  Same box, interleaved.  Parity first.
#     scripts/r05_split_ab.sh > gpurun_out/r05_split_ab.txt
set -e
cd "$(dirname "$0")/.."
parity() {
python3 - <<'P'
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, bgsa_amd as B, oracle as O
bad = 0
for qlen, slen in [(997, 1000), (1021, 1024), (120, 1000), (300, 961), (950, 930), (40, 897), (1000, 1000)]:
    q = O.gen_reads(5000 + qlen, 11, qlen); s = O.gen_reads(6000 + slen, 200, slen)
    m = min(qlen, slen)
    s[:20, :m] = O.mutate(q[np.arange(20) % 11][:, :m], np.arange(20) % 7, slen)
    got = B.align_all_pairs(q, s, algo=B.ALGO_MYERS)
    name = B.lib().bgsa_hip_kernel_name(B.ALGO_MYERS, (slen + 31) // 32).decode()
    ok = np.array_equal(got, O.myers64(q, s))
    bad += not ok
    print(f"  parity {qlen}x{slen} {name}: {'ok' if ok else 'MISMATCH'}")
# long carries: homopolymers and shifted repeats
for length in (1000, 1024, 930):
    a = np.frombuffer(b"A" * length, dtype=np.uint8); ac = np.frombuffer((b"AC" * length)[:length], dtype=np.uint8)
    ca = np.frombuffer((b"CA" * length)[:length], dtype=np.uint8); n = np.frombuffer(b"N" * length, dtype=np.uint8)
    q = np.stack([a, ac, ca, n]); s = np.concatenate([q] * 16)
    ok = np.array_equal(B.align_all_pairs(q, s, algo=B.ALGO_MYERS), O.myers64(q, s)); bad += not ok
    print(f"  carries {length}: {'ok' if ok else 'MISMATCH'}")
sys.exit(1 if bad or B.lib().bgsa_hip_stream_faults(1) else 0)
P
}
run() { python3 bench.py --config 5 --steps 2 --warmup 1 --no-cpu-baseline --no-total "$@" 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.readline()); print('  ', r['config']['kernel'], '|', r['roofline']['kernel_ms'], 'ms |', r['value'], 'GCUPS | tile', (r['roofline']['traffic_model'] or {}).get('query_tile'), '| MHz', (r.get('clock') or {}).get('sustained_mhz'), '| checksum', r['checksum'])"; }
echo "== parity, resident Peq planes at 30/32 words (BGSA_MYERS_PEQ_MAX_WORDS=32), counter and static grids"
BGSA_MYERS_PEQ_MAX_WORDS=32 parity
BGSA_MYERS_PEQ_MAX_WORDS=32 BGSA_DYNAMIC_TASKS=0 parity
for lib in split4 split12; do echo "== parity $lib"; BGSA_HIP_LIB=$PWD/bgsa_amd/_ab/libbgsa_hip_$lib.so BGSA_MYERS_PEQ_MAX_WORDS=32 BGSA_DYNAMIC_TASKS=0 parity; done
for i in 1 2; do
echo "== round $i: config 5 (1k x 1M x 1000 bp)"
echo "code planes (default)";            run
echo "Peq resident, K=8, counter";        BGSA_MYERS_PEQ_MAX_WORDS=32 run
echo "Peq resident, K=8, static";         BGSA_MYERS_PEQ_MAX_WORDS=32 BGSA_DYNAMIC_TASKS=0 run
echo "Peq resident, K=4, counter";        BGSA_HIP_LIB=$PWD/bgsa_amd/_ab/libbgsa_hip_split4.so BGSA_MYERS_PEQ_MAX_WORDS=32 run
echo "Peq resident, K=4, static";         BGSA_HIP_LIB=$PWD/bgsa_amd/_ab/libbgsa_hip_split4.so BGSA_MYERS_PEQ_MAX_WORDS=32 BGSA_DYNAMIC_TASKS=0 run
echo "Peq resident, K=12, static";        BGSA_HIP_LIB=$PWD/bgsa_amd/_ab/libbgsa_hip_split12.so BGSA_MYERS_PEQ_MAX_WORDS=32 BGSA_DYNAMIC_TASKS=0 run
echo "code planes, static";               BGSA_DYNAMIC_TASKS=0 run
done
echo "== 930 bp (30 words), 1k x 512k"
echo "code planes";                 run --nq 1000 --ns 524288 --length 930
echo "Peq resident K=8";            BGSA_MYERS_PEQ_MAX_WORDS=32 run --nq 1000 --ns 524288 --length 930
