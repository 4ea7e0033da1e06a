#!/bin/bash
# GPU box, round 5: the Myers row bodies as written (every instruction right behind the one it reads from) against the same bodies
# list-scheduled so that a true dependency keeps `gap` other instructions between producer and consumer (rows_ir.schedule_ilp;
# libraries built by scripts/build_variant.sh ilp1 BGSA_GEN_MYERS_ILP=1,12 / ilp2 BGSA_GEN_MYERS_ILP=2,24).  Same box, interleaved.
#     scripts/r05_ilp_ab.sh > gpurun_out/r05_ilp_ab.txt
set -e
cd "$(dirname "$0")/.."
parity() {
python3 - <<'P'
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, bgsa_amd as B, oracle as O
bad = 0
for qlen, slen in [(150, 150), (147, 150), (60, 33), (300, 512), (700, 768), (997, 1000), (1021, 1024), (300, 961), (950, 930), (200, 577)]:
    q = O.gen_reads(5000 + qlen, 11, qlen); s = O.gen_reads(6000 + slen, 200, slen)
    m = min(qlen, slen)
    s[:20, :m] = O.mutate(q[np.arange(20) % 11][:, :m], np.arange(20) % 7, slen)
    got = B.align_all_pairs(q, s, algo=B.ALGO_MYERS)
    name = B.lib().bgsa_hip_kernel_name(B.ALGO_MYERS, (slen + 31) // 32).decode()
    ok = np.array_equal(got, O.myers64(q, s))
    bad += not ok
    print(f"  parity {qlen}x{slen} {name}: {'ok' if ok else 'MISMATCH'}")
for length in (150, 1000, 1024, 930):
    a = np.frombuffer(b"A" * length, dtype=np.uint8); ac = np.frombuffer((b"AC" * length)[:length], dtype=np.uint8)
    ca = np.frombuffer((b"CA" * length)[:length], dtype=np.uint8); n = np.frombuffer(b"N" * length, dtype=np.uint8)
    q = np.stack([a, ac, ca, n]); s = np.concatenate([q] * 16)
    ok = np.array_equal(B.align_all_pairs(q, s, algo=B.ALGO_MYERS), O.myers64(q, s)); bad += not ok
    print(f"  carries {length}: {'ok' if ok else 'MISMATCH'}")
sys.exit(1 if bad or B.lib().bgsa_hip_stream_faults(1) else 0)
P
}
run() { python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-total --no-other-configs "$@" 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.readline()); print('  ', r['config']['kernel'], '|', r['roofline']['kernel_ms'], 'ms |', r['value'], 'GCUPS | MHz', (r.get('clock') or {}).get('sustained_mhz'), '| checksum', r['checksum'])"; }
lib() { echo $PWD/bgsa_amd/_ab/libbgsa_hip_$1.so; }
for v in ilp1 ilp2; do
  echo "== parity $v (default kernels, then resident Peq planes at 30/32 words)"
  BGSA_HIP_LIB=$(lib $v) parity
  BGSA_HIP_LIB=$(lib $v) BGSA_MYERS_PEQ_MAX_WORDS=32 parity
done
for i in 1 2; do
echo "== round $i: config 5 (1k x 1M x 1000 bp), code planes"
echo "as written";  run --config 5
echo "ilp 1,12";    BGSA_HIP_LIB=$(lib ilp1) run --config 5
echo "ilp 2,24";    BGSA_HIP_LIB=$(lib ilp2) run --config 5
echo "== round $i: config 5, resident Peq planes, chains in turns over 8 words"
echo "as written";  BGSA_MYERS_PEQ_MAX_WORDS=32 run --config 5
echo "ilp 1,12";    BGSA_HIP_LIB=$(lib ilp1) BGSA_MYERS_PEQ_MAX_WORDS=32 run --config 5
echo "ilp 2,24";    BGSA_HIP_LIB=$(lib ilp2) BGSA_MYERS_PEQ_MAX_WORDS=32 run --config 5
echo "== round $i: config 2 (4k x 1M x 150 bp)"
echo "as written";  run --config 2 --nq 4000
echo "ilp 1,12";    BGSA_HIP_LIB=$(lib ilp1) run --config 2 --nq 4000
echo "ilp 2,24";    BGSA_HIP_LIB=$(lib ilp2) run --config 2 --nq 4000
done
echo "== 768 bp (24 words: two waves per SIMD), 1k x 512k"
echo "as written";  run --config 2 --nq 1000 --ns 524288 --length 768
echo "ilp 1,12";    BGSA_HIP_LIB=$(lib ilp1) run --config 2 --nq 1000 --ns 524288 --length 768
echo "ilp 2,24";    BGSA_HIP_LIB=$(lib ilp2) run --config 2 --nq 1000 --ns 524288 --length 768
echo "== 512 bp (16 words: three waves per SIMD), 1k x 512k"
echo "as written";  run --config 2 --nq 1000 --ns 524288 --length 512
echo "ilp 1,12";    BGSA_HIP_LIB=$(lib ilp1) run --config 2 --nq 1000 --ns 524288 --length 512
echo "ilp 2,24";    BGSA_HIP_LIB=$(lib ilp2) run --config 2 --nq 1000 --ns 524288 --length 512
