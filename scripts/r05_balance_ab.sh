#!/bin/bash
# GPU box, round 5: how far apart the links of the Myers rows' carry chains sit.  As written phase A has five (planes: six)
# instructions from link to link and phase B three; `balanced` forms HP of word w + 1 inside phase B (four and four);
# schedule_ilp's third number asks for a minimum distance.  Libraries: scripts/build_variant.sh
#   bal  BGSA_GEN_MYERS_BALANCED=1 | bali  ... BGSA_GEN_MYERS_ILP=2,24,4 | i244 BGSA_GEN_MYERS_ILP=2,24,4 | i245 ...=2,24,5 | ilp2 ...=2,24
#     scripts/r05_balance_ab.sh > gpurun_out/r05_balance_ab.txt
set -e
cd "$(dirname "$0")/.."
echo "== scripts/ubench/chain_rate"
scripts/ubench/chain_rate 3000
parity() {
python3 - <<'P'
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, bgsa_amd as B, oracle as O
bad = 0
for qlen, slen in [(150, 150), (60, 33), (64, 64), (300, 512), (700, 768), (997, 1000), (1021, 1024), (950, 930), (200, 577), (500, 880)]:
    q = O.gen_reads(5000 + qlen, 11, qlen); s = O.gen_reads(6000 + slen, 200, slen)
    m = min(qlen, slen)
    s[:20, :m] = O.mutate(q[np.arange(20) % 11][:, :m], np.arange(20) % 7, slen)
    got = B.align_all_pairs(q, s, algo=B.ALGO_MYERS)
    ok = np.array_equal(got, O.myers64(q, s)); bad += not ok
    print(f"  parity {qlen}x{slen} {B.lib().bgsa_hip_kernel_name(B.ALGO_MYERS, (slen + 31) // 32).decode()}: {'ok' if ok else 'MISMATCH'}")
for length in (150, 1000, 1024):
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
for v in bal bali i244 i245; do
  echo "== parity $v"; BGSA_HIP_LIB=$(lib $v) parity; BGSA_HIP_LIB=$(lib $v) BGSA_MYERS_PEQ_MAX_WORDS=32 parity | grep -v "150x150\|60x33\|64x64"
done
for i in 1 2; do
echo "== round $i: config 5 (1k x 1M x 1000 bp), code planes"
echo "as written";            run --config 5
for v in bal bali i244 i245; do echo "$v"; BGSA_HIP_LIB=$(lib $v) run --config 5; done
echo "== round $i: config 5, resident Peq planes, chains in turns over 8 words"
echo "ilp2 (2,24,3)";         BGSA_HIP_LIB=$(lib ilp2) BGSA_MYERS_PEQ_MAX_WORDS=32 run --config 5
for v in bal bali i244 i245; do echo "$v"; BGSA_HIP_LIB=$(lib $v) BGSA_MYERS_PEQ_MAX_WORDS=32 run --config 5; done
echo "== round $i: config 2 (4k x 1M x 150 bp)"
echo "as written";            run --config 2 --nq 4000
for v in ilp2 bal bali i244 i245; do echo "$v"; BGSA_HIP_LIB=$(lib $v) run --config 2 --nq 4000; done
done
echo "== 768 bp (24 words), 1k x 512k"
echo "as written";            run --config 2 --nq 1000 --ns 524288 --length 768
for v in ilp2 bal bali i244 i245; do echo "$v"; BGSA_HIP_LIB=$(lib $v) run --config 2 --nq 1000 --ns 524288 --length 768; done
