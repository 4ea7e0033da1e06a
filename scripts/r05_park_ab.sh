#!/bin/bash
# GPU box, round 5: config 5 on resident Peq planes with the two carry chains in turns over K words — where the pausing chain
# waits (a scalar pair: s_mov_b64 through VCC; a vector register: two fast-class VALU instructions per switch), K = 8 / 11, and the
# dependency-aware order (schedule_ilp gap,window).  Against the code planes as written (the default until round 5).
#     scripts/r05_park_ab.sh > gpurun_out/r05_park_ab.txt
set -e
cd "$(dirname "$0")/.."
parity() {
python3 - <<'P'
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, bgsa_amd as B, oracle as O
bad = 0
for qlen, slen in [(997, 1000), (1021, 1024), (300, 961), (950, 930), (40, 897)]:
    q = O.gen_reads(5000 + qlen, 11, qlen); s = O.gen_reads(6000 + slen, 200, slen)
    m = min(qlen, slen)
    s[:20, :m] = O.mutate(q[np.arange(20) % 11][:, :m], np.arange(20) % 7, slen)
    got = B.align_all_pairs(q, s, algo=B.ALGO_MYERS)
    name = B.lib().bgsa_hip_kernel_name(B.ALGO_MYERS, (slen + 31) // 32).decode()
    ok = np.array_equal(got, O.myers64(q, s)); bad += not ok
    print(f"  parity {qlen}x{slen} {name}: {'ok' if ok else 'MISMATCH'}")
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
r=json.loads(sys.stdin.readline()); print('  ', r['config']['kernel'], '|', r['roofline']['kernel_ms'], 'ms |', r['value'], 'GCUPS | MHz', (r.get('clock') or {}).get('sustained_mhz'), '| checksum', r['checksum'])"; }
lib() { echo $PWD/bgsa_amd/_ab/libbgsa_hip_$1.so; }
for v in v8i2 v8i1 v11i2 s11i2; do
  echo "== parity $v"; BGSA_HIP_LIB=$(lib $v) BGSA_MYERS_PEQ_MAX_WORDS=32 BGSA_DYNAMIC_TASKS=0 parity
done
BGSA_HIP_LIB=$(lib v8i2) BGSA_MYERS_PEQ_MAX_WORDS=32 parity
for i in 1 2; do
echo "== round $i: config 5 (1k x 1M x 1000 bp)"
echo "code planes as written (counter)";            run
echo "scalar pair, K=8, ilp 2,24 (counter)";        BGSA_HIP_LIB=$(lib ilp2) BGSA_MYERS_PEQ_MAX_WORDS=32 run
echo "vector register, K=8, ilp 2,24 (counter)";    BGSA_HIP_LIB=$(lib v8i2) BGSA_MYERS_PEQ_MAX_WORDS=32 run
echo "vector register, K=8, ilp 1,12 (counter)";    BGSA_HIP_LIB=$(lib v8i1) BGSA_MYERS_PEQ_MAX_WORDS=32 run
echo "vector register, K=8, ilp 2,24 (static)";     BGSA_HIP_LIB=$(lib v8i2) BGSA_MYERS_PEQ_MAX_WORDS=32 BGSA_DYNAMIC_TASKS=0 run
echo "vector register, K=11, ilp 2,24 (static)";    BGSA_HIP_LIB=$(lib v11i2) BGSA_MYERS_PEQ_MAX_WORDS=32 BGSA_DYNAMIC_TASKS=0 run
echo "scalar pair, K=11, ilp 2,24 (static)";        BGSA_HIP_LIB=$(lib s11i2) BGSA_MYERS_PEQ_MAX_WORDS=32 BGSA_DYNAMIC_TASKS=0 run
done
