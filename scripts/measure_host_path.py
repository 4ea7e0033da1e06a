#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer seam (hip_cal_align_score): Peq upload + kernel + score
download per call, pageable host memory, one query bucket of REF_BUCKET_COUNT = 100 against a
1M-subject bucket (what cal_on_<arch> does per block, original/BGSA_CPU/cal_cpu.c:363-401)."""
import ctypes
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bgsa_amd as B

L = B.lib()
nq, ns, length = 100, 1_000_000 // 64 * 64, 150
rng = np.random.default_rng(0)
acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
s = acgt[rng.integers(0, 4, (ns, length))]
q = rng.integers(0, 4, (nq, length)).astype(np.uint8)  # already mapped 0..3
qbuf = np.full((nq, length + 1), ord("\n"), dtype=np.uint8); qbuf[:, :length] = q
sbuf = B.rows_to_buffer(s)
L.bgsa_hip_select_algorithm(B.ALGO_MYERS)
wn = B.word_num(B.ALGO_MYERS, length, length)
peq = np.zeros(B.group_words(B.ALGO_MYERS, wn) * (ns // 64), dtype=np.uint32)
seq = B.SeqT(len=length, size=sbuf.size, count=ns, extra_size=0, extra_count=0, content=sbuf.ctypes.data)
t0 = time.time(); L.hip_handle_reads(ctypes.byref(seq), peq.ctypes.data, wn, 0, ns); t_pre = time.time() - t0
out = np.zeros((nq, ns), dtype=np.int16)
args = (qbuf.ctypes.data, peq.ctypes.data, out.ctypes.data, length, nq, length, ns, 0, nq, wn, 27, None)
L.hip_cal_align_score(*args)  # warm-up (allocations)
t0 = time.time()
for _ in range(3):
    L.hip_cal_align_score(*args)
dt = (time.time() - t0) / 3
cells = nq * ns * length * length
print(f"host preprocess (hip_handle_reads, {ns} reads): {t_pre*1e3:.1f} ms")
print(f"hip_cal_align_score 100 x {ns}: {dt*1e3:.1f} ms/call = {cells/dt/1e9:.0f} GCUPS PCIe-inclusive "
      f"({peq.nbytes/1e6:.0f} MB up, {out.nbytes/1e6:.0f} MB down per call)")
