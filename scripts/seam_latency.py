#!/usr/bin/env python3
"""Latency of the fine seam (align_hip) per call, single host thread: first call for a query (row miss: one
launch over the whole resident bucket) and the following calls for the same query (served from its row)."""
import ctypes
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bgsa_amd as B

L = B.lib()
ns, length, nq = 1_000_000 // 64 * 64, 150, int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(0)
acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
s = acgt[rng.integers(0, 4, (ns, length))]
qbuf = np.full((nq, length + 1), ord("\n"), dtype=np.uint8)
qbuf[:, :length] = rng.integers(0, 4, (nq, length))
sbuf = B.rows_to_buffer(s)
L.bgsa_hip_select_algorithm(B.ALGO_MYERS)
wn = 5
gw = B.group_words(B.ALGO_MYERS, wn)
n_peq = gw * (ns // 64)
p1, p2 = L.malloc_mem(n_peq * 4), L.malloc_mem(min(nq, 100) * ns * 2)
peq = np.ctypeslib.as_array(ctypes.cast(p1, ctypes.POINTER(ctypes.c_uint32)), shape=(n_peq,))
out = np.ctypeslib.as_array(ctypes.cast(p2, ctypes.POINTER(ctypes.c_int16)), shape=(min(nq, 100), ns))
peq[:] = 0
seq = B.SeqT(len=length, size=sbuf.size, count=ns, extra_size=0, extra_count=0, content=sbuf.ctypes.data)
L.hip_handle_reads(ctypes.byref(seq), peq.ctypes.data, wn, 0, ns)
chunk, groups = 27, ns // 64
miss, hit = [], []
for i in range(nq):
    row = qbuf[i].ctypes.data
    for n, j in enumerate(range(0, groups, chunk)):
        c = min(chunk, groups - j)
        t0 = time.perf_counter()
        L.align_hip(row, peq[gw * j:].ctypes.data, length, length, wn, c, (i % 100) * groups + j, out.ctypes.data, None)
        (miss if n == 0 else hit).append(time.perf_counter() - t0)
print(f"align_hip, 1M-subject resident bucket: first call per query {np.median(miss[1:])*1e3:.2f} ms (median), "
      f"following calls {np.median(hit)*1e6:.1f} us (median), {np.mean(hit)*1e6:.1f} us (mean); "
      f"{len(hit) // nq + 1} calls per query")
if nq >= 200:
    m = np.array(miss) * 1e3
    print("first-call ms by query index:", " ".join(f"{i}:{np.median(m[i:i + 20]):.2f}" for i in range(0, nq, max(nq // 10, 20))))
