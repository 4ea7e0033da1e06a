#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer seam (hip_cal_align_score): one query bucket of
REF_BUCKET_COUNT = 100 against a 1M-subject bucket (what cal_on_<arch> does per block,
original/BGSA_CPU/cal_cpu.c:363-401), with pageable or malloc_mem() buffers and with the bucket
re-uploaded per call or kept resident (include/bgsa_hip.h "Resident buckets")."""
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
seq = B.SeqT(len=length, size=sbuf.size, count=ns, extra_size=0, extra_count=0, content=sbuf.ctypes.data)
cells = nq * ns * length * length


def buffers(pinned):
    """Peq and result buffers the way the reference allocates them: malloc_mem() (page-locked from this
    library) or plain pageable memory."""
    n_peq = B.group_words(B.ALGO_MYERS, wn) * (ns // 64)
    if not pinned:
        return np.zeros(n_peq, dtype=np.uint32), np.zeros((nq, ns), dtype=np.int16), None
    p1, p2 = L.malloc_mem(n_peq * 4), L.malloc_mem(nq * ns * 2)
    peq = np.ctypeslib.as_array(ctypes.cast(p1, ctypes.POINTER(ctypes.c_uint32)), shape=(n_peq,))
    out = np.ctypeslib.as_array(ctypes.cast(p2, ctypes.POINTER(ctypes.c_int16)), shape=(nq, ns))
    peq[:] = 0
    return peq, out, (p1, p2)


for pinned in (False, True):
    for resident in (False, True):
        L.bgsa_hip_set_auto_resident(1 if resident else 0)
        peq, out, raw = buffers(pinned)
        t0 = time.time(); L.hip_handle_reads(ctypes.byref(seq), peq.ctypes.data, wn, 0, ns); t_pre = time.time() - t0
        args = (qbuf.ctypes.data, peq.ctypes.data, out.ctypes.data, length, nq, length, ns, 0, nq, wn, 27, None)
        t0 = time.time(); L.hip_cal_align_score(*args); first = time.time() - t0
        t0 = time.time()
        for _ in range(5):
            L.hip_cal_align_score(*args)
        dt = (time.time() - t0) / 5
        print(f"buffers {'malloc_mem (pinned)' if pinned else 'pageable'}, bucket {'resident' if resident else 're-uploaded per call'}: "
              f"hip_handle_reads {t_pre*1e3:.1f} ms, first call {first*1e3:.1f} ms, steady {dt*1e3:.2f} ms/call = "
              f"{cells/dt/1e9:.0f} GCUPS host-to-host (100 queries x {ns} subjects; {peq.nbytes/1e6:.0f} MB Peq, {out.nbytes/1e6:.0f} MB scores)")
        L.bgsa_hip_bucket_release(None)
        if raw:
            L.free_mem(raw[0]); L.free_mem(raw[1])
L.bgsa_hip_set_auto_resident(1)
