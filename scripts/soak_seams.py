#!/usr/bin/env python3
"""Randomised soak of the HOST seams on the GPU box (not part of pytest): hip_handle_reads + hip_cal_align_score
(query windows, page-locked or pageable buffers, blocks big enough for the tiled copy-out or not) and align_hip
(a malloc_mem query buffer walked in order, backwards or at random, whole-bucket or chunked calls), every result
compared with the oracle.

    python3 scripts/soak_seams.py [seconds] [seed]
"""
import ctypes
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bgsa_amd as B  # noqa: E402
import oracle as O  # noqa: E402


def main() -> int:
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    L = B.lib()
    L.init_mapping_table()
    table = np.ctypeslib.as_array((ctypes.c_uint32 * 128).in_dll(L, "mapping_table"))
    threshold = ctypes.c_int.in_dll(L, "threshold")
    t0 = last = time.time()
    done = {"cal": 0, "align": 0}
    case = 0
    while time.time() - t0 < budget:
        case += 1
        algo = int(rng.choice([B.ALGO_MYERS, B.ALGO_BANDED, B.ALGO_BITPAL]))
        big = rng.random() < 0.15
        k = int(rng.integers(1, 32)) if algo == B.ALGO_BANDED else 0
        if algo == B.ALGO_BANDED:
            length = int(rng.integers(2 * k + 2, 2 * k + 200))
            while 1 <= length % 64 <= k:
                length += 1
            qlen = slen = length
        else:
            qlen, slen = int(rng.integers(1, 260)), int(rng.integers(1, 260))
        nq = int(rng.integers(1, 160))
        groups = int(rng.integers(600, 1100)) if big else int(rng.integers(1, 40))
        n = 64 * groups
        q = O.gen_reads(int(rng.integers(1 << 30)), nq, qlen)
        s = O.gen_reads(int(rng.integers(1 << 30)), n, slen)
        m = min(qlen, slen)
        rel = min(n, 48)
        s[:rel, :m] = O.mutate(q[np.arange(rel) % nq][:, :m], rng.integers(0, 10, rel), int(rng.integers(1 << 30)))
        fn = {B.ALGO_MYERS: O.myers64, B.ALGO_BITPAL: O.bitpal, B.ALGO_BANDED: lambda x, y: O.banded64(x, y, k)}[algo]
        check_cols = n if not big else 256              # the oracle on the whole matrix only for the small cases
        want = fn(q, s[:check_cols])
        esz = 1 if algo == B.ALGO_BANDED else 2
        dtype = np.int8 if esz == 1 else np.int16
        L.bgsa_hip_select_algorithm(algo)
        threshold.value = k if algo == B.ALGO_BANDED else 31
        sbuf = B.rows_to_buffer(s)
        seq = B.SeqT(len=slen, size=sbuf.size, count=n, extra_size=0, extra_count=0, content=sbuf.ctypes.data)
        qb = B.rows_to_buffer(q)
        keep = qb == 10
        qm = table[qb].astype(np.uint8)
        qm[keep] = 10
        wn = B.word_num(algo, qlen, slen, k)
        gw = B.group_words(algo, wn, k)
        pinned = rng.random() < 0.6
        blocks = []
        if pinned:
            p1, p2 = L.malloc_mem(gw * groups * 4), L.malloc_mem(max(nq * n * esz, 8))
            blocks = [p1, p2]
            peq = np.ctypeslib.as_array(ctypes.cast(p1, ctypes.POINTER(ctypes.c_uint32)), shape=(gw * groups,))
            out = np.ctypeslib.as_array(ctypes.cast(p2, ctypes.POINTER(ctypes.c_int8 if esz == 1 else ctypes.c_int16)), shape=(nq, n))
        else:
            peq, out = np.zeros(gw * groups, dtype=np.uint32), np.zeros((nq, n), dtype=dtype)
        qblock = L.malloc_mem(max(qm.size + 64, 1 << (16 if rng.random() < 0.5 else 21)))
        blocks.append(qblock)
        qbuf = np.ctypeslib.as_array(ctypes.cast(qblock, ctypes.POINTER(ctypes.c_uint8)), shape=(qm.size + 64,))
        qbuf[:] = 0
        qbuf[: qm.size] = qm
        try:
            peq[:] = 0
            L.hip_handle_reads(ctypes.byref(seq), peq.ctypes.data, wn, 0, n)
            # the coarse seam on a random query window
            lo = int(rng.integers(0, nq))
            hi = int(rng.integers(lo + 1, nq + 1))
            out[:] = 99
            L.hip_cal_align_score(qblock, peq.ctypes.data, out.ctypes.data, qlen, nq, slen, n, lo, hi, wn, 27, None)
            ok = np.array_equal(out[: hi - lo, :check_cols], want[lo:hi]) and bool((out[hi - lo:] == 99).all())
            done["cal"] += 1
            # the fine seam: some queries of the buffer, in a random visiting order, whole bucket or chunks
            if ok:
                order = {0: np.arange(nq), 1: np.arange(nq)[::-1], 2: rng.permutation(nq)}[int(rng.integers(0, 3))][: int(rng.integers(1, min(nq, 50) + 1))]
                res = np.zeros(n, dtype=dtype)
                for i in order:
                    j = int(rng.integers(0, groups)) if rng.random() < 0.5 else 0
                    c = int(rng.integers(1, groups - j + 1))
                    res[:] = 77
                    L.align_hip(qblock + int(i) * (qlen + 1), peq[gw * j:].ctypes.data, qlen, slen, wn, c, j, res.ctypes.data, None)
                    hi_col = min(64 * (j + c), check_cols)
                    if 64 * j < hi_col and not np.array_equal(res[64 * j: hi_col], want[i, 64 * j: hi_col]):
                        ok = False
                        break
                    if (res[: 64 * j] != 77).any() or (res[64 * (j + c):] != 77).any():
                        ok = False
                        break
                done["align"] += len(order)
            # every third case: the registered bucket is overwritten behind the library's back — another preprocessed
            # bucket memmove'd over it, no call tells the library — and both seams must score the NEW content
            # (every scoring call fingerprints the resident range)
            if ok and case % 3 == 0:
                s2 = O.gen_reads(int(rng.integers(1 << 30)), n, slen)
                s2[:rel, :m] = O.mutate(q[np.arange(rel) % nq][:, :m], rng.integers(0, 10, rel), int(rng.integers(1 << 30)))
                want2 = fn(q, s2[:check_cols])
                sbuf2 = B.rows_to_buffer(s2)
                seq2 = B.SeqT(len=slen, size=sbuf2.size, count=n, extra_size=0, extra_count=0, content=sbuf2.ctypes.data)
                peq2 = np.zeros(gw * groups, dtype=np.uint32)
                L.hip_handle_reads(ctypes.byref(seq2), peq2.ctypes.data, wn, 0, n)
                before = ctypes.c_uint64()
                L.bgsa_hip_stale_ranges(ctypes.byref(before))
                ctypes.memmove(peq.ctypes.data, peq2.ctypes.data, peq2.nbytes)
                fine_first = rng.random() < 0.5
                for which in (("fine", "coarse") if fine_first else ("coarse", "fine")):
                    if which == "coarse":
                        out[:] = 99
                        L.hip_cal_align_score(qblock, peq.ctypes.data, out.ctypes.data, qlen, nq, slen, n, lo, hi, wn, 27, None)
                        ok = ok and np.array_equal(out[: hi - lo, :check_cols], want2[lo:hi])
                    else:
                        res = np.zeros(n, dtype=dtype)
                        for i in order[:8]:
                            L.align_hip(qblock + int(i) * (qlen + 1), peq.ctypes.data, qlen, slen, wn, groups, 0, res.ctypes.data, None)
                            ok = ok and np.array_equal(res[:check_cols], want2[i])
                after = ctypes.c_uint64()
                L.bgsa_hip_stale_ranges(ctypes.byref(after))
                ok = ok and after.value == before.value + 1      # detected once, by whichever seam came first
                done["rewritten"] = done.get("rewritten", 0) + 1
            if not ok:
                print(f"MISMATCH case {case}: algo {algo} k {k} qlen {qlen} slen {slen} nq {nq} groups {groups} pinned {pinned} window {lo}:{hi}", flush=True)
                return 1
        finally:
            L.bgsa_hip_bucket_release(None)
            for p in blocks:
                L.free_mem(p)
        if time.time() - last > 20:
            last = time.time()
            print(f"... {case} cases, {done}", flush=True)
    L.bgsa_hip_select_algorithm(B.ALGO_MYERS)
    print(f"seam soak ok: {case} cases in {time.time() - t0:.0f} s: {done}", flush=True)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
