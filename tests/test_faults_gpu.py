"""GPU tests of the stream-address and stream-fault handling.

1. Regression for the memory access fault recorded in round 1: the first generated-asm
   kernel built the 64-bit stream address as `int readfirstlane(lo) | (u64(hi) << 32)`, which sign-extends
   the low half — any workspace whose address had bit 31 set became 0xffffffffXXXXXXXX and the first scalar
   load faulted.  Here every buffer of every kernel family is placed at addresses with bit 31 set (and,
   for contrast, clear), through the C ABI.
2. The sticky fault word: a stream damaged between the packer and the row loop (fault injection) is
   reported by bgsa_hip_stream_faults() instead of producing silent wrong scores, for every loop shape.
"""
import ctypes

import numpy as np
import pytest

import bgsa_amd as B

pytestmark = pytest.mark.gpu

FAMILIES = [  # (name, algo, qlen, slen, k, scores)
    ("myers asm 5 words", B.ALGO_MYERS, 150, 150, 0, None),
    ("myers two rows per token", B.ALGO_MYERS, 57, 40, 0, None),
    ("myers code planes", B.ALGO_MYERS, 90, 1000, 0, None),
    ("myers column blocks", B.ALGO_MYERS, 70, 2500, 0, None),
    ("bitpal plain", B.ALGO_BITPAL, 150, 150, 0, (2, -3, -5)),
    ("bitpal column blocks", B.ALGO_BITPAL, 64, 600, 0, (2, -3, -5)),
    ("banded 32-bit band", B.ALGO_BANDED, 150, 150, 8, None),
    ("banded 64-bit band", B.ALGO_BANDED, 150, 150, 20, None),
]


def _want(oracle, algo, q, s, k):
    return {B.ALGO_MYERS: lambda: oracle.myers64(q, s), B.ALGO_BITPAL: lambda: oracle.bitpal(q, s),
            B.ALGO_BANDED: lambda: oracle.banded64(q, s, k)}[algo]()


class Arena:
    """Carves 4 KiB-aligned device buffers out of one big allocation, starting at an address whose
    bit 31 is `high`."""

    def __init__(self, torch, high: bool):
        self.buf = torch.empty(6 << 30, dtype=torch.uint8, device="cuda:0")
        base = self.buf.data_ptr()
        target = 0x80000000 if high else 0x10000000
        self.off = (target - (base & 0xFFFFFFFF)) % (1 << 32)
        self.high = high

    def take(self, nbytes: int):
        t = self.buf[self.off:self.off + nbytes]
        assert ((t.data_ptr() >> 31) & 1) == (1 if self.high else 0)
        assert (((t.data_ptr() + nbytes) >> 31) & 1) == (1 if self.high else 0)
        self.off += (nbytes + 4095) // 4096 * 4096
        return t


@pytest.mark.parametrize("high", [True, False])
def test_buffers_at_addresses_with_bit_31_set(oracle, high):
    import torch
    L = B.lib()
    arena = Arena(torch, high)
    for name, algo, qlen, slen, k, scores in FAMILIES:
        q = oracle.gen_reads(11 + qlen, 9, qlen)
        s = oracle.gen_reads(12 + slen, 192, slen)
        m = min(qlen, slen)
        s[:9, :m] = oracle.mutate(q[:, :m], np.arange(9), 13)
        p = B.Params(algo, 0, *(scores or (0, -1, -1)), k)
        qbuf = B.rows_to_buffer(q)
        d_content = arena.take(qbuf.size + 8)
        d_content.zero_()
        d_content[:qbuf.size].copy_(torch.from_numpy(qbuf))
        B.check(L.bgsa_hip_map_queries_dev(d_content.data_ptr(), qbuf.size, None))
        sbuf = B.rows_to_buffer(s)
        d_rows = arena.take(sbuf.size)
        d_rows.copy_(torch.from_numpy(sbuf))
        wn = B.word_num(algo, qlen, slen, k)
        d_peq = arena.take(B.group_words(algo, wn, k) * 3 * 4)
        torch.cuda.synchronize()
        B.check(L.bgsa_hip_handle_reads_dev(algo, d_rows.data_ptr(), d_rows.numel(), slen, 192, wn, k, d_peq.data_ptr(), None))
        esz = 1 if algo == B.ALGO_BANDED else 2
        d_out = arena.take(9 * 192 * esz)
        need = int(L.bgsa_hip_workspace_bytes_ex(ctypes.byref(p), qlen, slen, 9))
        d_work = arena.take(need)
        B.check(L.bgsa_hip_cal_align_score_ex(ctypes.byref(p), d_content.data_ptr(), d_peq.data_ptr(), d_out.data_ptr(),
                                              qlen, slen, 192, 0, 9, wn, d_work.data_ptr(), need, None), name)
        torch.cuda.synchronize()
        assert L.bgsa_hip_stream_faults(1) == 0, name
        got = d_out.cpu().numpy().view(np.int8 if esz == 1 else np.int16).reshape(9, 192)
        assert np.array_equal(got, _want(oracle, algo, q, s, k)), name


@pytest.mark.parametrize("kind", [1, 2])
@pytest.mark.parametrize("family", FAMILIES, ids=[f[0] for f in FAMILIES])
def test_damaged_stream_is_reported_not_scored_silently(oracle, family, kind):
    import torch
    L = B.lib()
    name, algo, qlen, slen, k, scores = family
    q = oracle.gen_reads(21 + qlen, 6, qlen)
    s = oracle.gen_reads(22 + slen, 128, slen)
    a = B.DeviceAligner(algo, k=k, scores=scores)
    a.set_queries(q)
    a.set_subjects(s)
    want = _want(oracle, algo, q, s, k)
    torch.cuda.synchronize()
    assert L.bgsa_hip_stream_faults(1) == 0
    assert L.bgsa_hip_debug_inject_stream_fault(kind) == 0
    out = a.score()
    with pytest.raises(B.BgsaHipError, match="stream fault"):
        a.check_faults()
    msg = L.bgsa_hip_last_error()
    # kind 2 (a byte that is no token) only exists where the dispatch mask is wider than the slot table
    has_fail_slots = name in ("myers asm 5 words", "myers code planes", "bitpal plain", "banded 32-bit band", "banded 64-bit band")
    assert (b"no stream code" in msg) if (kind == 2 and has_fail_slots) else (b"budget" in msg)
    # only query 0's stream was damaged: every other row of the tile is intact
    assert np.array_equal(out[1:, :128].cpu().numpy(), want[1:])
    # the word is sticky until cleared, then the next call is clean and correct
    assert L.bgsa_hip_stream_faults(0) == 0
    out = a.score()
    a.check_faults()
    assert np.array_equal(out[:, :128].cpu().numpy(), want)


def test_host_seam_dies_loudly_on_a_stream_fault(oracle, tmp_path):
    """hip_cal_align_score follows the reference's convention — print and exit(1) — when the device
    reports a fault; run in a child process."""
    import subprocess
    import sys
    from pathlib import Path
    root = Path(B.__file__).resolve().parent.parent
    script = r'''
import sys, ctypes, numpy as np
sys.path.insert(0, sys.argv[1])
import bgsa_amd as B, oracle as O
L = B.lib()
q = O.gen_reads(1, 4, 150); s, _ = B.pad_rows(O.gen_reads(2, 64, 150))
L.bgsa_hip_select_algorithm(B.ALGO_MYERS); L.init_mapping_table()
table = np.ctypeslib.as_array((ctypes.c_uint32 * 128).in_dll(L, "mapping_table"))
sbuf = B.rows_to_buffer(s)
seq = B.SeqT(len=150, size=sbuf.size, count=64, extra_size=0, extra_count=0, content=sbuf.ctypes.data)
peq = np.zeros(B.group_words(B.ALGO_MYERS, 5), dtype=np.uint32)
L.hip_handle_reads(ctypes.byref(seq), peq.ctypes.data, 5, 0, 64)
qbuf = B.rows_to_buffer(q); keep = qbuf == 10; qm = table[qbuf].astype(np.uint8); qm[keep] = 10
out = np.zeros((4, 64), dtype=np.int16)
L.hip_cal_align_score(qm.ctypes.data, peq.ctypes.data, out.ctypes.data, 150, 4, 150, 64, 0, 4, 5, 27, None)
assert np.array_equal(out, O.myers64(q, s)); print("clean call ok", flush=True)
L.bgsa_hip_debug_inject_stream_fault(1)
L.hip_cal_align_score(qm.ctypes.data, peq.ctypes.data, out.ctypes.data, 150, 4, 150, 64, 0, 4, 5, 27, None)
print("NOT REACHED")
'''
    p = subprocess.run([sys.executable, "-c", script, str(root)], capture_output=True, text=True, timeout=300)
    assert p.returncode == 1, p.stdout + p.stderr
    assert "clean call ok" in p.stdout and "NOT REACHED" not in p.stdout
    assert "Error - hip_cal_align_score" in p.stdout and "stream fault" in p.stdout


def test_concurrent_callers_with_library_scratch(oracle):
    """Two host threads, two streams, different algorithms and scores, both letting the library own the scratch
    (d_workspace = NULL): the scratch is per (device, stream) and the parameters travel with the call, so neither
    sees the other's streams or settings (process-global scratch and settings raced)."""
    import threading
    import torch
    L = B.lib()
    q = oracle.gen_reads(91, 24, 150)
    s = oracle.gen_reads(92, 640, 150)
    s[:24] = oracle.mutate(q, np.arange(24) % 7, 93)
    want = {B.ALGO_MYERS: oracle.myers64(q, s), B.ALGO_BITPAL: oracle.bitpal(q, s), B.ALGO_BANDED: oracle.banded64(q, s, 8)}
    errors = []

    def worker(algo):
        try:
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                a = B.DeviceAligner(algo, k=8)
                a.set_queries(q)
                a.set_subjects(s)
                p = a.params()
                out = torch.empty((24, a.ns), dtype=a.out_dtype, device="cuda:0")
                for _ in range(40):
                    out.zero_()
                    B.check(L.bgsa_hip_cal_align_score_ex(ctypes.byref(p), a.d_content.data_ptr(), a.d_peq.data_ptr(), out.data_ptr(),
                                                          150, 150, a.ns, 0, 24, a.wn, None, 0, ctypes.c_void_p(stream.cuda_stream)))
                    stream.synchronize()
                    if not np.array_equal(out[:, :640].cpu().numpy(), want[algo]):
                        errors.append(f"algo {algo}: wrong scores")
                        return
        except Exception as e:   # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(algo,)) for algo in (B.ALGO_MYERS, B.ALGO_BITPAL, B.ALGO_BANDED)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert L.bgsa_hip_stream_faults(1) == 0
