"""Host-only checks of the packed query streams the row loops walk (no GPU).

The generated row loops (bgsa_amd/csrc/gen_rows_asm.py) fetch a query as 8-byte windows through the
scalar cache, one window ahead, under a window budget of `stride/8 - 2` REFILLs.  These tests walk every
stream the library can produce exactly the way the loop does and check that
  * the walk ends on an END token with the budget exactly used up (so the budget never fires on a
    well-formed stream and always fires on one that lost its END),
  * no byte the walk dispatches is outside the loop's slot table,
  * the last byte the loop can touch (it loads window i+1 while it runs window i) lies inside the stream's
    stride, and `bgsa_hip_workspace_bytes()` covers n_queries strides.
"""
import ctypes

import numpy as np
import pytest

import bgsa_amd as B

# slot tables of the four loop shapes (gen_rows_asm.py): code -> what the dispatch does
PLAIN = {"mask": 7, "rows": {c: 1 for c in range(5)}, "end": 5, "refill": 6, "fail": {7}}
BLOCKED = {"mask": 7, "rows": {c: 1 for c in range(5)}, "end": 5, "refill": 6, "carry": 7, "fail": set()}
PAIR = {"mask": 0x1F, "rows": {**{c: 2 for c in range(25)}, **{c: 1 for c in range(25, 30)}}, "end": 30, "refill": 31,
        "fail": set()}
BANDED = {"mask": 0x3F, "rows": {**{c: 2 for c in range(25)}, **{c: 1 for c in range(25, 30)}}, "end": 30, "refill": 31,
          "event": 63, "fail": set(range(32, 63))}


@pytest.fixture(scope="module")
def L():
    return B.lib()


def _stream(L, algo, row, k):
    n = L.bgsa_hip_query_stream(algo, row.ctypes.data, len(row), k, None, 0)
    buf = np.full(n + 64, 0xEE, dtype=np.uint8)          # canary behind the stream
    assert L.bgsa_hip_query_stream(algo, row.ctypes.data, len(row), k, buf.ctypes.data, n) == n
    assert (buf[n:] == 0xEE).all()
    return buf[:n]


def walk(raw, table):
    """Returns (rows consumed, budget left at END, last byte offset the loop loaded)."""
    stride = len(raw)
    assert stride % 8 == 0
    budget = stride // 8 - 2
    ptr, touched = 0, 16            # prologue: loads [ptr, ptr+8) and [ptr+8, ptr+16)
    win = list(raw[0:8])
    nxt = list(raw[8:16])
    rows = 0
    while True:
        assert win, "ran off a window without REFILL"
        code = win.pop(0) & table["mask"]
        assert code not in table["fail"], f"byte {code} is no token"
        if code in table["rows"]:
            rows += table["rows"][code]
        elif code == table["end"]:
            return rows, budget, touched
        elif code == table["refill"]:
            budget -= 1
            assert budget >= 0, "window budget exhausted before END"
            win = nxt
            ptr += 8
            assert ptr + 16 <= stride, "prefetch beyond the stream's stride"
            nxt = list(raw[ptr + 8:ptr + 16])
            touched = max(touched, ptr + 16)
        elif code == table.get("carry", -1):
            pass
        elif code == table.get("event", -1):
            assert win, "EVENT argument byte in the next window"
            win.pop(0)
        else:
            raise AssertionError(f"code {code} has no slot")


@pytest.mark.parametrize("lo,hi", [(1, 600), (600, 1400), (1400, 2800), (2800, 4101)])
def test_plain_pair_and_blocked_streams_stay_inside_their_stride(L, lo, hi):
    rng = np.random.default_rng(lo)
    for length in range(lo, hi):
        row = rng.integers(0, 5, length).astype(np.uint8)
        for k, table in ((0, PLAIN), (-2, PAIR), (-1, BLOCKED)):
            raw = _stream(L, B.ALGO_MYERS, row, k)
            rows, left, touched = walk(raw, table)
            assert rows == length and left == 0 and touched <= len(raw), (length, k)


@pytest.mark.parametrize("k", [1, 4, 8, 15, 16, 31])
def test_banded_streams_stay_inside_their_stride(L, k):
    rng = np.random.default_rng(k)
    for length in list(range(2 * k + 2, 700)) + [1000, 1023, 1024, 1025, 2000, 4000, 4100]:
        row = rng.integers(0, 5, length).astype(np.uint8)
        raw = _stream(L, B.ALGO_BANDED, row, k)
        rows, left, touched = walk(raw, BANDED)
        assert rows == length and left == 0 and touched <= len(raw), (length, k)
        # the workspace is sized without knowing k: the bound must cover every k
        assert L.bgsa_hip_workspace_bytes(B.ALGO_BANDED, length, length, 3) >= 3 * len(raw)


def test_workspace_covers_every_stream(L):
    L.bgsa_hip_select_algorithm(B.ALGO_MYERS)
    row = np.zeros(4100, dtype=np.uint8)
    for length in range(1, 4101):
        plain = L.bgsa_hip_query_stream(B.ALGO_MYERS, row.ctypes.data, length, 0, None, 0)
        pair = L.bgsa_hip_query_stream(B.ALGO_MYERS, row.ctypes.data, length, -2, None, 0)
        blocked = L.bgsa_hip_query_stream(B.ALGO_MYERS, row.ctypes.data, length, -1, None, 0)
        assert pair <= plain                           # <= 64 bp subjects share the plain sizing
        for nq in (1, 7):
            assert L.bgsa_hip_workspace_bytes(B.ALGO_MYERS, length, 150, nq) >= nq * plain
            assert L.bgsa_hip_workspace_bytes(B.ALGO_MYERS, length, 40, nq) >= nq * pair
            assert L.bgsa_hip_workspace_bytes(B.ALGO_MYERS, length, 2000, nq) >= nq * blocked
            assert L.bgsa_hip_workspace_bytes(B.ALGO_BITPAL, length, 150, nq) >= nq * plain
            assert L.bgsa_hip_workspace_bytes(B.ALGO_BITPAL, length, 2000, nq) >= nq * blocked


def test_a_stream_without_end_exhausts_the_budget(L):
    """What the fault word reports on the GPU, on the host model: all-REFILL bytes trip the budget after
    exactly stride/8 - 2 windows — before the prefetch can leave the stride — and a byte outside the table
    is caught by a fail slot."""
    row = np.zeros(150, dtype=np.uint8)
    for k, table, refill in ((0, PLAIN, 6), (-2, PAIR, 31), (-1, BLOCKED, 6)):
        raw = _stream(L, B.ALGO_MYERS, row, k).copy()
        raw[:] = refill
        with pytest.raises(AssertionError, match="budget"):
            walk(raw, table)
    raw = _stream(L, B.ALGO_BANDED, row, 8).copy()
    raw[:] = 31
    with pytest.raises(AssertionError, match="budget"):
        walk(raw, BANDED)
    raw[:] = 40
    with pytest.raises(AssertionError, match="no token"):
        walk(raw, BANDED)
    raw = _stream(L, B.ALGO_MYERS, row, 0).copy()
    raw[:] = 7
    with pytest.raises(AssertionError, match="no token"):
        walk(raw, PLAIN)


def test_fault_api_is_exported_and_validates(L):
    assert L.bgsa_hip_debug_inject_stream_fault(3) != 0
    assert L.bgsa_hip_debug_inject_stream_fault(0) == 0
    p = B.Params()
    assert L.bgsa_hip_current_params(ctypes.byref(p)) == 0
    assert (p.algo, p.alignment) == (L.bgsa_hip_current_algorithm(), L.bgsa_hip_current_alignment())
