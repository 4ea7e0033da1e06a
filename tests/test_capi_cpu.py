"""CPU-side checks of the C-ABI library: it loads, exports every declared symbol, and its host
logic (word counts, host preprocess layout) is right.  No compute entry point is called."""
import ctypes

import numpy as np
import pytest

import bgsa_amd as B


@pytest.fixture(scope="module")
def L():
    if not B.LIB_PATH.exists():
        B.build_library()
    return B.lib()


def test_exports_every_declared_symbol(L):
    names = B.declared_symbols()
    assert len(names) >= 20
    for fn in ("hip_handle_reads", "align_hip", "hip_cal_align_score", "init_mapping_table",
               "bgsa_hip_cal_align_score_dev", "bgsa_hip_handle_reads_dev"):
        assert fn in names
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, f"declared in include/bgsa_hip.h but not exported: {missing}"
    for var in ("match_score", "mismatch_score", "gap_score", "dvdh_len", "full_bits", "threshold",
                "cpu_threads", "mapping_table"):
        ctypes.c_int.in_dll(L, var)


def test_algorithm_globals_follow_selection(L):
    # the five ints of align_core.c:13-17 track the selected algorithm
    assert L.bgsa_hip_select_algorithm(B.ALGO_BITPAL) == 0
    assert [ctypes.c_int.in_dll(L, v).value for v in ("match_score", "mismatch_score", "gap_score")] == [2, -3, -5]
    assert L.bgsa_hip_select_algorithm(B.ALGO_MYERS) == 0
    assert [ctypes.c_int.in_dll(L, v).value for v in ("match_score", "mismatch_score", "gap_score")] == [0, -1, -1]
    assert ctypes.c_int.in_dll(L, "full_bits").value == 1
    assert L.bgsa_hip_select_algorithm(99) != 0
    assert L.bgsa_hip_current_algorithm() == B.ALGO_MYERS


def test_word_num(L):
    assert B.word_num(B.ALGO_MYERS, 150, 150) == 5
    assert B.word_num(B.ALGO_MYERS, 1000, 1000) == 32
    assert B.word_num(B.ALGO_MYERS, 1, 32) == 1 and B.word_num(B.ALGO_MYERS, 1, 33) == 2
    assert B.word_num(B.ALGO_BANDED, 150, 150, 8) == 8   # 32-bit words of the offset match string + 3 spare
    assert B.word_num(B.ALGO_BANDED, 150, 150, 16) == 8  # same layout for the 64-bit band
    assert B.group_words(B.ALGO_MYERS, 5) == 5 * 5 * 64
    assert B.group_words(B.ALGO_BANDED, 8, 8) == 5 * 8 * 64
    assert B.group_words(B.ALGO_BANDED, 8, 16) == 5 * 8 * 64


def test_mapping_table(L):
    L.init_mapping_table()
    table = (ctypes.c_uint32 * 128).in_dll(L, "mapping_table")
    assert [table[ord(c)] for c in "ACGTN"] == [0, 1, 2, 3, 4]
    assert table[ord("a")] == 0 and table[ord("X")] == 0 and table[ord("\n")] == 0


def _host_preprocess(L, algo, rows, k=0, qlen=None):
    rows, _ = B.pad_rows(rows)
    n, length = rows.shape
    buf = B.rows_to_buffer(rows)
    wn = B.word_num(algo, length if qlen is None else qlen, length, k)
    out = np.zeros(B.group_words(algo, wn, k) * (n // 64), dtype=np.uint32)
    seq = B.SeqT(len=length, size=buf.size, count=n, extra_size=0, extra_count=0, content=buf.ctypes.data)
    assert L.bgsa_hip_select_algorithm(algo) == 0
    ctypes.c_int.in_dll(L, "threshold").value = k
    ctypes.c_int.in_dll(L, "cpu_threads").value = 3
    L.hip_handle_reads(ctypes.byref(seq), out.ctypes.data, wn, 0, n)
    L.bgsa_hip_select_algorithm(B.ALGO_MYERS)
    return rows, out, wn


CODE = {ord("A"): 0, ord("C"): 1, ord("G"): 2, ord("T"): 3, ord("N"): 4}


def test_host_handle_reads_layout_myers(L, oracle):
    rows = oracle.gen_reads(77, 100, 150)
    rows[3, 10:20] = ord("N")
    rows, peq, wn = _host_preprocess(L, B.ALGO_MYERS, rows)
    peq = peq.reshape(-1, 5, wn, 64)  # [group][char][word][lane]
    for s in (0, 3, 63, 64, 99, 127):
        g, lane = divmod(s, 64)
        for p in range(150):
            c = CODE[rows[s, p]]
            for cc in range(5):
                bit = (int(peq[g, cc, p // 32, lane]) >> (p % 32)) & 1
                assert bit == (cc == c)
    # every column is claimed by exactly one plane; nothing beyond the read
    assert int(np.bitwise_count(peq).sum()) == 128 * 150


def test_host_handle_reads_layout_banded(L, oracle):
    # Mext: bit i of plane c is set iff i >= k+1 and subject[i-(k+1)] == c, 32-bit words for every k
    rows = oracle.gen_reads(78, 64, 150)
    for k in (8, 16, 31):
        padded, peq, wn = _host_preprocess(L, B.ALGO_BANDED, rows, k=k)
        assert wn == (150 + 31) // 32 + 3
        peq = peq.reshape(-1, 5, wn, 64)
        s = 5
        for p in range(150):
            i = p + k + 1
            for cc in range(5):
                bit = (int(peq[0, cc, i // 32, s]) >> (i % 32)) & 1
                assert bit == (cc == CODE[padded[s, p]])
        assert int(np.bitwise_count(peq[0, :, :, s]).sum()) == 150


def _decode_banded_stream(raw):
    """Walk a packed banded stream the way the row loop does: returns the (row | event) sequence."""
    tokens, pos, win = [], 0, 0
    while True:
        base = 8 * win
        code = int(raw[base + pos])
        pos += 1
        if code < 25:                       # two rows
            tokens += [("row", code // 5), ("row", code % 5)]
        elif code < 30:                     # one row
            tokens.append(("row", code - 25))
        elif code == 30:
            return tokens
        elif code == 31:
            win, pos = win + 1, 0
        else:
            assert code == 63   # EVENT: the last of the 64 dispatch slots
            tokens.append(("event", int(raw[base + pos])))
            pos += 1
        assert pos <= 8


@pytest.mark.parametrize("length,k", [(150, 8), (150, 1), (150, 15), (150, 16), (150, 31), (64, 8), (65, 3), (33, 2),
                                      (1000, 8), (1000, 7), (97, 9), (257, 30)])
def test_banded_query_stream_matches_the_token_model(L, length, k):
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(B.__file__).resolve().parent / "csrc"))
    import rows_ir as R
    rng = np.random.default_rng(length + k)
    row = rng.integers(0, 5, length).astype(np.uint8)
    n = L.bgsa_hip_query_stream(B.ALGO_BANDED, row.ctypes.data, length, k, None, 0)
    buf = np.full(n, 0xEE, dtype=np.uint8)
    assert L.bgsa_hip_query_stream(B.ALGO_BANDED, row.ctypes.data, length, k, buf.ctypes.data, n) == n
    # byte for byte the Python model of the layout ...
    cut = R.banded_cut_rows(k)      # the default form for k <= 12 carries cut events (BGSA_BANDED_IMPL unset)
    assert buf.tolist() == R.banded_stream_bytes(length, k, row, cut=cut)
    # ... which decodes to exactly the row / event sequence the simulator executes
    want = [("row", int(row[v])) if kind == "row" else ("event", v) for kind, v in R.banded_tokens(length, k, cut=cut)]
    assert _decode_banded_stream(buf) == want
    assert n % 8 == 0 and (buf[-8:] == 30).all()
    assert (buf < 25).sum() >= length // 2 - 2 * (length // 16 + 3)      # most rows travel in pairs
    assert B.lib().bgsa_hip_workspace_bytes(B.ALGO_BANDED, length, length, 2) >= 2 * n


def test_banded_workspace_bound_covers_every_threshold(L):
    for length in (33, 64, 65, 100, 150, 151, 255, 1000, 1001):
        bound = B.lib().bgsa_hip_workspace_bytes(B.ALGO_BANDED, length, length, 1)
        row = np.zeros(length, dtype=np.uint8)
        for k in range(1, 32):
            if 2 * k + 1 < length:
                assert 0 < L.bgsa_hip_query_stream(B.ALGO_BANDED, row.ctypes.data, length, k, None, 0) <= bound


@pytest.mark.parametrize("qlen", [1, 31, 32, 33, 64, 100, 1000])
def test_column_block_query_stream(L, qlen):
    rng = np.random.default_rng(qlen)
    row = rng.integers(0, 5, qlen).astype(np.uint8)
    n = L.bgsa_hip_query_stream(B.ALGO_MYERS, row.ctypes.data, qlen, -1, None, 0)
    buf = np.full(n, 0xEE, dtype=np.uint8)
    assert L.bgsa_hip_query_stream(B.ALGO_MYERS, row.ctypes.data, qlen, -1, buf.ctypes.data, n) == n
    # walk it: code 7 carries no argument byte in this stream
    tokens, pos, win = [], 0, 0
    while True:
        code = buf[8 * win + pos]
        pos += 1
        if code <= 4: tokens.append(int(code))
        elif code == 5: break
        elif code == 6: win, pos = win + 1, 0
        else: tokens.append("carry")
    want = []
    for r in range(qlen):
        if r > 0 and r % 32 == 0:
            want.append("carry")
        want.append(int(row[r]))
    assert tokens == want and (buf[-8:] == 5).all()
    assert B.lib().bgsa_hip_workspace_bytes(B.ALGO_MYERS, qlen, 2000, 2) >= 2 * n


def test_score_sets_are_listed_and_selectable(L):
    sets = B.score_sets()
    assert sets[0] == (2, -3, -5)          # the reference's committed instance is always index 0
    assert len(set(sets)) == len(sets)
    for m, x, g in sets:
        assert m > x >= 2 * g and g < 0
        assert L.bgsa_hip_select_scores(m, x, g) == 0
        assert L.bgsa_hip_current_algorithm() == B.ALGO_BITPAL
        assert [ctypes.c_int.in_dll(L, v).value for v in ("match_score", "mismatch_score", "gap_score")] == [m, x, g]
    assert L.bgsa_hip_score_set(len(sets), None, None, None, None) != 0
    assert L.bgsa_hip_select_algorithm(B.ALGO_MYERS) == 0


def test_unknown_score_set_fails_loudly_and_changes_nothing(L):
    assert L.bgsa_hip_select_algorithm(B.ALGO_BITPAL) == 0
    assert L.bgsa_hip_select_scores(9, -9, -9) != 0
    assert b"BITPAL_SETS" in L.bgsa_hip_last_error()
    assert [ctypes.c_int.in_dll(L, v).value for v in ("match_score", "mismatch_score", "gap_score")] == [2, -3, -5]
    # a maintainer writing the three ints directly (they are plain globals in the reference) gets
    # the same refusal when scoring is attempted
    ctypes.c_int.in_dll(L, "match_score").value = 9
    try:
        assert L.bgsa_hip_kernel_name(B.ALGO_BITPAL, 5).startswith(b"bitpal: score set not compiled")
    finally:
        assert L.bgsa_hip_select_algorithm(B.ALGO_MYERS) == 0


def test_workspace_follows_the_selected_score_set(L):
    # carry buffers scale with the chain count of the set; plain kernels only need the query stream
    assert L.bgsa_hip_select_scores(2, -3, -5) == 0
    plain = L.bgsa_hip_workspace_bytes(B.ALGO_BITPAL, 150, 150, 100)
    blocked_default = L.bgsa_hip_workspace_bytes(B.ALGO_BITPAL, 150, 1000, 100)
    assert blocked_default > plain > 0
    if (0, -1, -1) in B.score_sets():
        assert L.bgsa_hip_select_scores(0, -1, -1) == 0
        assert L.bgsa_hip_workspace_bytes(B.ALGO_BITPAL, 150, 150, 100) == plain
        assert 0 < L.bgsa_hip_workspace_bytes(B.ALGO_BITPAL, 150, 1000, 100) < blocked_default   # 3 chains vs 13
    assert L.bgsa_hip_select_algorithm(B.ALGO_MYERS) == 0


def test_scores_with_a_common_factor_use_the_reduced_set(L):
    ints = lambda: [ctypes.c_int.in_dll(L, v).value for v in ("match_score", "mismatch_score", "gap_score")]
    assert L.bgsa_hip_select_scores(4, -6, -10) == 0          # = 2 x (2, -3, -5), the generator's commonFactor
    assert ints() == [4, -6, -10] and L.bgsa_hip_current_algorithm() == B.ALGO_BITPAL
    assert L.bgsa_hip_kernel_name(B.ALGO_BITPAL, 5).startswith(b"bitpal_asm_kernel<5>")
    assert L.bgsa_hip_select_scores(6, -9, -15) == 0
    assert L.bgsa_hip_select_scores(4, -6, -11) != 0           # no common factor, and not compiled
    assert ints() == [6, -9, -15]
    assert L.bgsa_hip_select_scores(1, 1, -1) != 0 and b"match > mismatch" in L.bgsa_hip_last_error()
    # a mismatch below two gaps is never taken: 2/-9/-4 runs as 2/-8/-4 = 2 x (1/-4/-2)
    if (1, -4, -2) in B.score_sets():
        assert L.bgsa_hip_select_scores(2, -9, -4) == 0 and ints() == [2, -9, -4]
        assert L.bgsa_hip_kernel_name(B.ALGO_BITPAL, 5).startswith(b"bitpal_asm_kernel<5>")
    assert L.bgsa_hip_select_scores(3, -50, -7) != 0           # 3/-14/-7 is not compiled
    # edit-distance scores under BitPAl run on the Myers kernels (the generator's isEdit case)
    assert L.bgsa_hip_select_scores(0, -3, -3) == 0
    assert L.bgsa_hip_kernel_name(B.ALGO_BITPAL, 5).startswith(b"myers_global_asm_kernel<5, 1>")
    assert L.bgsa_hip_select_scores(0, -7, -2) != 0            # = 0/-4/-2 = 2 x (0/-2/-1): not the edit set, not compiled
    assert L.bgsa_hip_select_algorithm(B.ALGO_MYERS) == 0


def test_myers_positive_weights_are_the_generators_m1(L):
    assert L.bgsa_hip_select_scores(0, 1, 1) == 0
    assert L.bgsa_hip_current_algorithm() == B.ALGO_MYERS
    assert [ctypes.c_int.in_dll(L, v).value for v in ("match_score", "mismatch_score", "gap_score")] == [0, 1, 1]
    assert L.bgsa_hip_select_algorithm(B.ALGO_MYERS) == 0
    assert [ctypes.c_int.in_dll(L, v).value for v in ("match_score", "mismatch_score", "gap_score")] == [0, -1, -1]


@pytest.mark.parametrize("qlen", [1, 2, 13, 14, 15, 27, 28, 29, 64, 151, 1000])
def test_two_rows_per_token_query_stream(L, qlen):
    # the stream of the <= 64 bp Myers kernels (k = -2): tokens of two rows, an odd last row alone
    rng = np.random.default_rng(qlen)
    row = rng.integers(0, 5, qlen).astype(np.uint8)
    n = L.bgsa_hip_query_stream(B.ALGO_MYERS, row.ctypes.data, qlen, -2, None, 0)
    buf = np.full(n, 0xEE, dtype=np.uint8)
    assert L.bgsa_hip_query_stream(B.ALGO_MYERS, row.ctypes.data, qlen, -2, buf.ctypes.data, n) == n
    assert [v for kind, v in _decode_banded_stream(buf) if kind == "row"] == row.tolist()
    assert "event" not in {kind for kind, _ in _decode_banded_stream(buf)}
    n_tokens = (qlen + 1) // 2
    assert n == 8 * (n_tokens // 7 + 2) and (buf[-8:] == 30).all()
    assert int((buf < 25).sum()) == qlen // 2 and int(((buf >= 25) & (buf < 30)).sum()) == qlen % 2
    assert n <= L.bgsa_hip_workspace_bytes(B.ALGO_MYERS, qlen, 64, 1)     # sized for the longer one-row stream
