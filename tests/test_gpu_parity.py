"""GPU parity: the HIP path (through the C ABI) against the oracle and the golden fixtures.

Bit-exact: every score is an integer.  Run with `pytest -m gpu` on an MI355X.
"""
import ctypes

import numpy as np
import pytest

import bgsa_amd as B
from conftest import golden_names, load_golden

pytestmark = pytest.mark.gpu

ALGO_OF = {"original_cpu": B.ALGO_MYERS, "original_avx2": B.ALGO_BITPAL, "banded_cpu": B.ALGO_BANDED}


@pytest.mark.parametrize("name", golden_names())
def test_golden_fixture(name):
    g = load_golden(name)
    algo = ALGO_OF[g["variant"]]
    got = B.align_all_pairs(g["queries"], g["subjects"], algo=algo, k=max(g["k"], 0))
    assert got.dtype == g["scores"].dtype
    assert np.array_equal(got, g["scores"])


@pytest.mark.parametrize("slen", [1, 7, 32, 33, 64, 100, 150, 151, 192, 250, 256, 300, 384, 448, 500, 640, 700, 800, 1000, 1024])
def test_myers_lengths_vs_oracle(oracle, slen):
    qlen = max(1, slen - 3)
    q = oracle.gen_reads(1000 + slen, 5, qlen)
    s = oracle.gen_reads(2000 + slen, 70, slen)
    m = min(qlen, slen)
    s[:20, :m] = oracle.mutate(q[np.arange(20) % 5][:, :m], np.arange(20) % 7, slen)
    got = B.align_all_pairs(q, s, algo=B.ALGO_MYERS)
    assert np.array_equal(got, oracle.myers64(q, s))


@pytest.mark.parametrize("qlen", [1, 6, 7, 8, 13, 14, 15, 21, 49, 50])
def test_query_stream_window_boundaries(oracle, qlen):
    # the packed query stream holds 7 characters per 8-byte window: lengths around multiples of 7
    q = oracle.gen_reads(700 + qlen, 9, qlen)
    s = oracle.gen_reads(800 + qlen, 64, 40)
    assert np.array_equal(B.align_all_pairs(q, s, algo=B.ALGO_MYERS), oracle.myers64(q, s))
    assert np.array_equal(B.align_all_pairs(q, s, algo=B.ALGO_BITPAL), oracle.bitpal(q, s))


@pytest.mark.parametrize("qlen,slen", [(120, 897), (333, 929), (1000, 960), (64, 961), (990, 992), (997, 1000), (1100, 1023), (1021, 1024)])
def test_myers_897_to_1024_bp_on_resident_peq_planes(oracle, qlen, slen):
    """30 and 32 words (round 5): five Peq planes resident, the two carry chains in turns over blocks of nine words
    (rows_ir.myers_body(split = 9)).  Reads with N columns, homopolymers (carries through every word and across every block
    boundary of a chain's turn) and near-copies, against the oracle."""
    L = B.lib()
    L.bgsa_hip_select_alignment(0)
    assert L.bgsa_hip_kernel_name(B.ALGO_MYERS, (slen + 31) // 32).startswith(b"myers_global_asm_kernel<%d, 1>" % (30 if slen <= 960 else 32))
    q = oracle.gen_reads(8100 + qlen, 6, qlen)
    s = oracle.gen_reads(8200 + slen, 140, slen)
    m = min(qlen, slen)
    s[:24, :m] = oracle.mutate(q[np.arange(24) % 6][:, :m], np.arange(24) % 9, 8300)
    s[30] = ord("A")
    s[31] = ord("N")
    s[32, ::2] = ord("C")
    s[33, 280:300] = ord("N")           # across the boundary of the first and second turn (words 8 | 9)
    q[4] = ord("A")
    q[5, : qlen // 2] = ord("N")
    assert np.array_equal(B.align_all_pairs(q, s, algo=B.ALGO_MYERS), oracle.myers64(q, s))
    assert L.bgsa_hip_stream_faults(1) == 0


def test_myers_query_tiles_and_offsets(oracle):
    # ref_start/ref_end windows and many queries (several q-tiles), odd row alignment (stride 151)
    q = oracle.gen_reads(31, 203, 150)
    s = oracle.gen_reads(32, 256, 150)
    want = oracle.myers64(q, s)
    a = B.DeviceAligner(B.ALGO_MYERS)
    a.set_queries(q)
    a.set_subjects(s)
    assert np.array_equal(a.score().cpu().numpy(), want)
    assert np.array_equal(a.score(100, 203).cpu().numpy(), want[100:203])
    assert np.array_equal(a.score(7, 8).cpu().numpy(), want[7:8])


@pytest.mark.parametrize("slen", [1, 31, 32, 33, 64, 65, 100, 150, 200, 256])
def test_bitpal_lengths_vs_oracle(oracle, slen):
    qlen = max(1, slen - 2)
    q = oracle.gen_reads(3000 + slen, 4, qlen)
    s = oracle.gen_reads(4000 + slen, 70, slen)
    m = min(qlen, slen)
    s[:20, :m] = oracle.mutate(q[np.arange(20) % 4][:, :m], np.arange(20) % 7, slen)
    assert np.array_equal(B.align_all_pairs(q, s, algo=B.ALGO_BITPAL), oracle.bitpal(q, s))


@pytest.mark.parametrize("length,k", [(150, 8), (150, 4), (150, 15), (150, 16), (150, 31), (64, 8), (65, 8), (73, 8),
                                      (100, 12), (128, 8), (136, 8), (200, 8), (250, 20), (500, 8), (1000, 8), (2000, 8), (1500, 25),
                                      (150, 13), (150, 14), (97, 13), (700, 15), (250, 24), (320, 31)])
def test_banded_vs_oracle(oracle, length, k):
    # includes lengths (65, 136, 200) where the reference itself writes out of bounds: the oracle
    # (and the kernel) follow the in-bounds semantics there (DESIGN.md "banded domain")
    q = oracle.gen_reads(5000 + length + k, 6, length)
    s = oracle.gen_reads(6000 + length + k, 130, length)
    s[:60] = oracle.mutate(q[np.arange(60) % 6], np.arange(60) % (2 * k + 6), length + k)
    got = B.align_all_pairs(q, s, algo=B.ALGO_BANDED, k=k)
    want = oracle.banded64(q, s, k)
    assert got.dtype == np.int8 and np.array_equal(got, want)
    assert (want != 127).any() and (want == 127).any()


@pytest.mark.parametrize("k,length", [(8, 150), (20, 150), (8, 500), (12, 97), (14, 150), (13, 320), (31, 250)])
def test_banded_sparse_survivors_take_the_queue(oracle, k, length):
    """One or two near-duplicates per wave of 64 subjects among random reads: the first pass stops those waves
    at a late test and the survivor queue's second pass scores the pairs (banded.hip).  Dense survivors
    (a whole wave of near-duplicates) stay in the first pass.  Both against the oracle."""
    rng = np.random.default_rng(k * 1000 + length)
    q = oracle.gen_reads(7000 + k, 40, length)
    s = oracle.gen_reads(7100 + length, 64 * 12, length)
    for g in range(12):                                   # groups 0..9: 1-3 survivors each, at random lanes
        if g < 10:
            lanes = rng.choice(64, size=1 + g % 3, replace=False)
        else:                                             # groups 10, 11: every lane a near-duplicate of some query
            lanes = np.arange(64)
        src = rng.integers(0, 40, len(lanes))
        s[g * 64 + lanes] = oracle.mutate(q[src], rng.integers(0, k + 3, len(lanes)), 7200 + g)
    got = B.align_all_pairs(q, s, algo=B.ALGO_BANDED, k=k)
    want = oracle.banded64(q, s, k)
    assert np.array_equal(got, want)
    assert (want != 127).sum() >= 20


@pytest.mark.parametrize("k,length", [(8, 150), (12, 100), (15, 150), (31, 150)])
def test_banded_foreign_query_bytes_score_as_class_0_in_every_pass(oracle, k, length):
    """The device layer accepts any bytes in the mapped query buffer; a byte that is no class (> 4) scores as class 0 — in the
    packed streams (bgsa_common.h: the packers clamp) and in the dense pass over regrouped survivors, which reads the query
    rows itself (banded_finish_pair / banded_finish_pair_cut: foreign bytes are cleared four at a time when a chunk is
    loaded).  Lone survivors in otherwise random groups force the dense pass; the same rows with zeros in those places must
    score identically, and equal to the oracle on queries whose characters there are 'A'."""
    import torch
    nq = 9
    q = oracle.gen_reads(7300 + k, nq, length)
    s = oracle.gen_reads(7400 + k, 64 * 4, length)
    for g, lane, qi in ((0, 5, 2), (1, 63, 7), (3, 0, 4), (2, 31, 2)):        # one or two near-duplicates per group of 64
        s[64 * g + lane] = oracle.mutate(q[qi:qi + 1], [min(3, k - 1)], 7500 + g)[0]
    a = B.DeviceAligner(B.ALGO_BANDED, k=k)
    a.set_queries(q)
    a.set_subjects(s)
    rng = np.random.default_rng(k)
    pos = rng.integers(0, length, (nq, 6))
    content = a.d_content.cpu().numpy().copy()
    zeroed = content.copy()
    foreign = content.copy()
    for i in range(nq):
        for j, p in enumerate(pos[i]):
            zeroed[i * (length + 1) + p] = 0
            foreign[i * (length + 1) + p] = (5, 7, 9, 65, 200, 255)[j]
    out = {}
    for name, buf in (("zeroed", zeroed), ("foreign", foreign)):
        a.d_content.copy_(torch.from_numpy(buf).to(a.d_content.device))
        out[name] = a.score().cpu().numpy()[:, : s.shape[0]]
        a.check_faults()
    assert np.array_equal(out["foreign"], out["zeroed"])
    qa = q.copy()
    for i in range(nq):
        qa[i, pos[i]] = ord("A")
    assert np.array_equal(out["zeroed"], oracle.banded64(qa, s, k))
    assert (out["zeroed"] != 127).sum() >= 3          # the planted pairs survive: the dense pass ran


def test_banded_refuses_unequal_lengths(oracle):
    with pytest.raises(B.BgsaHipError):
        B.align_all_pairs(oracle.gen_reads(1, 2, 140), oracle.gen_reads(2, 64, 150), algo=B.ALGO_BANDED, k=8)


@pytest.mark.parametrize("algo,qlen,slen", [(B.ALGO_MYERS, 1025, 1025), (B.ALGO_MYERS, 300, 2500), (B.ALGO_MYERS, 4000, 4000),
                                            (B.ALGO_MYERS, 31, 1100), (B.ALGO_MYERS, 64, 2049), (B.ALGO_MYERS, 97, 3300),
                                            (B.ALGO_BITPAL, 257, 257), (B.ALGO_BITPAL, 500, 1000), (B.ALGO_BITPAL, 90, 300),
                                            (B.ALGO_BITPAL, 31, 320), (B.ALGO_BITPAL, 64, 513), (B.ALGO_BITPAL, 33, 2000)])
def test_beyond_register_limits(oracle, algo, qlen, slen):
    # Myers > 1024 bp / BitPAl > 256 bp: column blocks with carry words between blocks
    # (myers_blocked_kernel, bitpal_blocked_kernel)
    q = oracle.gen_reads(8000 + qlen, 3, qlen)
    s = oracle.gen_reads(9000 + slen, 130, slen)
    m = min(qlen, slen)
    s[:12, :m] = oracle.mutate(q[np.arange(12) % 3][:, :m], np.arange(12) * 7, slen)
    s[3, 5:40] = ord("N")
    got = B.align_all_pairs(q, s, algo=algo)
    want = oracle.myers64(q, s) if algo == B.ALGO_MYERS else oracle.bitpal(q, s)
    assert np.array_equal(got, want)


def test_empty_inputs_are_no_ops(oracle):
    # zero queries in the window / zero subjects: nothing to launch, nothing touched
    L = B.lib()
    q = oracle.gen_reads(5, 4, 50)
    a = B.DeviceAligner(B.ALGO_MYERS)
    a.set_queries(q)
    a.set_subjects(oracle.gen_reads(6, 64, 50))
    out = a.score(2, 2)
    assert tuple(out.shape) == (0, 64)
    import torch
    sentinel = torch.full((4, 64), 7, dtype=torch.int16, device="cuda:0")
    rc = L.bgsa_hip_cal_align_score_dev(B.ALGO_MYERS, a.d_content.data_ptr(), a.d_peq.data_ptr(), sentinel.data_ptr(),
                                        50, 50, 0, 0, 4, 2, 0, None, 0, None)
    assert rc == 0 and bool((sentinel == 7).all())


def test_launch_is_graph_capturable(oracle):
    """With a caller-owned workspace the hot path allocates nothing and never synchronises, so the
    pack + score launches can be captured into a hipGraph and replayed."""
    import torch
    q = oracle.gen_reads(71, 40, 150)
    s = oracle.gen_reads(72, 256, 150)
    want = oracle.myers64(q, s)
    a = B.DeviceAligner(B.ALGO_MYERS)
    a.set_queries(q)
    a.set_subjects(s)
    out = torch.zeros((40, 256), dtype=torch.int16, device="cuda:0")
    a.score(out=out)  # warm-up allocates the workspace outside the capture
    torch.cuda.synchronize()
    out.zero_()
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            a.score(out=out)
    torch.cuda.synchronize()
    out.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), want)


def test_large_batch_properties(oracle):
    """A batch too large for the CPU oracle (2k x 256k = 5e8 pairs), checked through properties
    that do not depend on size: planted identical pairs score 0, duplicated subjects agree
    wherever they sit (other group, other lane, other block), shuffling the subjects permutes the
    columns, the distance is symmetric in its arguments, and a random sample equals the oracle."""
    import torch
    rng = np.random.default_rng(7)
    nq, ns = 2000, 256 * 1024
    q = oracle.gen_reads(81, nq, 150)
    s = oracle.gen_reads(82, ns, 150)
    planted = rng.choice(ns, size=nq, replace=False)
    s[planted] = q                                    # subject planted[i] == query i
    dup_src = rng.choice(ns, size=4096, replace=False)
    dup_dst = (dup_src + ns // 2 + 37) % ns
    keep = ~np.isin(dup_dst, planted) & ~np.isin(dup_dst, dup_src)
    dup_src, dup_dst = dup_src[keep], dup_dst[keep]
    s[dup_dst] = s[dup_src]
    a = B.DeviceAligner(B.ALGO_MYERS)
    a.set_queries(q)
    a.set_subjects(s)
    scores = a.score()
    idx = torch.arange(nq, device="cuda:0")
    assert bool((scores[idx, torch.from_numpy(planted).cuda()] == 0).all())
    assert bool((scores[:, torch.from_numpy(dup_src).cuda()] == scores[:, torch.from_numpy(dup_dst).cuda()]).all())
    assert int(scores.max()) <= 0 and int(scores.min()) >= -150
    perm = rng.permutation(ns)
    b = B.DeviceAligner(B.ALGO_MYERS)
    b.set_queries(q)
    b.set_subjects(s[perm])
    assert bool((b.score() == scores[:, torch.from_numpy(perm).cuda()]).all())
    # symmetry: d(q_i, s_j) == d(s_j, q_i) on a 64 x 64 corner with roles swapped
    c = B.DeviceAligner(B.ALGO_MYERS)
    c.set_queries(s[:64])
    c.set_subjects(q[:64])
    assert bool((c.score() == scores[:64, :64].T).all())
    qi = rng.integers(0, nq, 3000)
    sj = rng.integers(0, ns, 3000)
    got = scores[torch.from_numpy(qi).cuda(), torch.from_numpy(sj).cuda()].cpu().numpy()
    want = np.array([oracle.myers64(q[i:i + 1], s[j:j + 1])[0, 0] for i, j in zip(qi[:400], sj[:400])])
    assert np.array_equal(got[:400], want)


@pytest.mark.parametrize("length", [31, 32, 33, 64, 150, 160, 256, 257, 512, 1000, 1024])
def test_long_carry_chains(oracle, length):
    """Inputs that drive the inter-word carry chains end to end: identical homopolymers (the
    addition carries across every word), a single mismatch at each end, period-2 repeats against
    their shift, all-N reads."""
    def rows(*seqs):
        return np.stack([np.frombuffer(x, dtype=np.uint8) for x in seqs])
    a, c = b"A" * length, b"C" * length
    ac = (b"AC" * length)[:length]
    ca = (b"CA" * length)[:length]
    q = rows(a, c, ac, b"T" + a[1:], a[:-1] + b"G", b"N" * length)
    s = np.concatenate([q, rows(ca, c[:-1] + b"A", b"G" + c[1:])] * 8)
    for algo, fn in ((B.ALGO_MYERS, oracle.myers64), (B.ALGO_BITPAL, oracle.bitpal)):
        if algo == B.ALGO_BITPAL and length > 512:
            continue  # state-in-memory kernel: covered elsewhere, slow on purpose
        assert np.array_equal(B.align_all_pairs(q, s, algo=algo), fn(q, s))
    if length > 40:
        for k in (8, 20):
            assert np.array_equal(B.align_all_pairs(q, s, algo=B.ALGO_BANDED, k=k), oracle.banded64(q, s, k))


@pytest.mark.parametrize("algo,k", [(B.ALGO_BITPAL, 0), (B.ALGO_BANDED, 8), (B.ALGO_BANDED, 31)])
def test_mid_size_properties_bitpal_and_banded(oracle, algo, k):
    """400 x 64k pairs: planted identical pairs take the extreme score, duplicated subjects agree in
    any group / lane, shuffling subjects permutes columns, and a random sample equals the oracle."""
    import torch
    rng = np.random.default_rng(11 + algo + k)
    nq, ns = 400, 64 * 1024
    q = oracle.gen_reads(181, nq, 150)
    s = oracle.gen_reads(182, ns, 150)
    planted = rng.choice(ns, size=nq, replace=False)
    s[planted] = q
    near = rng.choice(np.setdiff1d(np.arange(ns), planted), size=2000, replace=False)
    s[near] = oracle.mutate(q[rng.integers(0, nq, 2000)], rng.integers(0, 12, 2000), 5)   # pairs that survive the band
    a = B.DeviceAligner(algo, k=k)
    a.set_queries(q)
    a.set_subjects(s)
    scores = a.score()
    diag = scores[torch.arange(nq, device="cuda:0"), torch.from_numpy(planted).cuda()]
    assert bool((diag == (300 if algo == B.ALGO_BITPAL else 0)).all())
    perm = rng.permutation(ns)
    b = B.DeviceAligner(algo, k=k)
    b.set_queries(q)
    b.set_subjects(s[perm])
    assert bool((b.score() == scores[:, torch.from_numpy(perm).cuda()]).all())
    fn = (lambda x, y: oracle.bitpal(x, y)) if algo == B.ALGO_BITPAL else (lambda x, y: oracle.banded64(x, y, k))
    cols = np.concatenate([near[:96], rng.integers(0, ns, 96)])
    want = fn(q[:24], s[cols])
    got = scores[:24][:, torch.from_numpy(cols).cuda()].cpu().numpy()
    assert np.array_equal(got, want)
    if algo == B.ALGO_BANDED:
        assert (want != 127).any() and (want == 127).any()


def test_fuzz_all_algorithms(oracle):
    """Random lengths / counts / thresholds / alphabets through every kernel family."""
    rng = np.random.default_rng(2026)
    alphabet = np.frombuffer(b"ACGTACGTACGTNacgtX-", dtype=np.uint8)
    for trial in range(60):
        algo = (B.ALGO_MYERS, B.ALGO_BITPAL, B.ALGO_BANDED)[trial % 3]
        nq, ns = int(rng.integers(1, 6)), int(rng.integers(1, 150))
        if algo == B.ALGO_BANDED:
            k = int(rng.integers(1, 32))
            qlen = slen = int(rng.integers(2 * k + 2, 400))
        else:
            k = 0
            qlen, slen = int(rng.integers(1, 330)), int(rng.integers(1, 1200 if algo == B.ALGO_MYERS else 330))
        q = alphabet[rng.integers(0, len(alphabet), (nq, qlen))]
        s = alphabet[rng.integers(0, len(alphabet), (ns, slen))]
        m = min(qlen, slen)
        near = min(ns, 8)
        s[:near, :m] = q[rng.integers(0, nq, near)][:, :m]
        flip = rng.random((near, m)) < 0.04
        s[:near, :m][flip] = alphabet[rng.integers(0, 4, int(flip.sum()))]
        got = B.align_all_pairs(q, s, algo=algo, k=k)
        want = {B.ALGO_MYERS: lambda: oracle.myers64(q, s), B.ALGO_BITPAL: lambda: oracle.bitpal(q, s),
                B.ALGO_BANDED: lambda: oracle.banded64(q, s, k)}[algo]()
        assert np.array_equal(got, want), (trial, algo, nq, ns, qlen, slen, k)


def test_bad_arguments_fail_loudly():
    L = B.lib()
    assert L.bgsa_hip_cal_align_score_dev(B.ALGO_MYERS, None, None, None, 150, 150, 64, 0, 1, 5, 0, None, 0, None) == -1
    assert b"bad argument" in L.bgsa_hip_last_error()


def test_host_surface_matches_device_surface(oracle):
    """hip_handle_reads + hip_cal_align_score + align_hip on HOST buffers (the reference's seams)."""
    L = B.lib()
    q = oracle.gen_reads(41, 9, 150)
    s, _ = B.pad_rows(oracle.gen_reads(42, 130, 150))
    want = oracle.myers64(q, s)
    n, length = s.shape
    L.bgsa_hip_select_algorithm(B.ALGO_MYERS)
    L.init_mapping_table()
    table = np.ctypeslib.as_array((ctypes.c_uint32 * 128).in_dll(L, "mapping_table"))
    sbuf = B.rows_to_buffer(s)
    seq = B.SeqT(len=length, size=sbuf.size, count=n, extra_size=0, extra_count=0, content=sbuf.ctypes.data)
    wn = B.word_num(B.ALGO_MYERS, 150, length)
    peq = np.zeros(B.group_words(B.ALGO_MYERS, wn) * (n // 64), dtype=np.uint32)
    L.hip_handle_reads(ctypes.byref(seq), peq.ctypes.data, wn, 0, n)
    # queries mapped through mapping_table the way get_ref_from_file does (file.c:134-139)
    qbuf = B.rows_to_buffer(q)
    keep = qbuf == ord("\n")
    qmapped = table[qbuf].astype(np.uint8)
    qmapped[keep] = ord("\n")
    out = np.zeros((9, n), dtype=np.int16)
    L.hip_cal_align_score(qmapped.ctypes.data, peq.ctypes.data, out.ctypes.data, 150, 9, length, n, 0, 9, wn, 27, None)
    assert np.array_equal(out, want)
    # a ref window
    out2 = np.zeros((4, n), dtype=np.int16)
    L.hip_cal_align_score(qmapped.ctypes.data, peq.ctypes.data, out2.ctypes.data, 150, 9, length, n, 3, 7, wn, 27, None)
    assert np.array_equal(out2, want[3:7])
    # fine-grained README-style call: one query against the second group only
    res = np.zeros(n, dtype=np.int16)
    row = np.ascontiguousarray(qmapped[2 * 151: 3 * 151])
    group1 = peq[B.group_words(B.ALGO_MYERS, wn):]
    L.align_hip(row.ctypes.data, group1.ctypes.data, 150, length, wn, 1, 1, res.ctypes.data, None)
    assert np.array_equal(res[64:128], want[2, 64:128]) and not res[:64].any()


def _host_seam_inputs(L, q, s):
    table = np.ctypeslib.as_array((ctypes.c_uint32 * 128).in_dll(L, "mapping_table"))
    sbuf = B.rows_to_buffer(s)
    seq = B.SeqT(len=s.shape[1], size=sbuf.size, count=s.shape[0], extra_size=0, extra_count=0, content=sbuf.ctypes.data)
    qbuf = B.rows_to_buffer(q)
    keep = qbuf == ord("\n")
    qmapped = table[qbuf].astype(np.uint8)
    qmapped[keep] = ord("\n")
    return sbuf, seq, qmapped


@pytest.mark.parametrize("fixture,k,ref_word_num", [("f8_banded_k8_150", 8, False), ("f8_banded_k8_150", 8, True),
                                                    ("f8_banded_k16_150", 16, True), ("f8_banded_k4_150", 4, False),
                                                    (None, 31, False), (None, 31, True)])
def test_host_surface_banded(oracle, fixture, k, ref_word_num):
    """hip_handle_reads + hip_cal_align_score + align_hip for the banded filter: int8 results, the global
    `threshold`, and both word_num conventions the seams accept — the library's own (32-bit words) and the
    reference's banded formula over 64-bit words (banded/BGSA_CPU/cal_cpu.c:253-254)."""
    L = B.lib()
    if fixture:
        g = load_golden(fixture)
        q, s0, want0 = g["queries"], g["subjects"], g["scores"]
    else:
        q = oracle.gen_reads(51, 5, 150)
        s0 = oracle.gen_reads(52, 130, 150)
        s0[:50] = oracle.mutate(q[np.arange(50) % 5], np.arange(50) % 45, 53)
        want0 = oracle.banded64(q, s0, k)
    s, _ = B.pad_rows(s0)
    n, length = s.shape
    nq = q.shape[0]
    want = np.full((nq, n), 0, dtype=np.int8)
    want[:, : s0.shape[0]] = want0
    want[:, s0.shape[0]:] = oracle.banded64(q, s[s0.shape[0]:], k) if n > s0.shape[0] else 0
    L.bgsa_hip_select_algorithm(B.ALGO_BANDED)
    threshold = ctypes.c_int.in_dll(L, "threshold")
    old_threshold = threshold.value
    threshold.value = k
    try:
        L.init_mapping_table()
        sbuf, seq, qmapped = _host_seam_inputs(L, q, s)
        if ref_word_num:
            wn = (length - k + 63) // 64 + 1                       # words of the reference's uint64_t cpu_read_t
            peq = np.zeros(5 * wn * 64 * (n // 64), dtype=np.uint64)
            group_elems = 5 * wn * 64
        else:
            wn = B.word_num(B.ALGO_BANDED, length, length, k)
            peq = np.zeros(B.group_words(B.ALGO_BANDED, wn, k) * (n // 64), dtype=np.uint32)
            group_elems = B.group_words(B.ALGO_BANDED, wn, k)
        L.hip_handle_reads(ctypes.byref(seq), peq.ctypes.data, wn, 0, n)
        out = np.zeros((nq, n), dtype=np.int8)
        L.hip_cal_align_score(qmapped.ctypes.data, peq.ctypes.data, out.ctypes.data, length, nq, length, n, 0, nq, wn, 27, None)
        assert np.array_equal(out, want)
        out2 = np.zeros((2, n), dtype=np.int8)                      # a query window
        L.hip_cal_align_score(qmapped.ctypes.data, peq.ctypes.data, out2.ctypes.data, length, nq, length, n, 1, 3, wn, 27, None)
        assert np.array_equal(out2, want[1:3])
        # fine-grained call: one query against the last group only, int8 results at result_index
        res = np.zeros(n, dtype=np.int8)
        row = np.ascontiguousarray(qmapped[2 * (length + 1): 3 * (length + 1)])
        last = n // 64 - 1
        grp = peq[last * group_elems:]
        L.align_hip(row.ctypes.data, grp.ctypes.data, length, length, wn, 1, last, res.ctypes.data, None)
        assert np.array_equal(res[last * 64:], want[2, last * 64:]) and not res[: last * 64].any()
    finally:
        threshold.value = old_threshold
        L.bgsa_hip_select_algorithm(B.ALGO_MYERS)


@pytest.mark.parametrize("fixture", ["f7_bitpal_150", "f9_bitpal_140x150", "f5_bitpal_specials"])
def test_host_surface_bitpal(oracle, fixture):
    """The same three seams with BGSA_ALGO_BITPAL selected (int16 results, the five score ints)."""
    L = B.lib()
    g = load_golden(fixture)
    q, s0 = g["queries"], g["subjects"]
    s, _ = B.pad_rows(s0)
    n, length = s.shape
    nq, qlen = q.shape
    want = oracle.bitpal(q, s)
    assert np.array_equal(want[:, : s0.shape[0]], g["scores"])
    L.bgsa_hip_select_algorithm(B.ALGO_BITPAL)
    try:
        L.init_mapping_table()
        sbuf, seq, qmapped = _host_seam_inputs(L, q, s)
        wn = B.word_num(B.ALGO_BITPAL, qlen, length)
        peq = np.zeros(B.group_words(B.ALGO_BITPAL, wn) * (n // 64), dtype=np.uint32)
        L.hip_handle_reads(ctypes.byref(seq), peq.ctypes.data, wn, 0, n)
        out = np.zeros((nq, n), dtype=np.int16)
        L.hip_cal_align_score(qmapped.ctypes.data, peq.ctypes.data, out.ctypes.data, qlen, nq, length, n, 0, nq, wn, 27, None)
        assert np.array_equal(out, want)
        res = np.zeros(n, dtype=np.int16)
        row = np.ascontiguousarray(qmapped[0: qlen + 1])
        L.align_hip(row.ctypes.data, peq.ctypes.data, qlen, length, wn, n // 64, 0, res.ctypes.data, None)
        assert np.array_equal(res, want[0])
    finally:
        L.bgsa_hip_select_algorithm(B.ALGO_MYERS)


@pytest.mark.parametrize("algo,nq,k", [(B.ALGO_MYERS, 70, 0), (B.ALGO_BANDED, 133, 8), (B.ALGO_BITPAL, 67, 0)])
@pytest.mark.parametrize("pinned", [True, False])
def test_host_seam_copy_out_in_tiles(oracle, algo, nq, k, pinned):
    """A block large enough (>= 8 MiB of scores) for hip_cal_align_score to score it in query tiles and copy tile t
    out while tile t+1 runs: the caller's buffer must hold every score on return — whole matrix against the
    device-resident path (same kernels), a slice of it against the oracle; with page-locked and pageable buffers,
    uneven tiles (nq not a multiple of 8) and a ref_start / ref_end window."""
    import torch
    L = B.lib()
    n, length = 64 * 1024, 150
    q = oracle.gen_reads(1300 + algo, nq, length)
    s = oracle.gen_reads(1400 + algo, n, length)
    s[:256] = oracle.mutate(q[np.arange(256) % nq], np.arange(256) % 12, 1500 + algo)
    esz = 1 if algo == B.ALGO_BANDED else 2
    dtype = np.int8 if esz == 1 else np.int16
    a = B.DeviceAligner(algo, k=k)
    a.set_queries(q)
    a.set_subjects(s)
    want = a.score().cpu().numpy()[:, :n]
    fn = {B.ALGO_MYERS: oracle.myers64, B.ALGO_BITPAL: oracle.bitpal, B.ALGO_BANDED: lambda x, y: oracle.banded64(x, y, k)}[algo]
    assert np.array_equal(want[:, :320], fn(q, s[:320]))
    threshold = ctypes.c_int.in_dll(L, "threshold")
    old_threshold = threshold.value
    L.bgsa_hip_select_algorithm(algo)
    bufs = []
    try:
        threshold.value = k if algo == B.ALGO_BANDED else old_threshold
        L.init_mapping_table()
        sbuf, seq, qmapped = _host_seam_inputs(L, q, s)
        wn = B.word_num(algo, length, length, k)
        n_peq = B.group_words(algo, wn, k) * (n // 64)
        if pinned:
            p1, p2 = L.malloc_mem(n_peq * 4), L.malloc_mem(nq * n * esz)
            bufs = [p1, p2]
            peq = np.ctypeslib.as_array(ctypes.cast(p1, ctypes.POINTER(ctypes.c_uint32)), shape=(n_peq,))
            out = np.ctypeslib.as_array(ctypes.cast(p2, ctypes.POINTER(ctypes.c_int8 if esz == 1 else ctypes.c_int16)), shape=(nq, n))
        else:
            peq, out = np.zeros(n_peq, dtype=np.uint32), np.zeros((nq, n), dtype=dtype)
        peq[:] = 0
        L.hip_handle_reads(ctypes.byref(seq), peq.ctypes.data, wn, 0, n)
        out[:] = 99
        L.hip_cal_align_score(qmapped.ctypes.data, peq.ctypes.data, out.ctypes.data, length, nq, length, n, 0, nq, wn, 27, None)
        assert np.array_equal(out, want)
        # a window of the queries: rows relative to ref_start, the rest of the buffer untouched
        out[:] = 99
        lo, hi = 3, nq - 2
        L.hip_cal_align_score(qmapped.ctypes.data, peq.ctypes.data, out.ctypes.data, length, nq, length, n, lo, hi, wn, 27, None)
        assert np.array_equal(out[: hi - lo], want[lo:hi]) and bool((out[hi - lo:] == 99).all())
        torch.cuda.synchronize()
        assert L.bgsa_hip_stream_faults(1) == 0
    finally:
        for p in bufs:
            L.free_mem(p)
        threshold.value = old_threshold
        L.bgsa_hip_select_algorithm(B.ALGO_MYERS)


def test_host_seam_keeps_the_bucket_resident(oracle):
    """100-query blocks against one bucket: the Peq words cross PCIe once, not once per call; rewriting
    the buffer with hip_handle_reads invalidates the device copy; auto-residency can be switched off."""
    L = B.lib()
    q = oracle.gen_reads(61, 12, 150)
    s, _ = B.pad_rows(oracle.gen_reads(62, 256, 150))
    s2, _ = B.pad_rows(oracle.gen_reads(63, 256, 150))
    L.bgsa_hip_select_algorithm(B.ALGO_MYERS)
    L.init_mapping_table()
    sbuf, seq, qmapped = _host_seam_inputs(L, q, s)
    wn = 5
    peq = np.zeros(B.group_words(B.ALGO_MYERS, wn) * 4, dtype=np.uint32)

    def uploads():
        u = ctypes.c_uint64()
        L.bgsa_hip_seam_stats(None, ctypes.byref(u), None)
        return u.value

    L.hip_handle_reads(ctypes.byref(seq), peq.ctypes.data, wn, 0, 256)
    before = uploads()
    out = np.zeros((4, 256), dtype=np.int16)
    for lo in (0, 4, 8):
        L.hip_cal_align_score(qmapped.ctypes.data, peq.ctypes.data, out.ctypes.data, 150, 12, 150, 256, lo, lo + 4, wn, 27, None)
        assert np.array_equal(out, oracle.myers64(q[lo:lo + 4], s))
    assert uploads() == before + 1                       # three calls, one upload
    # a sub-range of the resident bucket (what align_hip passes) uses the same copy
    res = np.zeros(256, dtype=np.int16)
    L.align_hip(qmapped[:151].copy().ctypes.data, peq[B.group_words(B.ALGO_MYERS, wn) * 2:].ctypes.data, 150, 150, wn, 2, 2,
                res.ctypes.data, None)
    assert np.array_equal(res[128:], oracle.myers64(q[:1], s)[0, 128:]) and uploads() == before + 1
    # new content in the same host buffer: hip_handle_reads drops the stale copy
    sbuf2, seq2, _ = _host_seam_inputs(L, q, s2)
    peq[:] = 0
    L.hip_handle_reads(ctypes.byref(seq2), peq.ctypes.data, wn, 0, 256)
    L.hip_cal_align_score(qmapped.ctypes.data, peq.ctypes.data, out.ctypes.data, 150, 12, 150, 256, 0, 4, wn, 27, None)
    assert np.array_equal(out, oracle.myers64(q[:4], s2)) and uploads() == before + 2
    # a bucket buffer handed back with free_mem() stops being resident: a new buffer at the same address, filled
    # by other means, is uploaded afresh
    nbytes = peq.nbytes
    pa = L.malloc_mem(nbytes)
    a = np.ctypeslib.as_array(ctypes.cast(pa, ctypes.POINTER(ctypes.c_uint32)), shape=(peq.size,))
    a[:] = 0
    L.hip_handle_reads(ctypes.byref(seq), pa, wn, 0, 256)          # subjects s
    L.hip_cal_align_score(qmapped.ctypes.data, pa, out.ctypes.data, 150, 12, 150, 256, 0, 4, wn, 27, None)
    assert np.array_equal(out, oracle.myers64(q[:4], s))
    n1 = uploads()
    L.free_mem(pa)
    pb = L.malloc_mem(nbytes)
    b = np.ctypeslib.as_array(ctypes.cast(pb, ctypes.POINTER(ctypes.c_uint32)), shape=(peq.size,))
    b[:] = peq                                                       # subjects s2, written without hip_handle_reads
    L.hip_cal_align_score(qmapped.ctypes.data, pb, out.ctypes.data, 150, 12, 150, 256, 0, 4, wn, 27, None)
    assert np.array_equal(out, oracle.myers64(q[:4], s2)) and uploads() == n1 + 1
    L.free_mem(pb)
    # stateless mode: every call uploads
    assert L.bgsa_hip_set_auto_resident(0) == 0
    try:
        peq[:] = 0
        L.hip_handle_reads(ctypes.byref(seq), peq.ctypes.data, wn, 0, 256)
        n0 = uploads()
        for _ in range(2):
            L.hip_cal_align_score(qmapped.ctypes.data, peq.ctypes.data, out.ctypes.data, 150, 12, 150, 256, 0, 4, wn, 27, None)
            assert np.array_equal(out, oracle.myers64(q[:4], s))
        assert uploads() == n0 + 2
    finally:
        L.bgsa_hip_set_auto_resident(1)
        L.bgsa_hip_bucket_release(None)


def test_resident_bucket_rewritten_behind_the_librarys_back_is_not_scored_stale(oracle):
    """A registered Peq range that the caller overwrites by other means than hip_handle_reads — a memmove of another
    preprocessed bucket, one group patched in place — must give the NEW content's scores through hip_cal_align_score and
    through align_hip (its locked path and the lock-free path that serves a thread's last row).  For a bucket of this size
    (1.9 MB <= 8 MiB) that holds in the DEFAULT mode and exactly: the library keeps the host bytes it uploaded and compares
    what every call uses (SURVEY 8(b) "Ownership"; BGSA_KNC/cal_mic.c:348-356).  Larger ranges get the sampled fingerprint
    (mode -1 here): it sees the whole-bucket rewrite and, as documented, not a patch that misses every sampled line;
    switching to strict mode afterwards must not serve rows of the old device copy either."""
    L = B.lib()
    nq, length, wn = 6, 150, 5
    n = 64 * 300                       # many more groups than fingerprint samples: some group holds no sampled line
    q = oracle.gen_reads(91, nq, length)
    s_a, _ = B.pad_rows(oracle.gen_reads(92, n, length))
    s_b, _ = B.pad_rows(oracle.gen_reads(93, n, length))
    s_a[:nq] = oracle.mutate(q, np.arange(nq), 94)
    s_b[:nq] = oracle.mutate(q, np.arange(nq) + 3, 95)
    want_a, want_b = oracle.myers64(q, s_a), oracle.myers64(q, s_b)
    assert not np.array_equal(want_a, want_b)
    L.bgsa_hip_select_algorithm(B.ALGO_MYERS)
    L.init_mapping_table()
    _sa, seq_a, qmapped = _host_seam_inputs(L, q, s_a)
    _sb, seq_b, _ = _host_seam_inputs(L, q, s_b)
    gw = B.group_words(B.ALGO_MYERS, wn)
    x = np.zeros(gw * (n // 64), dtype=np.uint32)       # the registered bucket
    y = np.zeros_like(x)                                 # another bucket, preprocessed elsewhere ("saved")

    def stale():
        c = ctypes.c_uint64()
        L.bgsa_hip_stale_ranges(ctypes.byref(c))
        return c.value

    def coarse():
        out = np.zeros((nq, n), dtype=np.int16)
        L.hip_cal_align_score(qmapped.ctypes.data, x.ctypes.data, out.ctypes.data, length, nq, length, n, 0, nq, wn, 27, None)
        return out

    def fine(chunk=41):
        out = np.zeros((nq, n), dtype=np.int16)
        groups = n // 64
        for i in range(nq):
            row = np.ascontiguousarray(qmapped[i * (length + 1):(i + 1) * (length + 1)])
            for j in range(0, groups, chunk):
                c = min(chunk, groups - j)
                L.align_hip(row.ctypes.data, x[gw * j:].ctypes.data, length, length, wn, c, i * groups + j, out.ctypes.data, None)
        return out

    try:
        L.hip_handle_reads(ctypes.byref(seq_a), x.ctypes.data, wn, 0, n)
        L.hip_handle_reads(ctypes.byref(seq_b), y.ctypes.data, wn, 0, n)
        s0 = stale()
        assert np.array_equal(coarse(), want_a) and np.array_equal(fine(), want_a) and stale() == s0
        # --- the whole bucket replaced by a memmove: no call told the library
        ctypes.memmove(x.ctypes.data, y.ctypes.data, x.nbytes)
        assert np.array_equal(coarse(), want_b)
        assert stale() == s0 + 1
        assert np.array_equal(fine(), want_b) and stale() == s0 + 1          # uploaded once, rows scored from the new copy
        # --- and back, this time the fine seam sees it first: every thread's last row (the lock-free path) and the
        # cached rows belong to the old content and must not be served
        y[:] = 0                                                             # the preprocess ORs bits in (cal_cpu.c:273)
        L.hip_handle_reads(ctypes.byref(seq_a), y.ctypes.data, wn, 0, n)     # y := bucket A (y itself is registered too)
        ctypes.memmove(x.ctypes.data, y.ctypes.data, x.nbytes)
        assert np.array_equal(fine(), want_a) and stale() == s0 + 2
        assert np.array_equal(coarse(), want_a) and stale() == s0 + 2
        # --- one group patched in place, away from every line the fingerprint samples: the default mode compares the
        # bytes themselves for a range of this size — through both seams, nothing switched on
        lines = x.nbytes // 64
        sampled, weyl = set(), 0                  # capi.hip: range_fingerprint — first line, last line, a golden-ratio sequence
        for j in range(min(lines, 66)):
            weyl = (weyl + 0x9E3779B97F4A7C15) & ((1 << 64) - 1)
            sampled.add(0 if j == 0 else (lines - 1 if j == 1 else (weyl * lines) >> 64))
        g_lines = gw * 4 // 64
        victims = [g for g in range(1, n // 64) if not any(g * g_lines <= ln < (g + 1) * g_lines for ln in sampled)]
        victim, victim2 = victims[0], victims[len(victims) // 2]
        y[:] = 0
        L.hip_handle_reads(ctypes.byref(seq_b), y.ctypes.data, wn, 0, n)     # y := bucket B
        assert x.nbytes <= 8 << 20
        x[gw * victim: gw * (victim + 1)] = y[gw * victim: gw * (victim + 1)]
        mixed = want_a.copy()
        mixed[:, 64 * victim: 64 * (victim + 1)] = want_b[:, 64 * victim: 64 * (victim + 1)]
        before = stale()
        assert np.array_equal(fine(), mixed) and stale() == before + 1
        assert np.array_equal(coarse(), mixed) and stale() == before + 1
        # --- the fingerprint alone (what a range above 8 MiB gets; mode -1): a second patch that misses every sampled line is
        # not seen — the documented limit of the best-effort check, and the reason the contract is in the header ...
        assert L.bgsa_hip_set_strict_resident(-1) == 0
        assert np.array_equal(coarse(), mixed)                               # re-uploaded under the new mode
        x[gw * victim2: gw * (victim2 + 1)] = y[gw * victim2: gw * (victim2 + 1)]
        mixed2 = mixed.copy()
        mixed2[:, 64 * victim2: 64 * (victim2 + 1)] = want_b[:, 64 * victim2: 64 * (victim2 + 1)]
        before = stale()
        assert np.array_equal(coarse(), mixed) and np.array_equal(fine(), mixed) and stale() == before
        # ... and switching strict mode on uploads the range again as new content: no row of the old device copy is served
        assert L.bgsa_hip_set_strict_resident(1) == 0
        assert np.array_equal(fine(), mixed2)
        assert np.array_equal(coarse(), mixed2)
        assert L.bgsa_hip_stream_faults(1) == 0
    finally:
        L.bgsa_hip_set_strict_resident(0)
        L.bgsa_hip_bucket_release(None)


def test_align_hip_grid_uses_the_row_cache(oracle):
    """The reference's grid (cal_cpu.c:63-84) through align_hip: every (query, chunk) pair is a call; with the
    bucket resident each query is scored once against the whole bucket and the other calls copy their chunk."""
    L = B.lib()
    q = oracle.gen_reads(71, 6, 150)
    s, _ = B.pad_rows(oracle.gen_reads(72, 64 * 7, 150))
    s[:6] = oracle.mutate(q, np.arange(6), 73)
    want = oracle.myers64(q, s)
    L.bgsa_hip_select_algorithm(B.ALGO_MYERS)
    L.init_mapping_table()
    sbuf, seq, qmapped = _host_seam_inputs(L, q, s)
    wn, n = 5, s.shape[0]
    gw = B.group_words(B.ALGO_MYERS, wn)
    peq = np.zeros(gw * (n // 64), dtype=np.uint32)
    L.hip_handle_reads(ctypes.byref(seq), peq.ctypes.data, wn, 0, n)

    def stats():
        h, m = ctypes.c_uint64(), ctypes.c_uint64()
        L.bgsa_hip_row_cache_stats(ctypes.byref(h), ctypes.byref(m))
        return h.value, m.value

    h0, m0 = stats()
    out = np.zeros((6, n), dtype=np.int16)
    chunk = 3                                              # 7 groups in chunks of 3, 3, 1 (cal_cpu.c:56-62)
    for i in range(6):
        row = np.ascontiguousarray(qmapped[i * 151:(i + 1) * 151])
        for j in range(0, 7, chunk):
            c = min(chunk, 7 - j)
            L.align_hip(row.ctypes.data, peq[gw * j:].ctypes.data, 150, 150, wn, c, i * 7 + j, out.ctypes.data, None)
    assert np.array_equal(out, want)
    h1, m1 = stats()
    assert m1 - m0 == 6 and h1 - h0 == 6 * 2               # one launch per query, two more calls served from its row
    # a rewritten bucket drops its rows: same query, new subjects
    s2, _ = B.pad_rows(oracle.gen_reads(74, 64 * 7, 150))
    sbuf2, seq2, _ = _host_seam_inputs(L, q, s2)
    peq[:] = 0
    L.hip_handle_reads(ctypes.byref(seq2), peq.ctypes.data, wn, 0, n)
    res = np.zeros(n, dtype=np.int16)
    L.align_hip(np.ascontiguousarray(qmapped[:151]).ctypes.data, peq.ctypes.data, 150, 150, wn, 7, 0, res.ctypes.data, None)
    assert np.array_equal(res, oracle.myers64(q[:1], s2)[0]) and stats()[1] == m1 + 1
    L.bgsa_hip_bucket_release(None)


def test_align_hip_reads_ahead_in_a_query_buffer(oracle):
    """align_hip walking a malloc_mem() query buffer row after row, as the reference's grid does: from the second miss
    on, the rows behind the requested one are scored in the same launch and land in the page-locked row arena; a row is
    only served for the same query bytes, so a query edited in place afterwards is scored again."""
    L = B.lib()
    nq, length = 40, 150
    q = oracle.gen_reads(81, nq, length)
    s, _ = B.pad_rows(oracle.gen_reads(82, 64 * 5, length))
    s[:nq] = oracle.mutate(q, np.arange(nq) % 9, 83)
    want = oracle.myers64(q, s)
    L.bgsa_hip_select_algorithm(B.ALGO_MYERS)
    L.init_mapping_table()
    sbuf, seq, qmapped = _host_seam_inputs(L, q, s)
    wn, n = 5, s.shape[0]
    gw = B.group_words(B.ALGO_MYERS, wn)
    peq = np.zeros(gw * (n // 64), dtype=np.uint32)
    L.hip_handle_reads(ctypes.byref(seq), peq.ctypes.data, wn, 0, n)
    block = L.malloc_mem(2 << 20)                      # page-locked, its extent known to the library
    try:
        qbuf = np.ctypeslib.as_array(ctypes.cast(block, ctypes.POINTER(ctypes.c_uint8)), shape=(2 << 20,))
        qbuf[:] = 0
        qbuf[: nq * (length + 1)] = qmapped[: nq * (length + 1)]

        def misses():
            h, m = ctypes.c_uint64(), ctypes.c_uint64()
            L.bgsa_hip_row_cache_stats(ctypes.byref(h), ctypes.byref(m))
            return m.value

        m0 = misses()
        out = np.zeros((nq, n), dtype=np.int16)
        for i in range(nq):
            L.align_hip(block + i * (length + 1), peq.ctypes.data, length, length, wn, n // 64, i * (n // 64), out.ctypes.data, None)
        assert np.array_equal(out, want)
        assert misses() - m0 == 2                      # query 0 alone, query 1 with 2..32; 33..39 were issued ahead of the calls
        # the query bytes decide, not the address: an edited row is scored again, its neighbours still come from their rows
        q2 = q.copy()
        q2[5] = oracle.gen_reads(84, 1, length)[0]
        _, _, qm2 = _host_seam_inputs(L, q2, s)
        qbuf[5 * (length + 1): 6 * (length + 1)] = qm2[5 * (length + 1): 6 * (length + 1)]
        res = np.zeros(n, dtype=np.int16)
        m1 = misses()
        L.align_hip(block + 5 * (length + 1), peq.ctypes.data, length, length, wn, n // 64, 0, res.ctypes.data, None)
        assert np.array_equal(res, oracle.myers64(q2[5:6], s)[0]) and misses() > m1
        m2 = misses()
        L.align_hip(block + 6 * (length + 1), peq.ctypes.data, length, length, wn, n // 64, 0, res.ctypes.data, None)
        assert np.array_equal(res, want[6]) and misses() == m2
    finally:
        L.bgsa_hip_bucket_release(None)
        L.free_mem(block)


_ROW_SCRIPT = r"""
import sys, ctypes, numpy as np
sys.path.insert(0, sys.argv[1])
import bgsa_amd as B, oracle as O
L = B.lib()
nq, length = 70, 150
q = O.gen_reads(91, nq, length); s, _ = B.pad_rows(O.gen_reads(92, 64 * 9, length))
s[:nq] = O.mutate(q, np.arange(nq) % 9, 93)
want = O.myers64(q, s)
L.bgsa_hip_select_algorithm(B.ALGO_MYERS); L.init_mapping_table()
table = np.ctypeslib.as_array((ctypes.c_uint32 * 128).in_dll(L, "mapping_table"))
sbuf = B.rows_to_buffer(s)
seq = B.SeqT(len=length, size=sbuf.size, count=s.shape[0], extra_size=0, extra_count=0, content=sbuf.ctypes.data)
qb = B.rows_to_buffer(q); keep = qb == 10; qm = table[qb].astype(np.uint8); qm[keep] = 10
wn, n = 5, s.shape[0]
peq = np.zeros(B.group_words(B.ALGO_MYERS, wn) * (n // 64), dtype=np.uint32)
L.hip_handle_reads(ctypes.byref(seq), peq.ctypes.data, wn, 0, n)
block = L.malloc_mem(1 << 16)
buf = np.ctypeslib.as_array(ctypes.cast(block, ctypes.POINTER(ctypes.c_uint8)), shape=(1 << 16,)); buf[:] = 0; buf[: qm.size] = qm
out = np.zeros((nq, n), dtype=np.int16)
order = list(range(nq)) + [5, 40, 69, 0]          # a walk, then a few rows again
for i in order:
    L.align_hip(block + i * (length + 1), peq.ctypes.data, length, length, wn, n // 64, i * (n // 64), out.ctypes.data, None)
assert np.array_equal(out, want)
h, m = ctypes.c_uint64(), ctypes.c_uint64(); L.bgsa_hip_row_cache_stats(ctypes.byref(h), ctypes.byref(m))
print("rows ok", m.value)
"""


@pytest.mark.parametrize("env", [{"BGSA_HIP_ROW_ARENA": "0"},                          # no page-locked arena: every row staged and copied
                                 {"BGSA_HIP_ROW_AHEAD": "1"},                          # one row per launch, nothing ahead of the calls
                                 {"BGSA_HIP_ROW_AHEAD": "7", "BGSA_HIP_ROW_ARENA": "0"},
                                 {"BGSA_HIP_ROW_AHEAD": "64"},
                                 {"BGSA_HIP_ROW_AHEAD": "128"},                        # the most the knob takes
                                 {}])                                                  # the default: 100 rows, the reference's query block
def test_align_hip_row_cache_variants(env):
    """align_hip walking a query buffer with the row cache's fallbacks and knobs (read once per process: child processes)."""
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(B.__file__).resolve().parent.parent
    p = subprocess.run([sys.executable, "-c", _ROW_SCRIPT, str(root)], env=dict(os.environ, **env), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "rows ok" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]
    launches = int(p.stdout.split()[-1])
    assert launches == (70 if env.get("BGSA_HIP_ROW_AHEAD") == "1" else 2)      # without read-ahead one launch per query; else query 0, then query 1 with its followers


_TWO_BUCKETS_SCRIPT = r"""
import sys, ctypes, threading, numpy as np
sys.path.insert(0, sys.argv[1])
import bgsa_amd as B, oracle as O
L = B.lib()
length, wn = 150, 5
L.bgsa_hip_select_algorithm(B.ALGO_MYERS); L.init_mapping_table()
table = np.ctypeslib.as_array((ctypes.c_uint32 * 128).in_dll(L, "mapping_table"))
def bucket(seed, groups):
    s, _ = B.pad_rows(O.gen_reads(seed, 64 * groups, length))
    sbuf = B.rows_to_buffer(s)
    seq = B.SeqT(len=length, size=sbuf.size, count=s.shape[0], extra_size=0, extra_count=0, content=sbuf.ctypes.data)
    peq = np.zeros(B.group_words(B.ALGO_MYERS, wn) * groups, dtype=np.uint32)
    L.hip_handle_reads(ctypes.byref(seq), peq.ctypes.data, wn, 0, s.shape[0])
    return s, peq, (sbuf, seq)
nq = 70
q = O.gen_reads(111, nq, length)
qb = B.rows_to_buffer(q); keep = qb == 10; qm = table[qb].astype(np.uint8); qm[keep] = 10
block = L.malloc_mem(1 << 16)
buf = np.ctypeslib.as_array(ctypes.cast(block, ctypes.POINTER(ctypes.c_uint8)), shape=(1 << 16,)); buf[:] = 0; buf[: qm.size] = qm
small, peq_small, keep_small = bucket(112, 3)      # rows of 384 bytes: the arena is carved for them (4 KiB slots)
big, peq_big, keep_big = bucket(113, 40)           # rows of 5,120 bytes: wider than the arena's slots
want_small, want_big = O.myers64(q, small), O.myers64(q, big)
def ask(peq, n, i, out):
    L.align_hip(block + i * (length + 1), peq.ctypes.data, length, length, wn, n // 64, 0, out.ctypes.data, None)
# a worker of the host's thread team scores one row of the small bucket and then idles: its thread-local last row
# keeps an arena slot for as long as the thread lives
scored, leave, errors = threading.Event(), threading.Event(), []
def idle_worker():
    out = np.zeros(small.shape[0], dtype=np.int16)
    ask(peq_small, small.shape[0], 3, out)
    if not np.array_equal(out, want_small[3]): errors.append("worker row")
    scored.set(); leave.wait()
    ask(peq_small, small.shape[0], 3, out)          # its last row must still be intact after the big bucket went by
    if not np.array_equal(out, want_small[3]): errors.append("worker row afterwards")
t = threading.Thread(target=idle_worker); t.start(); scored.wait()
# the main thread walks the bigger bucket (launches chained ahead of the calls), with rows of the small one in between
out_big, out_small = np.zeros((nq, big.shape[0]), dtype=np.int16), np.zeros((nq, small.shape[0]), dtype=np.int16)
for i in range(nq):
    ask(peq_big, big.shape[0], i, out_big[i])
    if i % 9 == 4:
        ask(peq_small, small.shape[0], i, out_small[i])
        assert np.array_equal(out_small[i], want_small[i]), ("small", i)
assert np.array_equal(out_big, want_big)
leave.set(); t.join()
assert not errors, errors
L.bgsa_hip_bucket_release(None); L.free_mem(block)
print("two buckets ok")
"""


@pytest.mark.parametrize("env", [{}, {"BGSA_HIP_ROW_ARENA": "0"}, {"BGSA_HIP_ROW_AHEAD": "7"}])
def test_align_hip_bigger_bucket_while_another_thread_holds_a_row(env):
    """A host thread that scored a row of a small bucket stays alive (an idle OpenMP worker keeps its thread-local last
    row, and with it a slot of the row arena) while the main thread walks a bucket whose rows are wider than the arena's
    slots, interleaved with rows of the small one.  The arena cannot be carved anew then: the rows must take the heap /
    staged path with each launch's OWN row size, the rows of the launch in flight must survive the read-ahead launch
    chained behind it, and nobody's row may be lost (round 2's code emptied the cache on every miss and died with 'the
    row just scored is not in the cache').  Child process: the arena is carved once per process."""
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(B.__file__).resolve().parent.parent
    p = subprocess.run([sys.executable, "-c", _TWO_BUCKETS_SCRIPT, str(root)], env=dict(os.environ, **env), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "two buckets ok" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]


def test_align_hip_from_many_threads(oracle):
    """Eight host threads hammer align_hip with random (query, chunk) requests on one bucket, as an OpenMP team with a
    dynamic schedule would: every chunk must come back right whatever the interleaving of misses, launches kept ahead
    of the calls, evictions and each thread's own last-row shortcut."""
    import threading
    L = B.lib()
    nq, length, groups = 90, 150, 8
    q = oracle.gen_reads(101, nq, length)
    s, _ = B.pad_rows(oracle.gen_reads(102, 64 * groups, length))
    s[:nq] = oracle.mutate(q, np.arange(nq) % 9, 103)
    want = oracle.myers64(q, s)
    L.bgsa_hip_select_algorithm(B.ALGO_MYERS)
    L.init_mapping_table()
    sbuf, seq, qmapped = _host_seam_inputs(L, q, s)
    wn, n = 5, s.shape[0]
    gw = B.group_words(B.ALGO_MYERS, wn)
    peq = np.zeros(gw * groups, dtype=np.uint32)
    L.hip_handle_reads(ctypes.byref(seq), peq.ctypes.data, wn, 0, n)
    block = L.malloc_mem(1 << 16)
    errors = []
    try:
        qbuf = np.ctypeslib.as_array(ctypes.cast(block, ctypes.POINTER(ctypes.c_uint8)), shape=(1 << 16,))
        qbuf[:] = 0
        qbuf[: qmapped.size] = qmapped

        def worker(seed):
            rng = np.random.default_rng(seed)
            res = np.zeros(n, dtype=np.int16)
            try:
                for it in range(600):
                    i = int(rng.integers(0, nq)) if it % 3 else (it // 3 + seed * 11) % nq     # random and walking requests mixed
                    j = int(rng.integers(0, groups))
                    c = int(rng.integers(1, groups - j + 1))
                    res[:] = 77
                    L.align_hip(block + i * (length + 1), peq[gw * j:].ctypes.data, length, length, wn, c, j, res.ctypes.data, None)
                    if not np.array_equal(res[64 * j: 64 * (j + c)], want[i, 64 * j: 64 * (j + c)]) or (res[: 64 * j] != 77).any() or (res[64 * (j + c):] != 77).any():
                        errors.append((seed, it, i, j, c))
                        return
            except Exception as e:   # noqa: BLE001
                errors.append(repr(e))

        threads = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors, errors[:3]
    finally:
        L.bgsa_hip_bucket_release(None)
        L.free_mem(block)


def test_wrong_word_num_is_refused(oracle):
    # the kernels index the blocks with the caller's word_num: anything but the layout's own value is an error
    import torch
    L = B.lib()
    a = B.DeviceAligner(B.ALGO_BANDED, k=8)
    a.set_queries(oracle.gen_reads(1, 2, 150))
    a.set_subjects(oracle.gen_reads(2, 64, 150))
    out = torch.empty((2, 64), dtype=torch.int8, device="cuda:0")
    p = a.params()
    for bad in (a.wn - 1, a.wn + 1, 4):     # 4 = the reference's banded value for 150 bp, k = 8 (host seams only)
        rc = L.bgsa_hip_cal_align_score_ex(ctypes.byref(p), a.d_content.data_ptr(), a.d_peq.data_ptr(), out.data_ptr(),
                                           150, 150, 64, 0, 2, bad, None, 0, None)
        assert rc == -1 and b"word_num" in L.bgsa_hip_last_error()


# ---- BitPAl with other integer scores: every compiled set against the
# Needleman-Wunsch oracle; the reference commits generator output for 2/-3/-5 only, so for the other
# sets the DP definition is the checker -------------------------------------------------------------
def _related(oracle, q, n, slen, seed):
    s = oracle.gen_reads(seed, n, slen)
    m = min(q.shape[1], slen)
    rows = min(24, n)
    s[:rows, :m] = oracle.mutate(q[np.arange(rows) % q.shape[0]][:, :m], np.arange(rows) * 3, seed + 1)
    s[rows - 1, : m // 3] = ord("N")
    return s


def _sets_under_test():
    """The compiled score sets; under BGSA_TEST_SETS=ab_only (the child pytest of test_score_sets_of_the_ab_flavour, which
    loads libbgsa_hip_ab.so) only those the default flavour does not carry."""
    import os
    if not B.LIB_PATH.exists():
        return []
    sets = B.score_sets()
    if os.environ.get("BGSA_TEST_SETS") == "ab_only":
        default = {(2, -3, -5), (1, -1, -2), (1, -4, -2), (10, -9, -15), (0, -1, -1)}
        sets = [x for x in sets if x not in default]
    return sets


@pytest.mark.parametrize("scores", _sets_under_test())
@pytest.mark.parametrize("qlen,slen", [(150, 150), (1, 1), (40, 70), (97, 33), (250, 256), (31, 225), (150, 257),
                                       (90, 300), (300, 600), (64, 513), (33, 1100)])
def test_bitpal_score_sets_vs_needleman_wunsch(oracle, scores, qlen, slen):
    q = oracle.gen_reads(100 + qlen, 5, qlen)
    s = _related(oracle, q, 130, slen, 200 + slen)
    got = B.align_all_pairs(q, s, algo=B.ALGO_BITPAL, scores=scores)
    assert np.array_equal(got, oracle.dp_nw(q, s, *scores))


@pytest.mark.parametrize("scores", [(0, -1, -1), (0, -3, -3), (1, -9, -2), (2, -9, -4), (1, -4, -2)])
@pytest.mark.parametrize("qlen,slen", [(150, 150), (60, 300), (97, 1500)])
def test_normalised_score_sets_vs_needleman_wunsch(oracle, scores, qlen, slen):
    """Edit-distance sets (0,-f,-f) run on the Myers body x f; a mismatch below two gaps runs as its
    mismatch = 2*gap instance (1/-9/-2 = 1/-4/-2).  Checker: the textbook DP with the scores as given."""
    if scores[0] != 0 and (1, -4, -2) not in B.score_sets():
        pytest.skip("1/-4/-2 not compiled in")
    q = oracle.gen_reads(100 + qlen, 5, qlen)
    s = _related(oracle, q, 130, slen, 200 + slen)
    got = B.align_all_pairs(q, s, algo=B.ALGO_BITPAL, scores=scores)
    assert np.array_equal(got, oracle.dp_nw(q, s, *scores))
    if scores[0] == 0:
        a = B.DeviceAligner(B.ALGO_BITPAL, scores=scores)
        a.set_queries(q)
        a.set_subjects(s)
        a._select()
        assert "myers" in B.lib().bgsa_hip_kernel_name(B.ALGO_BITPAL, a.wn).decode()
        B.lib().bgsa_hip_select_algorithm(B.ALGO_MYERS)


def test_bitpal_edit_scores_agree_with_the_myers_kernel(oracle):
    # the global route needs no compiled set: make_plan hands 0/-1/-1 (and every 0/-f/-f) to the Myers body
    q = oracle.gen_reads(5, 40, 150)
    s = _related(oracle, q, 640, 150, 6)
    assert np.array_equal(B.align_all_pairs(q, s, algo=B.ALGO_BITPAL, scores=(0, -1, -1)),
                          B.align_all_pairs(q, s, algo=B.ALGO_MYERS))
    # semi-global BitPAl (query end to end, free subject overhangs) is NOT the Myers kernels' semi-global mode (subject end to
    # end inside the query): it runs on the compiled 0/-1/-1 BitPAl instance, which every flavour of the library carries
    assert (0, -1, -1) in B.score_sets()
    for scores, f in (((0, -1, -1), 1), ((0, -2, -2), 2)):
        semi = B.align_all_pairs(q[:8], s[:192], algo=B.ALGO_BITPAL, scores=scores, semi_global=True)
        assert np.array_equal(semi, oracle.dp_semiglobal(q[:8], s[:192], *scores))


def test_score_sets_do_not_leak_between_aligners(oracle):
    # the score set is process-global in the C ABI (the reference's ints): two aligners with
    # different sets, used alternately, each keep their own
    sets = [x for x in B.score_sets() if x != (2, -3, -5)]
    if not sets:
        pytest.skip("only the default set is compiled in")
    q = oracle.gen_reads(21, 6, 120)
    s = _related(oracle, q, 128, 120, 22)
    a = B.DeviceAligner(B.ALGO_BITPAL)
    b = B.DeviceAligner(B.ALGO_BITPAL, scores=sets[-1])
    for x in (a, b):
        x.set_queries(q)
        x.set_subjects(s)
    for _ in range(2):
        assert np.array_equal(a.score().cpu().numpy(), oracle.bitpal(q, s))
        assert np.array_equal(b.score().cpu().numpy(), oracle.dp_nw(q, s, *sets[-1]))


def test_uncompiled_score_set_is_refused(oracle):
    with pytest.raises(B.BgsaHipError, match="BITPAL_SETS"):
        B.align_all_pairs(oracle.gen_reads(1, 2, 50), oracle.gen_reads(2, 64, 50), algo=B.ALGO_BITPAL, scores=(9, -9, -9))


# ---- semi-global BitPAl (generator option -s): DP definition as the checker ------------------------------
@pytest.mark.parametrize("scores", _sets_under_test())
@pytest.mark.parametrize("qlen,slen", [(60, 150), (150, 150), (97, 33), (1, 1), (20, 256), (100, 300), (50, 1000), (300, 777)])
def test_bitpal_semiglobal_vs_dp(oracle, scores, qlen, slen):
    q = oracle.gen_reads(300 + qlen, 4, qlen)
    s = oracle.gen_reads(400 + slen, 130, slen)
    if slen >= qlen:
        for r in range(24):   # the query, lightly edited, somewhere inside the subject
            off = (r * 13) % (slen - qlen + 1)
            s[r, off:off + qlen] = oracle.mutate(q[r % 4: r % 4 + 1], [r % 6], 500 + r)[0]
    got = B.align_all_pairs(q, s, algo=B.ALGO_BITPAL, scores=scores, semi_global=True)
    assert np.array_equal(got, oracle.dp_semiglobal(q, s, *scores))


def test_semiglobal_is_refused_for_other_algorithms_and_does_not_stick(oracle):
    q, s = oracle.gen_reads(1, 3, 80), oracle.gen_reads(2, 64, 80)
    with pytest.raises(B.BgsaHipError):
        B.DeviceAligner(B.ALGO_BANDED, k=8, semi_global=True)
    semi = B.align_all_pairs(q, s, algo=B.ALGO_BITPAL, semi_global=True)
    assert np.array_equal(semi, oracle.dp_semiglobal(q, s))
    # the mode is process-global in the C ABI; the next aligner sets its own
    assert np.array_equal(B.align_all_pairs(q, s, algo=B.ALGO_BITPAL), oracle.bitpal(q, s))
    assert np.array_equal(B.align_all_pairs(q, s, algo=B.ALGO_MYERS), oracle.myers64(q, s))
    L = B.lib()
    assert L.bgsa_hip_select_alignment(1) == 0
    try:
        # the entry point that reads the process-global mode must refuse banded while it says semi-global
        import torch
        a = B.DeviceAligner(B.ALGO_BANDED, k=8)
        a.set_queries(q)
        a.set_subjects(s)
        out = torch.empty((3, 64), dtype=torch.int8, device="cuda:0")
        rc = L.bgsa_hip_cal_align_score_dev(B.ALGO_BANDED, a.d_content.data_ptr(), a.d_peq.data_ptr(), out.data_ptr(),
                                            80, 80, 64, 0, 3, a.wn, 8, None, 0, None)
        assert rc == -2 and b"semi-global" in L.bgsa_hip_last_error()
        # ... while an aligner that passes its own parameters is not affected by the global at all
        assert np.array_equal(a.score().cpu().numpy(), oracle.banded64(q, s, 8))
    finally:
        assert L.bgsa_hip_select_alignment(0) == 0


# ---- the generator's `factor` (Main.java:213-267): reduced score sets and Myers +distance ----------------
@pytest.mark.parametrize("qlen,slen", [(150, 150), (60, 300)])
def test_scores_with_common_factor(oracle, qlen, slen):
    q = oracle.gen_reads(41, 5, qlen)
    s = _related(oracle, q, 130, slen, 42)
    got = B.align_all_pairs(q, s, algo=B.ALGO_BITPAL, scores=(4, -6, -10))
    assert np.array_equal(got, oracle.dp_nw(q, s, 4, -6, -10))
    assert np.array_equal(got, 2 * oracle.bitpal(q, s).astype(np.int32))
    semi = B.align_all_pairs(q, s, algo=B.ALGO_BITPAL, scores=(6, -9, -15), semi_global=True)
    assert np.array_equal(semi, oracle.dp_semiglobal(q, s, 6, -9, -15))
    if (0, -1, -1) in B.score_sets():
        assert np.array_equal(B.align_all_pairs(q, s, algo=B.ALGO_BITPAL, scores=(0, -3, -3)), 3 * oracle.myers64(q, s).astype(np.int32))


@pytest.mark.parametrize("slen", [150, 600, 2500])
def test_myers_positive_distance(oracle, slen):
    q = oracle.gen_reads(43, 4, 150)
    s = _related(oracle, q, 130, slen, 44)
    neg = B.align_all_pairs(q, s, algo=B.ALGO_MYERS)
    assert np.array_equal(neg, oracle.myers64(q, s))
    pos = B.align_all_pairs(q, s, algo=B.ALGO_MYERS, scores=(0, 1, 1))      # generator option -m 1
    assert np.array_equal(pos, -neg) and (pos >= 0).all()
    assert np.array_equal(B.align_all_pairs(q, s, algo=B.ALGO_MYERS), neg)   # and the sign does not stick
    assert np.array_equal(pos, -oracle.dp_edit(q, s).astype(np.int32))   # dp_edit follows the reference sign: -distance


@pytest.mark.parametrize("slen", [1, 17, 32, 33, 50, 64])
@pytest.mark.parametrize("ns", [128, 192, 64 * 5 + 9])
def test_short_subjects_two_groups_per_wave(oracle, slen, ns):
    """Subjects of <= 64 bp run two rows per stream token AND two subject groups per wave (myers_global_asm_kernel<NW, 2>):
    even and odd group counts (the last wave then holds one live group), query lengths on both sides of the subject's."""
    for qlen in (slen, 2 * slen + 3, max(1, slen // 2)):
        q = oracle.gen_reads(4000 + slen + qlen, 7, qlen)
        s = oracle.gen_reads(4100 + slen + ns, ns, slen)
        m = min(qlen, slen)
        s[:40, :m] = oracle.mutate(q[np.arange(40) % 7][:, :m], np.arange(40) % 5, 4200 + slen)
        a = B.DeviceAligner(B.ALGO_MYERS)
        a.set_queries(q)
        a.set_subjects(s)
        assert a.kernel_name().startswith("myers_global_asm_kernel<%d, 2>" % (1 if slen <= 32 else 2))
        assert np.array_equal(a.score().cpu().numpy()[:, :ns], oracle.myers64(q, s))


# ---- semi-global Myers (generator -m 0 -s): the subject end to end inside the query --------------------
@pytest.mark.parametrize("qlen,slen", [(200, 60), (150, 150), (33, 97), (1, 1), (500, 250), (1000, 300), (700, 1000), (300, 1024),
                                       (64, 32), (90, 33), (400, 768), (400, 769), (900, 800), (1300, 1100), (3000, 2500), (200, 4000),
                                       (900, 832), (850, 833), (1000, 900), (990, 961), (1100, 1023), (120, 1000)])
def test_myers_semiglobal_vs_dp(oracle, qlen, slen):
    q = oracle.gen_reads(600 + qlen, 4, qlen)
    s = oracle.gen_reads(700 + slen, 130, slen)
    if qlen >= slen:
        for r in range(24):   # the subject = a lightly edited window of a query
            off = (r * 17) % (qlen - slen + 1)
            s[r] = oracle.mutate(q[r % 4: r % 4 + 1, off:off + slen], [r % 6], 800 + r)[0]
    got = B.align_all_pairs(q, s, algo=B.ALGO_MYERS, semi_global=True)
    want = oracle.dp_edit_semiglobal(q, s)
    assert np.array_equal(got, want)
    if qlen >= slen:
        assert (got[np.arange(24) % 4, np.arange(24)] >= -10).all()     # found: about the planted edits, far from random (~ -0.45 slen)
    # and the mode does not leak into the next global call
    assert np.array_equal(B.align_all_pairs(q[:1], s, algo=B.ALGO_MYERS), oracle.myers64(q[:1], s))


def test_myers_semiglobal_kernel_families(oracle):
    # generated-asm kernels: resident Peq planes up to 1024 bp, column blocks beyond — any length
    L = B.lib()
    L.bgsa_hip_select_algorithm(B.ALGO_MYERS)
    L.bgsa_hip_select_alignment(1)
    try:
        assert L.bgsa_hip_kernel_name(B.ALGO_MYERS, 5).startswith(b"myers_semi_asm_kernel<5>")
        assert L.bgsa_hip_kernel_name(B.ALGO_MYERS, 24).startswith(b"myers_semi_asm_kernel<24>")
        assert L.bgsa_hip_kernel_name(B.ALGO_MYERS, 25).startswith(b"myers_semi_asm_kernel<25>")
        assert L.bgsa_hip_kernel_name(B.ALGO_MYERS, 26).startswith(b"myers_semi_asm_kernel<26>")     # round 5: the chains in turns keep
        assert L.bgsa_hip_kernel_name(B.ALGO_MYERS, 31).startswith(b"myers_semi_asm_kernel<32>")     # five Peq planes resident up to 32 words
        assert L.bgsa_hip_kernel_name(B.ALGO_MYERS, 32).startswith(b"myers_semi_asm_kernel<32>")     # (the code planes: BGSA_MYERS_PEQ_MAX_WORDS)
        assert L.bgsa_hip_kernel_name(B.ALGO_MYERS, 33).startswith(b"myers_blocked_kernel<")
        assert L.bgsa_hip_kernel_name(B.ALGO_MYERS, 125).startswith(b"myers_blocked_kernel<18, true, true>")
    finally:
        L.bgsa_hip_select_alignment(0)


# ---- BASELINE.json configs[0] on the GPU: 1k x 1k x 150 bp against the reference binary run live ------
def test_config0_hip_matches_live_reference(oracle):
    if not oracle.have_reference("original_cpu"):
        pytest.skip("reference binaries not built (oracle/_ref)")
    q = oracle.gen_reads(0xC0, 1000, 150)
    s = oracle.gen_reads(0xC1, 1000, 150)
    s[:100] = oracle.mutate(q[:100], np.arange(100) % 13, 0xC2)
    ref, _ = oracle.run_reference("original_cpu", q, s, threads=8)
    assert np.array_equal(B.align_all_pairs(q, s, algo=B.ALGO_MYERS), ref)
    if oracle.have_reference("original_avx2"):
        ref_bp, _ = oracle.run_reference("original_avx2", q[:200], s, threads=8)
        assert np.array_equal(B.align_all_pairs(q[:200], s, algo=B.ALGO_BITPAL), ref_bp)
    if oracle.have_reference("banded_cpu"):
        ref_bd, _ = oracle.run_reference("banded_cpu", q[:200], s, threads=8, k=8)
        assert np.array_equal(B.align_all_pairs(q[:200], s, algo=B.ALGO_BANDED, k=8), ref_bd)


# ---- BASELINE.json configs 1-4 at their FULL size (10k queries x 1M subjects x 150 bp = 1e10 pairs, 20 GB of
# scores; 1k x 125k x 1000 bp per GPU): far beyond the CPU oracle, so checked through properties that do not depend on size ----------
@pytest.mark.parametrize("algo,k,nq,ns,length", [(B.ALGO_MYERS, 0, 10_000, 1_000_000, 150),
                                                 (B.ALGO_BANDED, 8, 10_000, 1_000_000, 150),
                                                 (B.ALGO_BITPAL, 0, 10_000, 1_000_000, 150),
                                                 (B.ALGO_MYERS, 0, 1_000, 1_000_000, 1000)])   # configs[4]: the whole bucket on one GPU
def test_full_baseline_size_properties(oracle, algo, k, nq, ns, length):
    import torch
    dev = torch.device("cuda:0")
    ns_pad = (ns + 63) // 64 * 64
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + algo)
    letters = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    q_rows = letters[torch.randint(0, 4, (nq, length), generator=gen, device=dev)]
    s_rows = torch.full((ns_pad, length + 1), ord("\n"), dtype=torch.uint8, device=dev)
    s_rows[:, :length] = ord("N")
    s_rows[:ns, :length] = letters[torch.randint(0, 4, (ns, length), generator=gen, device=dev)]
    # plant query i as subject planted[i]; duplicate 4096 other subjects far away (other group, lane, block)
    planted = torch.randperm(ns, generator=gen, device=dev)[:nq]
    s_rows[planted, :length] = q_rows
    cand = torch.randperm(ns, generator=gen, device=dev)[:8192]
    is_planted = torch.zeros(ns, dtype=torch.bool, device=dev)
    is_planted[planted] = True
    src = cand[:4096]
    dst = (src + ns // 2 + 37) % ns
    ok = ~is_planted[dst] & ~is_planted[src]
    src, dst = src[ok], dst[ok]
    taken = torch.zeros(ns, dtype=torch.bool, device=dev)
    taken[src] = True
    keep = ~taken[dst]
    src, dst = src[keep], dst[keep]
    s_rows[dst] = s_rows[src]

    a = B.DeviceAligner(algo, "cuda:0", k)
    q_host = q_rows.cpu().numpy()
    a.set_queries(q_host)
    a.set_subject_rows_device(s_rows.reshape(-1), ns_pad, length, qlen=length)
    scores = a.score()
    torch.cuda.synchronize()
    assert scores.shape == (nq, ns_pad)
    idx = torch.arange(nq, device=dev)
    exact = {B.ALGO_MYERS: 0, B.ALGO_BANDED: 0, B.ALGO_BITPAL: 2 * length}[algo]
    assert bool((scores[idx, planted] == exact).all())                       # identical pair
    assert bool((scores[:, src] == scores[:, dst]).all())                     # same subject, other place, same score
    lo, hi = {B.ALGO_MYERS: (-length, 0), B.ALGO_BANDED: (0, 127), B.ALGO_BITPAL: (-5 * 2 * length, 2 * length)}[algo]
    n_passed = 0
    for q0 in range(0, nq, 1000):        # reductions in blocks of 1e9 elements
        blk = scores[q0:q0 + 1000, :ns]
        assert int(blk.min()) >= lo and int(blk.max()) <= hi
        if algo == B.ALGO_BANDED:    # the filter: random pairs are rejected, and what passes is within the band
            passed = blk != 127
            n_passed += int(passed.sum())
            assert not bool((passed & (blk > 2 * k + 1)).any())
    if algo == B.ALGO_BANDED:
        assert nq <= n_passed < nq * 50
    # a random sample of the 1e10 pairs against the oracle
    rng = np.random.default_rng(99 + algo)
    n_sample = 600 if length <= 256 else 150
    qi, sj = rng.integers(0, nq, n_sample), rng.integers(0, ns, n_sample)
    got = scores[torch.from_numpy(qi).to(dev), torch.from_numpy(sj).to(dev)].cpu().numpy()
    sub = s_rows[torch.from_numpy(sj).to(dev), :length].cpu().numpy()
    fn = {B.ALGO_MYERS: oracle.myers64, B.ALGO_BITPAL: oracle.bitpal, B.ALGO_BANDED: lambda x, y: oracle.banded64(x, y, k)}[algo]
    want = np.array([fn(q_host[i:i + 1], sub[n:n + 1])[0, 0] for n, i in enumerate(qi)])
    assert np.array_equal(got, want)
    # the padding reads never alias a real subject: all-'N' against ACGT is all mismatches
    if algo == B.ALGO_MYERS:
        assert bool((scores[:, ns:] == -length).all())


# ---- the measurement knobs select alternative kernels; each must give the same scores.  The knobs are
# read once per process, so every variant runs in a child process (one at a time) ----------------------
_KNOB_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
import bgsa_amd as B, oracle as O
q = O.gen_reads(901, 3, 310)
for slen in (150, 310, 700, 930, 1000, 1100, 2300):   # 930 / 1000 bp: 30 / 32 words, the carry chains in turns (code planes under BGSA_MYERS_PEQ_MAX_WORDS=28)
    s = O.gen_reads(902 + slen, 130, slen)
    m = min(310, slen)
    s[:10, :m] = O.mutate(q[np.arange(10) % 3][:, :m], np.arange(10), 903)
    assert np.array_equal(B.align_all_pairs(q, s, algo=B.ALGO_MYERS), O.myers64(q, s)), ("myers", slen)
    if slen <= 1100:
        assert np.array_equal(B.align_all_pairs(q, s, algo=B.ALGO_BITPAL), O.bitpal(q, s)), ("bitpal", slen)
qb = O.gen_reads(904, 3, 150); sb = O.gen_reads(905, 130, 150); sb[:10] = O.mutate(qb[np.arange(10) % 3], np.arange(10), 906)
for k in (8, 31):
    assert np.array_equal(B.align_all_pairs(qb, sb, algo=B.ALGO_BANDED, k=k), O.banded64(qb, sb, k)), ("banded", k)
for length, k in ((64, 8), (97, 12), (128, 4), (160, 15), (500, 8), (33, 3), (150, 11), (90, 1), (481, 8), (450, 8)):   # chunk boundaries, tails, every special-row position class
    qc = O.gen_reads(910 + length, 40, length); sc = O.gen_reads(911 + length, 64 * 6, length)
    sc[:64] = O.mutate(qc[np.arange(64) % 40], np.arange(64) % (2 * k + 6), 912)
    sc[64 * 3 + 5] = O.mutate(qc[7:8], [2], 913)[0]; sc[64 * 4 + 60] = O.mutate(qc[9:10], [k], 914)[0]   # lone survivors: regroup pass
    assert np.array_equal(B.align_all_pairs(qc, sc, algo=B.ALGO_BANDED, k=k), O.banded64(qc, sc, k)), ("banded", length, k)
for length, k, groups in ((150, 8, 5), (70, 12, 1), (200, 4, 7), (150, 14, 3), (200, 13, 1), (250, 24, 3)):   # an odd number of subject groups: the last wave of the two-groups-per-wave kernels holds one
    qc = O.gen_reads(920 + length, 37, length); sc = O.gen_reads(921 + length, 64 * groups, length)
    sc[:48] = O.mutate(qc[np.arange(48) % 37], np.arange(48) % (2 * k + 6), 922)
    sc[-3] = O.mutate(qc[5:6], [1], 923)[0]                             # a lone survivor in the last group: regroup pass there too
    assert np.array_equal(B.align_all_pairs(qc, sc, algo=B.ALGO_BANDED, k=k), O.banded64(qc, sc, k)), ("banded odd groups", length, k)
for slen in (150, 700, 1000):   # semi-global Myers: asm kernels by default (1000 bp: chains in turns; code planes under BGSA_MYERS_PEQ_MAX_WORDS), the compiler-scheduled one under BGSA_MYERS_IMPL=c
    s = O.gen_reads(907 + slen, 70, slen)
    assert np.array_equal(B.align_all_pairs(q, s, algo=B.ALGO_MYERS, semi_global=True), O.dp_edit_semiglobal(q, s)), ("semi", slen)
# a launch captured into a hipGraph and replayed twice: with the task counter the packer inside the graph zeroes it on every replay
import torch
for algo, want_fn in ((B.ALGO_MYERS, O.myers64), (B.ALGO_BITPAL, O.bitpal)):
    qg = O.gen_reads(931, 70, 150); sg = O.gen_reads(932, 64 * 40, 150)
    want = want_fn(qg, sg)
    a = B.DeviceAligner(algo); a.set_queries(qg); a.set_subjects(sg)
    outg = torch.zeros(want.shape, dtype=torch.int16, device="cuda:0")
    a.score(out=outg); torch.cuda.synchronize()          # the workspace is allocated outside the capture
    side = torch.cuda.Stream(); graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            a.score(out=outg)
    for _ in range(2):
        torch.cuda.synchronize(); outg.zero_(); graph.replay(); torch.cuda.synchronize()
        assert np.array_equal(outg.cpu().numpy(), want), ("graph replay", algo)
print("knobs ok")
"""


@pytest.mark.parametrize("env", [{"BGSA_MYERS_IMPL": "c", "BGSA_BITPAL_IMPL": "c", "BGSA_BANDED_IMPL": "c"},
                                 {"BGSA_BANDED_IMPL": "s"},                                   # straight-line banded rows (LDS-selected words)
                                 {"BGSA_BANDED_IMPL": "p"},                                   # the band held in place for k <= 11 (re-anchor events)
                                 {"BGSA_BANDED_IMPL": "a"},                                   # the funnel-shift row loop at every k (round 2's default)
                                 {"BGSA_BANDED_GROUPS": "1"},                                 # one-word windows, one subject group per wave
                                 {"BGSA_DYNAMIC_TASKS": "0"},                                 # static grids at every launch size
                                 {"BGSA_DYNAMIC_MIN_TASKS": "1"},                             # the task counter at every launch size (default: long launches only)
                                 {"BGSA_DYNAMIC_MIN_TASKS": "1", "BGSA_DYNAMIC_TASK_WORDS": "1"},   # ... with one- and two-query tasks
                                 {"BGSA_BANDED_DYNAMIC": "0"},                                # the banded kernel on its static grid (the default until round 4)
                                 {"BGSA_BANDED_PAIR_LOOP": "0"},                              # k >= 16: round 2's row loop
                                 {"BGSA_BANDED_PAIR_LOOP": "2", "BGSA_DYNAMIC_MIN_TASKS": "1"},   # ... two groups per wave (measured slower: A/B flavour), on the counter
                                 {"BGSA_BANDED_PUSH_SOLID": "0", "BGSA_BANDED_SOLID_MARGIN": "0"},   # any lane within the limit counts as a solid survivor, from the first test
                                 {"BGSA_BANDED_PUSH_SOLID": "48", "BGSA_BANDED_GROUPS": "2"},
                                 {"BGSA_BANDED_PUSH_MAX": "0"},                               # no survivor queue
                                 {"BGSA_BANDED_PUSH_MAX": "64", "BGSA_BANDED_PUSH_ROW": "0"},   # everything alive at the first test is queued
                                 {"BGSA_MYERS_MAX_PLAIN_WORDS": "8"},
                                 {"BGSA_MYERS_MAX_PLAIN_WORDS": "8", "BGSA_MYERS_BLOCK_FORM": "planes"},
                                 {"BGSA_MYERS_PEQ_MAX_WORDS": "8"},
                                 {"BGSA_MYERS_PEQ_MAX_WORDS": "28"},                          # 29 .. 32 words on the code planes (the default until round 5)
                                 {"BGSA_QUERY_TILE_MAX": "8", "BGSA_DYNAMIC_MIN_TASKS": "1"},    # counter launches with small query tiles
                                 {"BGSA_QUERY_TILE_MAX": "256", "BGSA_MYERS_LONG_TILE": "8", "BGSA_DYNAMIC_MIN_TASKS": "1"},
                                 {"BGSA_BLOCKED_WORKGROUPS": "96"}])
def test_measurement_knobs_do_not_change_results(env):
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(B.__file__).resolve().parent.parent
    env = dict(env)
    if _needs_ab_flavour(env):      # a kernel only the A/B flavour of the library carries (make -C bgsa_amd/csrc ab)
        assert B.LIB_AB_PATH.exists(), "libbgsa_hip_ab.so is not built"
        env["BGSA_HIP_LIB"] = str(B.LIB_AB_PATH)
    p = subprocess.run([sys.executable, "-c", _KNOB_SCRIPT, str(root)], env=dict(os.environ, **env),
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "knobs ok" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]


def _needs_ab_flavour(env):
    return (any(env.get(k, "")[:1] == "c" for k in ("BGSA_MYERS_IMPL", "BGSA_BITPAL_IMPL", "BGSA_BANDED_IMPL"))
            or env.get("BGSA_BANDED_IMPL", "")[:1] in ("s", "p") or "BGSA_MYERS_PEQ_MAX_WORDS" in env
            or env.get("BGSA_MYERS_BLOCK_FORM") == "planes" or env.get("BGSA_BANDED_PAIR_LOOP") == "2")


@pytest.mark.parametrize("env,algo,length,k", [({"BGSA_MYERS_IMPL": "c"}, 0, 150, 0), ({"BGSA_BANDED_IMPL": "s"}, 1, 150, 8),
                                                ({"BGSA_BANDED_IMPL": "p"}, 1, 150, 8), ({"BGSA_BANDED_IMPL": "c"}, 1, 150, 8),
                                                ({"BGSA_MYERS_PEQ_MAX_WORDS": "8"}, 0, 400, 0),
                                                ({"BGSA_MYERS_MAX_PLAIN_WORDS": "8", "BGSA_MYERS_BLOCK_FORM": "planes"}, 0, 400, 0)])
def test_default_library_refuses_knobs_for_kernels_it_does_not_carry(env, algo, length, k):
    """The default flavour ships the kernels that are defaults; a measurement knob that selects one of the alternatives must
    fail loudly there (and name the A/B build), never fall back to another kernel in silence."""
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(B.__file__).resolve().parent.parent
    script = ("import sys; sys.path.insert(0, sys.argv[1])\n"
              "import bgsa_amd as B, oracle as O\n"
              f"q = O.gen_reads(1, 3, {length}); s = O.gen_reads(2, 128, {length})\n"
              "try:\n"
              f"    B.align_all_pairs(q, s, algo={algo}, k={k})\n"
              "except B.BgsaHipError as e:\n"
              "    print('refused:', e)\n")
    p = subprocess.run([sys.executable, "-c", script, str(root)], env=dict(os.environ, **env), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "refused:" in p.stdout and "A/B flavour" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]


def test_score_sets_of_the_ab_flavour(oracle):
    """The default flavour compiles five BitPAl score sets; the two others of round 3 (1/-3/-2, 5/-4/-10) ship in the A/B flavour.  Their global and semi-global suites run here against that library, in a child pytest."""
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(B.__file__).resolve().parent.parent
    assert B.LIB_AB_PATH.exists(), "libbgsa_hip_ab.so is not built"
    env = dict(os.environ, BGSA_HIP_LIB=str(B.LIB_AB_PATH), BGSA_TEST_SETS="ab_only")
    p = subprocess.run([sys.executable, "-m", "pytest", str(root / "tests" / "test_gpu_parity.py"), "-m", "gpu", "-x", "-q", "-k",
                        "test_bitpal_score_sets_vs_needleman_wunsch or test_bitpal_semiglobal_vs_dp or "
                        "test_bitpal_edit_scores_agree_with_the_myers_kernel"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=str(root))
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    assert " passed" in p.stdout and " failed" not in p.stdout
    n = int(p.stdout.strip().splitlines()[-1].split(" passed")[0].split()[-1])
    assert n >= 2 * (11 + 8) + 1, p.stdout[-500:]          # two sets x (11 global + 8 semi-global cases) + the edit-set check


# ---- the streamed gather on one GPU: the side stream moves block i while the compute stream scores block i+1 ----
def test_gather_stream_overlaps_the_next_block(oracle):
    import time
    import torch
    from bgsa_amd.multi_gpu import ScoreGatherStream
    nq, ns, rows = 1200, 64 * 4096, 100
    q = oracle.gen_reads(301, nq, 150)
    s = oracle.gen_reads(302, ns, 150)
    a = B.DeviceAligner(B.ALGO_MYERS)
    a.set_queries(q)
    a.set_subjects(s)
    out = torch.empty((nq, ns), dtype=torch.int16, device="cuda:0")
    got = []
    gs = ScoreGatherStream(None, "cuda:0", [ns], torch.int16, block_rows=rows,
                           on_block=lambda i, t: got.append((i, t[: 64 * rows].cpu().numpy().copy())))
    a.score(0, rows, out=out[:rows])                      # warm-up (workspace allocation)
    torch.cuda.synchronize()
    k_done = [torch.cuda.Event(enable_timing=True) for _ in range(nq // rows)]
    start = torch.cuda.Event(enable_timing=True)
    start.record()
    t0 = time.perf_counter()
    for b in range(nq // rows):
        a.score(b * rows, (b + 1) * rows, out=out[b * rows:(b + 1) * rows])
        k_done[b].record()
        gs.submit(out[b * rows:(b + 1) * rows])
    issue_s = time.perf_counter() - t0
    gs.drain()
    torch.cuda.synchronize()
    gpu_s = start.elapsed_time(k_done[-1]) / 1e3
    # the blocks arrive complete, in order, in the reference's block layout (one device: [rows][ns] flat)
    want = oracle.myers64(q[:rows], s[:64])
    assert [i for i, _ in got] == list(range(nq // rows))
    assert np.array_equal(got[0][1].reshape(-1)[: 64], out[0, :64].cpu().numpy())
    assert np.array_equal(out[:rows, :64].cpu().numpy(), want)
    # the compute stream never waited for a transfer: back-to-back kernels, issued ahead of the GPU except where
    # the consumer callback (a device-to-host copy here, as a writer thread would do) holds the host
    per_kernel = start.elapsed_time(k_done[0]) / 1e3
    assert gpu_s < 1.5 * per_kernel * (nq // rows) + 0.05
    assert issue_s < gpu_s * 3 + 0.5
