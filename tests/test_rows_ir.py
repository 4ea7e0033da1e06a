"""The generated GPU row bodies, interpreted on the CPU.

bgsa_amd/csrc/rows_ir.py describes every DP row update as an instruction list; the same list
is emitted as gfx950 assembly (myers_rows_gen.inc / bitpal_rows_gen.inc) and can be executed by
a numpy interpreter.  Here the interpreter's scores must equal the oracle's bit for bit, which
pins the instruction stream itself (truth tables, carry chains, phase order) without a GPU.
"""
import sys
from pathlib import Path

import numpy as np
import pytest

sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "bgsa_amd" / "csrc"))
import rows_ir as R  # noqa: E402
import gen_rows_asm as G  # noqa: E402

from conftest import load_golden  # noqa: E402

ROOT = Path(__file__).resolve().parent.parent


def _inputs(oracle, seed, nq, ns, qlen, slen):
    q = oracle.gen_reads(seed, nq, qlen)
    s = oracle.gen_reads(seed + 1, ns, slen)
    m = min(qlen, slen)
    k = min(ns // 2, 24)
    s[:k, :m] = oracle.mutate(q[np.arange(k) % nq][:, :m], np.arange(k) % 9, seed)
    return q, s


@pytest.mark.parametrize("qlen,slen", [(150, 150), (33, 31), (64, 64), (20, 100), (150, 140), (97, 161)])
def test_myers_body_matches_oracle(oracle, qlen, slen):
    q, s = _inputs(oracle, 500 + slen, 3, 48, qlen, slen)
    nw = (slen + 31) // 32
    body = R.myers_body(nw)
    peq = R.build_peq32(s, nw)
    want = oracle.myers64(q, s)
    for i in range(q.shape[0]):
        st = R.myers_init_state(nw, 1, s.shape[0])
        R.run_rows(body, st, peq, q[i])
        assert np.array_equal(R.myers_score(st, nw, qlen, slen), want[i])


@pytest.mark.parametrize("qlen,slen,nw", [(300, 832, 26), (90, 896, 28), (1000, 801, 26), (64, 40, 2)])
def test_myers_parked_body_matches_oracle(oracle, qlen, slen, nw):
    """The 9-registers-per-word form of the Myers body (HN parked in the VP register): 26 and 28 words, 801..896 bp."""
    q = oracle.gen_reads(5100 + qlen, 2, qlen)
    s = oracle.gen_reads(5200 + slen, 24, slen)
    m = min(qlen, slen)
    s[:8, :m] = oracle.mutate(q[np.arange(8) % 2][:, :m], np.arange(8) * 3, 5300)
    s[3, : slen // 3] = ord("N")
    want = oracle.myers64(q, s)
    body = R.myers_parked_body(nw)
    peq = R.build_peq32(s, nw)
    for i in range(q.shape[0]):
        st = R.myers_init_state(nw, 1, s.shape[0])
        R.run_rows(body, st, peq, q[i])
        assert np.array_equal(R.myers_score(st, nw, qlen, slen), want[i])
    assert body.valu_count() == 8 * nw and R.count_hazard_nops(body) == 0 and body.allocate_temps()[1] == 2 * nw


@pytest.mark.parametrize("qlen,slen,nw", [(200, 60, 2), (150, 150, 5), (33, 97, 4), (1, 1, 1), (120, 64, 2), (300, 250, 8),
                                         (90, 257, 10), (150, 140, 5), (64, 32, 1), (40, 33, 2), (500, 300, 12)])
def test_myers_semi_body_matches_the_dp(oracle, qlen, slen, nw):
    """Semi-global Myers (generator -s): right-aligned subject, carry-in 0, running last-column score from
    the chains' final carries — against the textbook DP (subject end to end inside the query)."""
    q = oracle.gen_reads(2500 + qlen, 3, qlen)
    s = oracle.gen_reads(2600 + slen, 40, slen)
    if qlen >= slen:
        for r in range(12):
            off = (r * 17) % (qlen - slen + 1)
            s[r] = oracle.mutate(q[r % 3: r % 3 + 1, off:off + slen], [r % 6], 2700 + r)[0]
    s[3, : slen // 2] = ord("N")
    want = oracle.dp_edit_semiglobal(q, s)
    for i in range(q.shape[0]):
        assert np.array_equal(R.myers_semi_simulate(s, q[i], nw), want[i])
    assert R.count_hazard_nops(R.myers_semi_body(nw)) <= 2      # two wait states per row, of 10 nw + 3 instructions


@pytest.mark.parametrize("qlen,slen,nw", [(900, 800, 26), (70, 769, 26), (1100, 1024, 32), (400, 1000, 32), (300, 833, 28),
                                         (200, 930, 30), (64, 40, 2), (150, 150, 6)])
def test_myers_semi_planes_body_matches_the_dp(oracle, qlen, slen, nw):
    """The code-plane form of the semi-global body (subjects of 769..1024 bp): the unused low columns are coded 7,
    the code that matches every class, instead of being ORed into five masks."""
    q = oracle.gen_reads(2800 + qlen, 2, qlen)
    s = oracle.gen_reads(2900 + slen, 24, slen)
    if qlen >= slen:
        for r in range(8):
            off = (r * 17) % (qlen - slen + 1)
            s[r] = oracle.mutate(q[r % 2: r % 2 + 1, off:off + slen], [r % 6], 3000 + r)[0]
    s[3, : slen // 2] = ord("N")
    q[1, 5:9] = ord("N")
    want = oracle.dp_edit_semiglobal(q, s)
    for i in range(q.shape[0]):
        assert np.array_equal(R.myers_semi_planes_simulate(s, q[i], nw), want[i])
    body = R.myers_semi_planes_body(nw)
    assert body.valu_count() == 9 * nw + 3 and R.count_hazard_nops(body) <= 2


@pytest.mark.parametrize("qlen,slen", [(300, 300), (150, 150), (90, 257)])
def test_myers_planes_body_matches_oracle(oracle, qlen, slen):
    q, s = _inputs(oracle, 1500 + slen, 2, 40, qlen, slen)
    s[5, 3:9] = ord("N")
    q[1, 4:8] = ord("N")
    nw = (slen + 31) // 32
    body = R.myers_planes_body(nw)
    peq = R.build_peq32(s, nw)
    want = oracle.myers64(q, s)
    for i in range(q.shape[0]):
        st = R.myers_init_state(nw, 1, s.shape[0])
        R.run_rows(body, st, peq, q[i])
        assert np.array_equal(R.myers_score(st, nw, qlen, slen), want[i])


@pytest.mark.parametrize("qlen,slen,nwb", [(150, 150, 2), (70, 200, 3), (33, 97, 1), (64, 64, 1), (100, 300, 4), (31, 150, 2)])
def test_myers_column_blocks_with_carry_words(oracle, qlen, slen, nwb):
    """The > 1024 bp scheme at toy scale: blocks of nwb words, carries through per-32-row words."""
    q, s = _inputs(oracle, 3300 + slen + qlen, 2, 40, qlen, slen)
    s[0] = ord("A")
    q[0] = ord("A")                      # carries run through every word and every block
    want = oracle.myers64(q, s)
    for i in range(q.shape[0]):
        assert np.array_equal(R.myers_blocked_simulate(s, q[i], nwb), want[i])


@pytest.mark.parametrize("qlen,slen,nwb", [(150, 150, 2), (70, 200, 3), (33, 97, 1), (64, 64, 1), (40, 300, 4)])
def test_bitpal_column_blocks_with_carry_words(oracle, qlen, slen, nwb):
    q, s = _inputs(oracle, 4400 + slen + qlen, 2, 40, qlen, slen)
    s[0] = ord("A")
    q[0] = ord("A")
    want = oracle.bitpal(q, s)
    for i in range(q.shape[0]):
        assert np.array_equal(R.bitpal_blocked_simulate(s, q[i], nwb), want[i])


def test_make_blocked_agrees_with_the_hand_written_myers_block_body():
    """The hand-written column-block body is the mechanical transformation of the planes body: two chains since round 4's
    eight-instruction row (the addition, whose carries are [v_in = 2], and the HP shift with the row edge), which the hand-written
    form keeps in the first two of the kernel's three carry-word pairs."""
    nw = 3
    auto, init = R.make_blocked(R.myers_planes_body(nw), 2 * nw)
    assert init == [0, 1]
    hand = R.myers_block_body(nw)
    rng = np.random.default_rng(3)
    st_h = [rng.integers(0, 2**32, 64, dtype=np.uint32) for _ in range(2 * nw + 6)]
    for w in range(nw):
        st_h[2 * w + 1] &= ~st_h[2 * w]          # VP & VN == 0, the recurrence's invariant
    planes = [rng.integers(0, 2**32, 64, dtype=np.uint32) for _ in range(3 * nw)]
    pick = list(range(2 * nw)) + [2 * nw + 0, 2 * nw + 1, 2 * nw + 3, 2 * nw + 4]     # hand: CIN 0..2, COUT 0..2; auto: cin 0..1, cout 0..1
    for cls in range(5):
        a = [st_h[i].copy() for i in pick]
        b = [x.copy() for x in st_h]
        auto.simulate(a, [], cls=cls, planes=planes)
        hand.simulate(b, [], cls=cls, planes=planes)
        assert all(np.array_equal(x, b[i]) for x, i in zip(a, pick))
        assert np.array_equal(b[2 * nw + 2], st_h[2 * nw + 2]) and np.array_equal(b[2 * nw + 5], st_h[2 * nw + 5])   # the third pair: untouched


def test_myers_two_groups_per_wave(oracle):
    q, s = _inputs(oracle, 77, 2, 32, 150, 150)
    nw = 5
    body = R.myers_body(nw, groups=2)
    peq = R.build_peq32(s, nw)
    want = oracle.myers64(q, s)
    st = R.myers_init_state(nw, 2, s.shape[0])
    R.run_rows(body, st, peq, q[1], groups=2)
    for g in range(2):
        assert np.array_equal(R.myers_score(st, nw, 150, 150, group=g), want[1])


@pytest.mark.parametrize("qlen,slen", [(150, 150), (31, 31), (33, 32), (64, 65), (40, 100), (150, 140)])
def test_bitpal_body_matches_oracle(oracle, qlen, slen):
    q, s = _inputs(oracle, 900 + slen, 3, 48, qlen, slen)
    nw = (slen + 31) // 32
    body = R.bitpal_body(nw)
    peq = R.build_peq32(s, nw)
    want = oracle.bitpal(q, s)
    for i in range(q.shape[0]):
        st = R.bitpal_init_state(nw, s.shape[0])
        R.run_rows(body, st, peq, q[i])
        assert np.array_equal(R.bitpal_score(st, nw, qlen, slen), want[i])


def test_bitpal_body_on_golden_specials(oracle):
    g = load_golden("f5_bitpal_specials")
    q, s = g["queries"][:2], g["subjects"]
    nw = 5
    body = R.bitpal_body(nw)
    peq = R.build_peq32(s, nw)
    for i in range(2):
        st = R.bitpal_init_state(nw, s.shape[0])
        R.run_rows(body, st, peq, q[i])
        assert np.array_equal(R.bitpal_score(st, nw, 150, 150), g["scores"][i])


@pytest.mark.parametrize("length,k", [(150, 8), (150, 4), (150, 15), (64, 8), (65, 8), (100, 12), (200, 8),
                                      (150, 16), (150, 31), (100, 25), (250, 20), (70, 31)])
def test_banded_body_and_events_match_oracle(oracle, length, k):
    q = oracle.gen_reads(2500 + length + k, 4, length)
    s = oracle.gen_reads(2600 + length + k, 96, length)
    s[:48] = oracle.mutate(q[np.arange(48) % 4], np.arange(48) % (2 * k + 6), length + k)
    want = oracle.banded64(q, s, k)
    for i in range(q.shape[0]):
        assert np.array_equal(R.banded_simulate(s, q[i], k), want[i])


@pytest.mark.parametrize("length,k", [(150, 8), (150, 4), (64, 8), (65, 8), (200, 8), (150, 11), (33, 1), (100, 10), (31, 3), (481, 8)])
def test_banded_band_held_in_place_matches_oracle(oracle, length, k):
    """The alternative form of the 32-bit band (BGSA_BANDED_IMPL=p): classical left shifts inside a phase of
    banded_phase_rows(k) rows, re-anchor events between phases — same results as the sliding form and the oracle."""
    phase = R.banded_phase_rows(k)
    assert phase == 31 - 2 * k
    q = oracle.gen_reads(3500 + length + k, 3, length)
    s = oracle.gen_reads(3600 + length + k, 96, length)
    s[:48] = oracle.mutate(q[np.arange(48) % 3], np.arange(48) % (2 * k + 6), length + k)
    want = oracle.banded64(q, s, k)
    for i in range(q.shape[0]):
        assert np.array_equal(R.banded_simulate(s, q[i], k, phase=phase), want[i])
    assert R.banded_phase_rows(12) == 0 and R.banded_phase_body().valu_count() == 13
    assert not any(op.kind in ("lshr1", "alignbit") for op in R.banded_phase_body().ops)
    # the stream: the sliding form's tokens plus one re-anchor event in front of every phase but the first
    ev = [v for kind, v in R.banded_tokens(length, k, phase=phase) if kind == "event" and v & 16]
    assert len(ev) == (length - 1) // phase


@pytest.mark.parametrize("length,k", [(150, 8), (150, 4), (64, 8), (65, 8), (200, 8), (150, 9), (33, 1), (100, 12), (31, 3), (481, 8), (150, 11)])
@pytest.mark.parametrize("groups", [1, 2])
def test_banded_one_word_windows_match_oracle(oracle, length, k, groups):
    """The default form for k <= 12 (banded_cut_kernel<G>): the row shifts ONE register — the 32 bits of the match string
    cut at the last multiple of banded_cut_rows(k) rows — so it holds no half-rate instruction; cut / advance events
    between the rows; one or two subject groups behind every token.  Same results as the oracle."""
    cut = R.banded_cut_rows(k)
    assert cut == (16 if k <= 8 else 8) and cut - 1 + 2 * k + 1 <= 32
    q = oracle.gen_reads(4500 + length + k, 3, length)
    s = oracle.gen_reads(4600 + length + k, 96, length)
    s[:48] = oracle.mutate(q[np.arange(48) % 3], np.arange(48) % (2 * k + 6), length + k)
    want = oracle.banded64(q, s, k)
    for i in range(q.shape[0]):
        assert np.array_equal(R.banded_simulate(s, q[i], k, cut=cut, groups=groups), want[i])
    body = R.banded_cut_body(groups)
    assert body.valu_count() == 12 * groups and not any(op.kind == "alignbit" for op in body.ops)
    assert R.banded_cut_rows(13) == 0
    # the stream: a cut event in front of every cut-th row that does not start a new 32-row chunk
    ev = [v for kind, v in R.banded_tokens(length, k, cut=cut) if kind == "event" and v & 32]
    assert len(ev) == (length - 1) // cut - (length - 1) // 32


def test_emitted_asm_respects_the_vcc_hazard():
    for body in (R.myers_body(5), R.myers_body(3, groups=2), R.bitpal_body(5), R.myers_planes_body(12), R.myers_block_body(12)):
        lines = body.emit_asm(lambda name: name)
        since = 99
        for ln in lines:
            if ln.startswith("s_nop"):
                since += int(ln.split()[1]) + 1
                continue
            if ln.startswith("v_addc_co_u32"):
                assert since >= 2, "carry-in read fewer than 2 instructions after a VCC write"
            if ln.startswith(("v_addc_co_u32", "v_add_co_u32")) or ln.startswith("s_mov_b64 vcc"):
                since = 0
            else:
                since += 1
        # only fast-issue-class VALU opcodes may appear (scripts/ubench/valu_rate.hip)
        allowed = {"v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_mov_b32", "v_bitop3_b32",
                   "v_add_co_u32", "v_addc_co_u32", "s_nop", "s_mov_b64"}
        assert {ln.split()[0] for ln in lines} <= allowed


def test_temp_allocation_is_consistent():
    for body in (R.myers_body(5), R.bitpal_body(5)):
        slot_of, n = body.allocate_temps()
        assert n <= len(body.temps())
        # simulate with temporaries renamed to their slots: results must not change
        renamed = R.Body()
        for op in body.ops:
            f = lambda r: r if (not r or r[0] in "SE") else f"slot{slot_of[r]}"
            renamed.ops.append(R.Op(op.kind, f(op.dst), tuple(f(x) for x in op.srcs), op.imm))
        rng = np.random.default_rng(1)
        n_state = max(int(r[1:]) for op in body.ops for r in (op.dst,) + op.srcs if r.startswith("S")) + 1
        n_eq = max(int(r[1:]) for op in body.ops for r in op.srcs if r.startswith("E")) + 1
        st_a = [rng.integers(0, 2**32, 64, dtype=np.uint32) for _ in range(n_state)]
        eq = [rng.integers(0, 2**32, 64, dtype=np.uint32) for _ in range(n_eq)]
        st_b = [x.copy() for x in st_a]
        body.simulate(st_a, eq)
        renamed.simulate(st_b, eq)
        assert all(np.array_equal(a, b) for a, b in zip(st_a, st_b))


# ---- BitPAl for other integer scores: the generator's bodies against Needleman-Wunsch ----------------
SCORE_SETS = [(2, -3, -5), (0, -1, -1), (1, -1, -1), (1, -1, -2), (1, -3, -2), (1, -2, -1), (3, -2, -4),
              (5, -4, -10), (2, -6, -3), (1, -4, -2), (4, -1, -1), (1, -2, -1), (7, -1, -3)]


def _related(oracle, q, n, slen, seed):
    s = oracle.gen_reads(seed, n, slen)
    m = min(q.shape[1], slen)
    rows = min(24, n)
    s[:rows, :m] = oracle.mutate(q[np.arange(rows) % q.shape[0]][:, :m], np.arange(rows), seed + 1)
    s[rows - 1, : m // 3] = ord("N")
    return s


@pytest.mark.parametrize("scores", SCORE_SETS)
@pytest.mark.parametrize("qlen,slen", [(150, 150), (40, 70), (97, 33)])
def test_bitpal_any_scores_matches_needleman_wunsch(oracle, scores, qlen, slen):
    sc = R.BitpalScores(*scores)
    nw = (slen + 31) // 32
    body = R.bitpal_body(nw, sc)
    q = oracle.gen_reads(700 + qlen, 3, qlen)
    s = _related(oracle, q, 64, slen, 800 + slen)
    peq = R.build_peq32(s, nw)
    want = oracle.dp_nw(q, s, *scores)
    for i in range(q.shape[0]):
        st = R.bitpal_init_state(nw, s.shape[0], sc)
        R.run_rows(body, st, peq, q[i])
        assert np.array_equal(R.bitpal_score(st, nw, qlen, slen, sc), want[i])


@pytest.mark.parametrize("scores", [(0, -1, -1), (1, -3, -2), (5, -4, -10), (2, -6, -3)])
@pytest.mark.parametrize("qlen,slen,nwb", [(70, 200, 3), (33, 97, 1), (64, 64, 1)])
def test_bitpal_any_scores_column_blocks(oracle, scores, qlen, slen, nwb):
    sc = R.BitpalScores(*scores)
    q = oracle.gen_reads(900 + qlen, 2, qlen)
    s = _related(oracle, q, 64, slen, 950 + slen)
    want = oracle.dp_nw(q, s, *scores)
    for i in range(q.shape[0]):
        assert np.array_equal(R.bitpal_blocked_simulate(s, q[i], nwb, sc), want[i])


def test_bitpal_default_scores_instance():
    # 2/-3/-5: u in 0..12 on four unsigned planes (the reference keeps -u in five, align_core.c:191-214),
    # five value classes above the mismatch class, thirteen carry chains, 62 instructions per word (69 in round 3, 68 with
    # "u <= 7" read off plane 3; 64 with the cell identity: new u = max(w, u) - v_in needs no clamp and no "u <= 7" at all;
    # 63 with the single-use mask of u = 4 folded into its seed product; 62 without the run mask: the classes propagate through
    # every u = 0 column and the extraction masks with the top class)
    sc = R.BITPAL_DEFAULT
    assert (sc.planes, sc.chains, sc.C, sc.D, sc.K) == (4, 13, 12, 7, 5)
    assert sc.weights() == (1, 2, 4, 8)
    assert R.bitpal_body(1).valu_count() == 62


def test_bitpal_edit_scores_equal_negated_myers(oracle):
    q = oracle.gen_reads(11, 3, 100)
    s = _related(oracle, q, 64, 100, 12)
    assert np.array_equal(oracle.dp_nw(q, s, 0, -1, -1), oracle.myers64(q, s))


@pytest.mark.parametrize("scores", [(1, 1, -1), (2, -7, -3), (2, -3, 0), (1, -1, 1)])
def test_bitpal_rejects_score_sets_outside_the_method(scores):
    with pytest.raises(ValueError):
        R.BitpalScores(*scores)


def test_bitpal_set_generator_picks_widths_that_fit(tmp_path):
    import gen_rows_asm as G
    for scores in [(2, -3, -5), (5, -4, -10), (0, -1, -1)]:
        sc = R.BitpalScores(*scores)
        plain, blocks, _packed = G.bitpal_widths(sc)
        assert plain[0] == 1 and plain == list(range(1, plain[-1] + 1)) and plain[-1] <= 12 and blocks[-1] <= 8
        assert sc.planes * plain[-1] + 5 * plain[-1] + R.bitpal_body(plain[-1], sc).allocate_temps()[1] <= G.BITPAL_VGPR_BUDGET
    assert G.bitpal_widths(R.BITPAL_DEFAULT) == (list(range(1, 13)), [5, 6, 7, 8], False)   # 384 bp in registers (2 waves/SIMD)


# ---- semi-global BitPAl (generator option -s): same row body, other first row and last-row maximum ----
@pytest.mark.parametrize("scores", [(2, -3, -5), (0, -1, -1), (1, -3, -2), (5, -4, -10)])
@pytest.mark.parametrize("qlen,slen", [(60, 150), (150, 150), (97, 33), (20, 200)])
def test_bitpal_semiglobal_matches_dp(oracle, scores, qlen, slen):
    sc = R.BitpalScores(*scores)
    nw = (slen + 31) // 32
    body = R.bitpal_body(nw, sc)
    q = oracle.gen_reads(1700 + qlen, 3, qlen)
    s = oracle.gen_reads(1800 + slen, 64, slen)
    if slen >= qlen:   # plant the query (lightly edited) inside some subjects at various offsets
        for r in range(20):
            off = (r * 7) % (slen - qlen + 1)
            s[r, off:off + qlen] = oracle.mutate(q[r % 3: r % 3 + 1], [r % 5], 1900 + r)[0]
    peq = R.build_peq32(s, nw)
    want = oracle.dp_semiglobal(q, s, *scores)
    for i in range(q.shape[0]):
        st = R.bitpal_init_state(nw, s.shape[0], sc, semi=True)
        R.run_rows(body, st, peq, q[i])
        assert np.array_equal(R.bitpal_score(st, nw, qlen, slen, sc, semi=True), want[i])
        assert np.array_equal(R.bitpal_blocked_simulate(s, q[i], 2, sc, semi=True), want[i])
    if slen >= qlen:
        assert want[:, :20].max(axis=0).min() > want[:, 20:].max()   # the planted copies beat every random subject


@pytest.mark.parametrize("qlen,slen,nwb", [(150, 150, 2), (70, 200, 3), (33, 97, 1), (64, 64, 1), (100, 300, 4), (31, 150, 2)])
def test_myers_peq_resident_column_blocks(oracle, qlen, slen, nwb):
    # the block body derived mechanically from myers_body (Peq planes resident) with carry words
    q = oracle.gen_reads(40 + qlen, 2, qlen)
    s = oracle.gen_reads(50 + slen, 64, slen)
    m = min(qlen, slen)
    s[:8, :m] = oracle.mutate(q[np.arange(8) % 2][:, :m], np.arange(8), 60)
    want = oracle.myers64(q, s)
    for i in range(q.shape[0]):
        assert np.array_equal(R.myers_blocked_simulate(s, q[i], nwb, peq_resident=True), want[i])


# ---- the BitPAl generator over its whole stated domain (BitPAlGenerator.java:151-534 emits any integers) ----
def _random_score_sets(seed, count, max_k):
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < count:
        K = int(rng.integers(1, max_k + 1))
        M = int(rng.integers(0, 13))
        I = M - K
        G = min(-1, I // 2) - int(rng.integers(0, 7))      # gap < 0 and mismatch >= 2 gap (make_plan normalises the rest)
        if I >= 2 * G and G < 0:
            out.append((M, I, G))
    return out


def test_bitpal_generator_random_score_sets_match_the_dp(oracle):
    """240 random (match, mismatch, gap) with match - mismatch up to 30 — far beyond the five sets the library ships —
    through the generator's row body on the IR interpreter, against linear-gap Needleman-Wunsch (and the semi-global DP
    for every fourth set): one and two words, ragged lengths."""
    for it, (M, I, G) in enumerate(_random_score_sets(2024, 240, 30)):
        sc = R.BitpalScores(M, I, G)
        nw = 1 + it % 2
        rng = np.random.default_rng(it)
        qlen, slen = int(rng.integers(6, 34)), int(rng.integers(32 * (nw - 1) + 1, 32 * nw + 1))
        q = oracle.gen_reads(9000 + it, 1, qlen)
        s = oracle.gen_reads(9500 + it, 10, slen)
        if slen >= qlen:
            s[:5, :qlen] = oracle.mutate(np.repeat(q, 5, axis=0), np.arange(5), it)      # related pairs: the value classes get used
        semi = it % 4 == 3
        body = R.bitpal_body(nw, sc)
        st = R.bitpal_init_state(nw, s.shape[0], sc, semi)
        R.run_rows(body, st, R.build_peq32(s, nw), q[0])
        want = (oracle.dp_semiglobal if semi else oracle.dp_nw)(q, s, M, I, G)[0]
        assert np.array_equal(R.bitpal_score(st, nw, qlen, slen, sc, semi), want), (M, I, G, nw, semi)


@pytest.mark.parametrize("scores,nw_block,qlen,slen", [((10, -9, -15), 2, 70, 150), ((2, -3, -5), 2, 100, 150),
                                                      ((16, -15, -16), 1, 40, 70), ((7, -12, -13), 2, 50, 100)])
def test_bitpal_packed_carry_blocks_match_the_dp(oracle, scores, nw_block, qlen, slen):
    """Column blocks with the carries of a row packed into ceil(chains / 32) words (make_blocked_packed): what lets score
    sets with dozens of carry chains — 43 for 10/-9/-15, 67 for 16/-15/-16 — run at any subject length."""
    sc = R.BitpalScores(*scores)
    q = oracle.gen_reads(9700 + qlen, 2, qlen)
    s = oracle.gen_reads(9800 + slen, 24, slen)
    n = min(qlen, slen)
    s[:12, :n] = oracle.mutate(q[np.arange(12) % 2][:, :n], np.arange(12), 5)
    for semi in (False, True):
        want = (oracle.dp_semiglobal if semi else oracle.dp_nw)(q, s, *scores)
        for i in range(2):
            assert np.array_equal(R.bitpal_packed_blocked_simulate(s, q[i], nw_block, sc, semi), want[i]), (scores, semi, i)
    body, init, n_words = R.bitpal_packed_block_body(nw_block, sc)
    assert n_words == (sc.chains + 31) // 32 and not any(init)


def test_bitpal_generator_domain_is_stated_and_diagnosed(tmp_path):
    """Where `any integers` ends: sets with many chains take the packed-carry column blocks; a set whose ONE-word row
    body is over the register budget is refused with a message that names the quantity — by the generator and by
    `make BITPAL_SETS=...` (gen_bitpal_sets.py exits 1), not with a traceback (round 2: max() of an empty sequence)."""
    import subprocess
    assert G.bitpal_widths(R.BitpalScores(5, -4, -10))[2] is False          # 22 chains: a register pair per chain still fits
    for scores in ((10, -9, -15), (16, -15, -16)):     # two of the four sets round 2 crashed on
        plain, blocks, packed = G.bitpal_widths(R.BitpalScores(*scores))
        assert packed and plain and blocks
    with pytest.raises(G.BitpalDomainError, match="match - mismatch = 150"):
        G.bitpal_widths(R.BitpalScores(100, -50, -200))
    gen = ROOT / "bgsa_amd" / "csrc" / "gen_bitpal_sets.py"
    p = subprocess.run([sys.executable, str(gen), "--out", str(tmp_path), "100,-50,-200"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 1 and "match - mismatch = 150" in p.stderr and "Traceback" not in p.stderr
    p = subprocess.run([sys.executable, str(gen), "--out", str(tmp_path), "2,3,-5"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 1 and "match > mismatch" in p.stderr and "Traceback" not in p.stderr


@pytest.mark.parametrize("wide", [False, True])
def test_banded_funnel_rows_for_two_groups_equal_two_single_rows(wide):
    """The two-group funnel-shift rows of thresholds 13 .. 31 (round 4: banded_cut_kernel<2, ., funnel32 / funnel64>): the
    instruction list that ships — two groups' rows interleaved, carry chains kept whole, list-scheduled — must do to each group
    exactly what the one-group row (banded_body / banded_body64, CPU-pinned against the oracle above) does, and needs no s_nop."""
    rng = np.random.default_rng(17 + wide)
    n_state, n_eq = (5, 3) if wide else (3, 2)
    one = R.banded_body64() if wide else R.banded_body()
    two = R.schedule(R.banded_funnel_body(2, wide), 8)
    assert two.valu_count() == 2 * one.valu_count() and R.count_hazard_nops(two) == 0
    kinds = [op.kind for op in two.ops]
    assert kinds[: 2 * (2 if wide else 1)] == ["alignbit"] * (2 * (2 if wide else 1))      # the window shifts lead (they read the shift counter)
    for trial in range(20):
        st = [rng.integers(0, 2**32, 64, dtype=np.uint32) for _ in range(2 * n_state)]
        eq = [rng.integers(0, 2**32, 64, dtype=np.uint32) for _ in range(2 * n_eq)]
        sc = {"$sh": int(rng.integers(0, 32)), "$mask": 0x7FFFFFFF, "$mask_lo": 0xFFFFFFFF, "$mask_hi": int(rng.integers(1, 2**31)), "$one": 1}
        got = [x.copy() for x in st]
        two.simulate(got, eq, scalars=sc)
        for g in range(2):
            want = [x.copy() for x in st[g * n_state:(g + 1) * n_state]]
            one.simulate(want, eq[g * n_eq:(g + 1) * n_eq], scalars=sc)
            assert all(np.array_equal(a, b) for a, b in zip(want, got[g * n_state:(g + 1) * n_state])), (wide, trial, g)


def test_banded_pair_row_with_one_64_bit_shift_equals_the_shipped_pair_row():
    """Round 4: banded_body64_sh64 — D0 of the pair row in a fixed aligned register pair, D0 >> 1 as ONE v_lshrrev_b64 (21 VALU
    instead of 22; the kernel of thresholds 16 .. 31 runs it) — is banded_body64 (pinned to the oracle above) on every state."""
    rng = np.random.default_rng(64)
    one, new = R.banded_body64(), R.schedule(R.banded_funnel_body(1, True, sh64=True), 8)
    assert new.valu_count() == one.valu_count() - 1 == 21 and R.count_hazard_nops(new) == 0
    assert sum(op.kind == "shr64" for op in new.ops) == 1 and sum(op.kind == "alignbit" for op in new.ops) == 2
    text = new.emit_asm(lambda n: {"P0": "v2", "P1": "v3"}.get(n, "v9"))
    assert "v_lshrrev_b64 v[2:3], 1, v[2:3]" in text
    for trial in range(50):
        st = [rng.integers(0, 2**32, 64, dtype=np.uint32) for _ in range(5)]
        eq = [rng.integers(0, 2**32, 64, dtype=np.uint32) for _ in range(3)]
        sc = {"$sh": int(rng.integers(0, 32)), "$mask_lo": 0xFFFFFFFF, "$mask_hi": int(rng.integers(1, 2**31)), "$one": 1}
        want, got = [x.copy() for x in st], [x.copy() for x in st]
        one.simulate(want, eq, scalars=sc)
        new.simulate(got, eq, scalars=sc, fixed={})
        assert all(np.array_equal(a, b) for a, b in zip(want, got)), trial


@pytest.mark.parametrize("nw,groups", [(1, 1), (2, 1), (5, 1), (1, 2), (2, 2), (8, 1), (28, 1)])
def test_myers_eight_instruction_row_equals_the_ten_instruction_row(nw, groups):
    """Round 4: myers_body — eight instructions and two carry chains per word, HN read off the carries of VP + (VP & E) — against
    myers_body10 (rounds 1-3: ten and three), row by row on random states that keep the recurrence's invariant VP & VN = 0 and on
    random match masks; both are pinned to the oracle by the tests above, this pins them to each other bit for bit."""
    rng = np.random.default_rng(800 + 10 * nw + groups)
    old, new = R.myers_body10(nw, groups), R.myers_body(nw, groups)
    assert new.valu_count() == 8 * nw * groups and old.valu_count() == 10 * nw * groups
    assert R.count_hazard_nops(new) == 0 and new.allocate_temps()[1] <= 2 * nw
    assert sum(op.kind in ("add_co", "setc1") for op in new.ops) == 2 * groups        # two chains per group (three before)
    st = []
    for _ in range(nw * groups):
        vp = rng.integers(0, 2**32, 64, dtype=np.uint32)
        st += [vp, rng.integers(0, 2**32, 64, dtype=np.uint32) & ~vp]
    a, b = [x.copy() for x in st], [x.copy() for x in st]
    for row in range(200):
        eq = [rng.integers(0, 2**32, 64, dtype=np.uint32) & rng.integers(0, 2**32, 64, dtype=np.uint32) for _ in range(nw * groups)]
        old.simulate(a, eq)
        new.simulate(b, eq)
        assert all(np.array_equal(x, y) for x, y in zip(a, b)), row
        assert all(not (b[2 * w] & b[2 * w + 1]).any() for w in range(nw * groups))      # the invariant survives


# ---- round 5: the two carry chains in turns over blocks of K words (SAVECC / LOADCC), and the dependency-aware order ----
@pytest.mark.parametrize("qlen,slen,nw,split", [(120, 1000, 32, 9), (60, 930, 30, 9), (200, 1024, 32, 8), (150, 1000, 32, 4), (150, 150, 5, 2),
                                                (90, 257, 9, 3), (40, 1000, 32, 12)])
def test_myers_split_body_matches_oracle(oracle, qlen, slen, nw, split):
    """myers_body(split=K): phase A and phase B of K words at a time, the pausing chain parked in a scalar pair — the form
    that keeps five Peq planes resident at 30 / 32 words.  Against the oracle, with reads that drive the carries through
    every word (homopolymers) and N columns."""
    q = oracle.gen_reads(7100 + qlen, 2, qlen)
    s = oracle.gen_reads(7200 + slen, 24, slen)
    m = min(qlen, slen)
    s[:8, :m] = oracle.mutate(q[np.arange(8) % 2][:, :m], np.arange(8) * 3, 7300)
    s[3, : slen // 3] = ord("N")
    s[9] = ord("A")
    q[1] = ord("A")
    want = oracle.myers64(q, s)
    body = R.myers_body(nw, 1, split=split)
    peq = R.build_peq32(s, nw)
    for i in range(q.shape[0]):
        st = R.myers_init_state(nw, 1, s.shape[0])
        R.run_rows(body, st, peq, q[i])
        assert np.array_equal(R.myers_score(st, nw, qlen, slen), want[i])
    blocks = -(-nw // split)
    assert body.valu_count() == 8 * nw and body.salu_count() == 4 * (blocks - 1) + 1
    assert R.count_hazard_nops(body) == 0 and body.allocate_temps()[1] == 2 * min(split, nw)


@pytest.mark.parametrize("make,nw,planes", [(lambda: R.myers_body(5), 5, False), (lambda: R.myers_body(3, groups=2), 6, False),
                                             (lambda: R.myers_body(32, 1, split=9), 32, False),
                                             (lambda: R.myers_planes_body(32), 32, True), (lambda: R.myers_planes_body(30, split=8), 30, True)])
@pytest.mark.parametrize("gap,window", [(1, 12), (2, 24)])
def test_schedule_ilp_keeps_the_function_and_separates_dependent_instructions(make, nw, planes, gap, window):
    """rows_ir.schedule_ilp reorders a body so that no instruction issues right behind the one whose result it reads: the same
    function of (state, masks) on random inputs for every class, no back-to-back dependent pair left, at most a few more
    temporaries, no hazard padding the original did not need."""
    rng = np.random.default_rng(nw * 100 + gap)
    body = make()
    sched = R.schedule_ilp(body, gap, window)
    assert sorted((o.kind, o.dst, o.srcs) for o in body.ops) == sorted((o.kind, o.dst, o.srcs) for o in sched.ops)
    for cls in range(5):
        vp = [rng.integers(0, 2 ** 32, 64, dtype=np.uint32) for _ in range(nw)]
        vn = [rng.integers(0, 2 ** 32, 64, dtype=np.uint32) & ~vp[w] for w in range(nw)]
        eq = [rng.integers(0, 2 ** 32, 64, dtype=np.uint32) for _ in range(nw)]
        pl = [rng.integers(0, 2 ** 32, 64, dtype=np.uint32) for _ in range(3 * nw)]
        s1 = [x.copy() for w in range(nw) for x in (vp[w], vn[w])]
        s2 = [x.copy() for x in s1]
        body.simulate(s1, eq, cls, planes=pl if planes else None)
        sched.simulate(s2, eq, cls, planes=pl if planes else None)
        assert all(np.array_equal(a, b) for a, b in zip(s1, s2))
    assert R.dependent_pairs(body) > nw and R.dependent_pairs(sched) == 0
    assert sched.allocate_temps()[1] <= body.allocate_temps()[1] + 8
    assert sched.valu_count() == body.valu_count() and sched.salu_count() == body.salu_count()


@pytest.mark.parametrize("qlen,slen,nw,split", [(300, 1000, 32, 9), (1100, 960, 30, 9), (120, 832, 26, 9), (500, 1024, 32, 9), (200, 150, 5, 2)])
def test_myers_semi_split_body_matches_the_dp(oracle, qlen, slen, nw, split):
    """Semi-global Myers with the chains in turns (myers_semi_body(split = K)): the carries that leave the two chains are taken
    where the LAST block's phases end.  As written and scheduled (what the generator emits for 26 .. 32 words), against the DP."""
    q = oracle.gen_reads(9100 + qlen, 3, qlen)
    s = oracle.gen_reads(9200 + slen, 40, slen)
    if qlen >= slen:
        for r in range(8):
            off = (r * 7) % (qlen - slen + 1)
            s[r] = oracle.mutate(q[r % 3: r % 3 + 1, off: off + slen], [r % 5], r)[0]
    s[9] = ord("A")
    q[2] = ord("A")
    s[10, : slen // 3] = ord("N")
    want = oracle.dp_edit_semiglobal(q, s)
    for body in (R.myers_semi_body(nw, split), R.schedule_ilp(R.myers_semi_body(nw, split), 2, 24)):
        for i in range(q.shape[0]):
            assert np.array_equal(R.myers_semi_simulate(s, q[i], nw, body), want[i])
        assert body.valu_count() == 8 * nw + 3 and body.allocate_temps()[1] <= 2 * min(split, nw) + 2


@pytest.mark.parametrize("nw,split", [(32, 9), (30, 8), (8, 3), (5, 2)])
@pytest.mark.parametrize("balanced", [False, True])
def test_myers_body_variants_of_the_measurement_builds_are_the_same_function(nw, split, balanced):
    """The forms only `scripts/build_variant.sh` builds (round 5's A/Bs: the pausing chain parked in a VECTOR register — SAVEV /
    LOADV —, phases balanced at four instructions per carry link, a minimum link distance in the scheduler) compute what the
    shipped body computes, on random states for every class, as written and scheduled."""
    rng = np.random.default_rng(1000 * nw + split + balanced)
    ref = R.myers_body(nw)
    forms = [R.myers_body(nw, 1, split=split, park=park, balanced=balanced) for park in ("sgpr", "vgpr")]
    forms += [R.schedule_ilp(f, 2, 24, 4) for f in forms]
    for body in forms:
        for _ in range(3):
            vp = [rng.integers(0, 2 ** 32, 64, dtype=np.uint32) for _ in range(nw)]
            vn = [rng.integers(0, 2 ** 32, 64, dtype=np.uint32) & ~vp[w] for w in range(nw)]
            eq = [rng.integers(0, 2 ** 32, 64, dtype=np.uint32) for _ in range(nw)]
            s1 = [x.copy() for w in range(nw) for x in (vp[w], vn[w])]
            s2 = [x.copy() for x in s1]
            ref.simulate(s1, eq)
            body.simulate(s2, eq)
            assert all(np.array_equal(a, b) for a, b in zip(s1, s2))
    blocks = -(-nw // split)
    assert forms[1].valu_count() == 8 * nw + 4 * (blocks - 1) and forms[1].salu_count() == 1      # vector park: VALU instead of scalar moves
    assert forms[0].valu_count() == 8 * nw and forms[0].salu_count() == 4 * (blocks - 1) + 1
