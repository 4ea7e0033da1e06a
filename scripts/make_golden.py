#!/usr/bin/env python3
"""Mint tests/golden/*.npz from the REAL reference (oracle/_ref binaries).

Run in the build container only (needs /root/reference to have been compiled by
`make -C oracle ref`).  Each fixture is data: the input reads and the scores the reference's own
`aligner` + `convert -r` produced for them.  No reference source is stored.

    python scripts/make_golden.py            # regenerate every fixture
    python scripts/make_golden.py --check    # regenerate in memory and compare with the files

Fixture list follows SURVEY.md §8(c) "Fixtures to mint" (F1..F9), plus F0: the reference's own sample data.
"""
from __future__ import annotations

import argparse
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import oracle as O  # noqa: E402

GOLDEN = ROOT / "tests" / "golden"


def planted(q: np.ndarray, ns: int, max_edits: int, seed: int, base_seed: int) -> np.ndarray:
    """ns subjects: first half mutated copies of the queries (0..max_edits edits), rest random."""
    nq, length = q.shape
    s = O.gen_reads(base_seed, ns, length)
    half = ns // 2
    src = q[np.arange(half) % nq]
    s[:half] = O.mutate(src, np.arange(half) % (max_edits + 1), seed)
    return s


def with_specials(rows: np.ndarray, seed: int) -> np.ndarray:
    """Sprinkle N, lowercase and foreign bytes (all < 128) over the reads (fixture F5)."""
    rng = np.random.default_rng(seed)
    rows = rows.copy()
    n, length = rows.shape
    specials = np.frombuffer(b"NNNNacgtnXRY-*", dtype=np.uint8)
    mask = rng.random((n, length)) < 0.08
    rows[mask] = specials[rng.integers(0, len(specials), size=int(mask.sum()))]
    rows[0, :] = ord("N")  # an all-N read: what the reference pads with (file.c:100-110)
    return rows


SAMPLE_DATA = Path("/root/reference/original/BGSA_CPU/sample-data")   # the reference's only smoke data


def read_lines(path: Path) -> np.ndarray:
    rows = [ln for ln in path.read_bytes().split(b"\n") if ln]
    assert len({len(r) for r in rows}) == 1
    return np.frombuffer(b"".join(rows), dtype=np.uint8).reshape(len(rows), -1).copy()


def fixtures():
    """Yield (name, variant, k, queries, subjects)."""
    # F0: the reference's own sample data (3 x 500 bp queries, 128 x 500 bp subjects: the `./aligner -q sample-data/
    # query.txt -d sample-data/subject.txt` run of its README), scored by each compiled variant.  Data files, read
    # here in the build container only; when /root/reference is absent the committed fixtures are simply kept.
    if SAMPLE_DATA.exists():
        q, s = read_lines(SAMPLE_DATA / "query.txt"), read_lines(SAMPLE_DATA / "subject.txt")
        yield "f0_sample_myers", "original_cpu", None, q, s
        yield "f0_sample_bitpal", "original_avx2", None, q, s
        yield "f0_sample_banded_k8", "banded_cpu", 8, q, s
        yield "f0_sample_banded_k31", "banded_cpu", 31, q, s
    # F1 + F3: Myers 150 bp, random + planted near-duplicates (0..30 edits)
    q = O.gen_reads(0xB65A0001, 64, 150)
    s = planted(q, 256, 30, 11, 0xB65A1001)
    yield "f1_myers_150", "original_cpu", None, q, s
    # F2: Myers 1000 bp multi-word
    q = O.gen_reads(0xB65A0002, 16, 1000)
    s = planted(q, 128, 120, 12, 0xB65A1002)
    yield "f2_myers_1000", "original_cpu", None, q, s
    # F4: lengths around the 31/32/63/64-bit word boundaries
    for length in (1, 2, 30, 31, 32, 33, 62, 63, 64, 65, 93, 94, 95, 96, 97, 126, 127, 128, 129, 160, 161, 255, 256, 257):
        q = O.gen_reads(0xB65A0400 + length, 6, length)
        s = planted(q, 24, max(1, length // 4), 40 + length, 0xB65A1400 + length)
        yield f"f4_myers_len{length}", "original_cpu", None, q, s
    for length in (31, 32, 62, 63, 93, 94, 124, 125, 150):
        q = O.gen_reads(0xB65A0500 + length, 6, length)
        s = planted(q, 24, max(1, length // 4), 50 + length, 0xB65A1500 + length)
        yield f"f4_bitpal_len{length}", "original_avx2", None, q, s
    # F5: N / lowercase / foreign bytes
    q = with_specials(O.gen_reads(0xB65A0005, 12, 150), 5)
    s = with_specials(planted(q, 64, 20, 15, 0xB65A1005), 6)
    yield "f5_myers_specials", "original_cpu", None, q, s
    yield "f5_bitpal_specials", "original_avx2", None, q, s
    # F6: subject counts that are not a multiple of 64 (HIP-side padding)
    for ns in (1, 63, 65, 100):
        q = O.gen_reads(0xB65A0600 + ns, 5, 150)
        s = planted(q, ns, 10, 60 + ns, 0xB65A1600 + ns)
        yield f"f6_myers_ns{ns}", "original_cpu", None, q, s
    # F7: BitPAl 150 bp
    q = O.gen_reads(0xB65A0007, 64, 150)
    s = planted(q, 256, 30, 17, 0xB65A1007)
    yield "f7_bitpal_150", "original_avx2", None, q, s
    # F8: banded, equal lengths, planted edits so that non-127 outputs are exercised
    q = O.gen_reads(0xB65A0008, 16, 150)
    s = planted(q, 256, 23, 18, 0xB65A1008)
    for k in (4, 8, 16):
        yield f"f8_banded_k{k}_150", "banded_cpu", k, q, s
    # Lengths with slen mod 64 in [1, k] are excluded: there the reference's banded preprocess
    # writes one word past each plane (banded/BGSA_CPU/global.c:64-82 runs k characters past the
    # row while cal_cpu.c:253-254 sizes word_num for slen-k characters), so its output depends on
    # a heap overflow — undefined, not a parity target (DESIGN.md "banded domain").
    for length in (64, 73, 100, 128, 137, 192, 250, 500):
        q = O.gen_reads(0xB65A0800 + length, 6, length)
        s = planted(q, 48, 12, 80 + length, 0xB65A1800 + length)
        yield f"f8_banded_k8_len{length}", "banded_cpu", 8, q, s
    # F8b (round 4): the thresholds that run the funnel-shift rows — 13 and 15 on one 32-bit word, 24 and the reference's
    # default 31 on a register pair — on pairs with up to 3k edits, so that both outcomes (a distance, 127) are frequent.
    # Lengths as above: 150 mod 64 = 22 rules 150 bp out for k >= 22.
    for length, k in ((150, 13), (150, 15), (250, 24), (250, 31), (128, 31)):
        q = O.gen_reads(0xB65A0880 + length + k, 8, length)
        s = planted(q, 192, 3 * k, 88 + length + k, 0xB65A1880 + length + k)
        yield f"f8_banded_k{k}_len{length}", "banded_cpu", k, q, s
    # F9: qlen != slen (non-banded only; banded is degenerate there)
    for ql, sl in ((140, 150), (150, 140), (100, 200), (33, 31)):
        q = O.gen_reads(0xB65A0900 + ql, 8, ql)
        s = O.gen_reads(0xB65A1900 + sl, 40, sl)
        m = min(ql, sl)
        s[:16, :m] = O.mutate(q[np.arange(16) % 8][:, :m], np.arange(16) % 9, 90 + ql)
        yield f"f9_myers_{ql}x{sl}", "original_cpu", None, q, s
        yield f"f9_bitpal_{ql}x{sl}", "original_avx2", None, q, s


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    args = ap.parse_args()
    for v in O.REF_VARIANTS:
        if not O.have_reference(v):
            print(f"reference variant {v} not built: run `make -C oracle ref`", file=sys.stderr)
            return 2
    GOLDEN.mkdir(parents=True, exist_ok=True)
    bad = 0
    for name, variant, k, q, s in fixtures():
        scores, _ = O.run_reference(variant, q, s, threads=2, k=k)
        # Cross-variant pin: the SSE Myers build must agree with the scalar one.
        # (The SSE build pads the subject file to a multiple of 4 inside a 2x-file-size buffer and
        # aborts on very small files, so it is skipped below 8 subjects.)
        if variant == "original_cpu" and s.shape[0] >= 8:
            sse, _ = O.run_reference("original_sse", q, s, threads=2)
            assert (sse == scores).all(), f"{name}: BGSA_SSE != BGSA_CPU"
            # ... and so must the generator's AVX2 instance of the SSE kernel (oracle/derive_avx2_myers.py), the Myers
            # cpu_baseline of bench.py
            if O.have_reference("original_avx2_myers"):
                avx, _ = O.run_reference("original_avx2_myers", q, s, threads=2)
                assert (avx == scores).all(), f"{name}: the derived AVX2 Myers instance != BGSA_CPU"
        path = GOLDEN / f"{name}.npz"
        if args.check:
            old = np.load(path)
            same = (old["scores"] == scores).all() and (old["queries"] == q).all() and (old["subjects"] == s).all()
            print(f"{name}: {'ok' if same else 'MISMATCH'}")
            bad += not same
        else:
            np.savez_compressed(path, queries=q, subjects=s, scores=scores,
                                variant=np.array(variant), k=np.array(-1 if k is None else k))
            print(f"{name}: {variant} k={k} q={q.shape} s={s.shape} "
                  f"scores[{scores.min()},{scores.max()}] -> {path.stat().st_size} B")
    return 1 if bad else 0


if __name__ == "__main__":
    raise SystemExit(main())
