#!/usr/bin/env python3
"""Randomised parity soak on the GPU box (not part of pytest): random algorithm / lengths / counts /
thresholds / score sets / alignment modes, every result compared with the oracle.

    python3 scripts/soak_parity.py [seconds] [seed]

Prints a progress line every ~20 s (the GPU box kills silent jobs) and a summary; exit code 1 on the
first mismatch, with the failing case spelled out so that it can be turned into a test."""
import os
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
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    sets = B.score_sets()
    t0 = last = time.time()
    done = {}
    case = 0
    while time.time() - t0 < budget:
        case += 1
        kind = rng.choice(["myers", "myers_semi", "myers_pos", "banded", "bitpal", "bitpal_semi", "bitpal_factor"],
                          p=[0.22, 0.08, 0.05, 0.2, 0.25, 0.12, 0.08])
        long_case = rng.random() < 0.12
        if kind == "banded":
            k = int(rng.integers(1, 32))
            lo = 2 * k + 2
            length = int(rng.integers(lo, max(lo + 1, 1200 if long_case else 400)))
            while 1 <= length % 64 <= k:      # the reference's own out-of-bounds domain
                length += 1
            qlen = slen = length
        else:
            k = 0
            hi = 4200 if long_case else 420
            qlen, slen = int(rng.integers(1, hi)), int(rng.integers(1, hi))
            if os.environ.get("SOAK_SLEN_RANGE"):     # focus on one kernel family, e.g. 769,1024: the code-plane kernels
                lo_s, hi_s = (int(x) for x in os.environ["SOAK_SLEN_RANGE"].split(","))
                slen = int(rng.integers(lo_s, hi_s + 1))
                qlen = int(rng.integers(1, 1400))
        nq = int(rng.integers(1, 9 if long_case else 40))
        ns = int(rng.integers(1, 200 if long_case else 700))
        q = O.gen_reads(int(rng.integers(1 << 30)), nq, qlen)
        s = O.gen_reads(int(rng.integers(1 << 30)), ns, slen)
        m = min(qlen, slen)
        rel = min(ns, 16)
        s[:rel, :m] = O.mutate(q[np.arange(rel) % nq][:, :m], rng.integers(0, 12, rel), int(rng.integers(1 << 30)))
        if rng.random() < 0.3:
            s[rng.integers(ns), : max(1, m // 4)] = ord("N")
        scores = None
        if kind == "myers":
            got, want = B.align_all_pairs(q, s, algo=B.ALGO_MYERS), O.myers64(q, s)
        elif kind == "myers_semi":
            got, want = B.align_all_pairs(q, s, algo=B.ALGO_MYERS, semi_global=True), O.dp_edit_semiglobal(q, s)
        elif kind == "myers_pos":
            got = B.align_all_pairs(q, s, algo=B.ALGO_MYERS, scores=(0, 1, 1))
            want = -O.myers64(q, s).astype(np.int32)
        elif kind == "banded":
            got, want = B.align_all_pairs(q, s, algo=B.ALGO_BANDED, k=k), O.banded64(q, s, k)
        else:
            scores = tuple(int(x) for x in sets[rng.integers(len(sets))])
            f = 1
            if kind == "bitpal_factor":
                f = int(rng.integers(2, 5))
                scores = tuple(f * x for x in scores)
                if scores[1] == 2 * scores[2]:  # a mismatch below two gaps runs as its mismatch = 2*gap instance
                    scores = (scores[0], scores[1] - int(rng.integers(0, 7)), scores[2])
                if rng.random() < 0.15:         # edit-distance sets run on the Myers body
                    scores = (0, -f, -f)
            semi = kind == "bitpal_semi"
            got = B.align_all_pairs(q, s, algo=B.ALGO_BITPAL, scores=scores, semi_global=semi)
            want = (O.dp_semiglobal if semi else O.dp_nw)(q, s, *scores)
        if not np.array_equal(got, want):
            bad = np.argwhere(got != want)
            print(f"MISMATCH case {case}: kind={kind} qlen={qlen} slen={slen} nq={nq} ns={ns} k={k} scores={scores} "
                  f"seed={seed}; first bad (q, s) = {bad[0].tolist()}: got {got[tuple(bad[0])]} want {want[tuple(bad[0])]}; "
                  f"{len(bad)} of {got.size} differ", flush=True)
            return 1
        done[kind] = done.get(kind, 0) + 1
        if time.time() - last > 20:
            last = time.time()
            print(f"[{last - t0:6.0f}s] {case} cases ok: {done}", flush=True)
    print(f"soak ok: {case} cases in {time.time() - t0:.0f} s, seed {seed}: {done}", flush=True)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
