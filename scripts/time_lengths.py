#!/usr/bin/env python3
"""Kernel time of one score() launch by read length, for whichever library BGSA_HIP_LIB names (A/B of two builds on one box):
    python3 scripts/time_lengths.py <algo 0|1|2> <queries> <subjects> <length> [<length> ...]"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402
import bgsa_amd as B  # noqa: E402


def main():
    algo, nq, ns = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    rng = np.random.default_rng(5)
    for length in (int(x) for x in sys.argv[4:]):
        q = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, (nq, length))]
        s = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, (ns, length))]
        a = B.DeviceAligner(algo, "cuda:0", 8)
        a.set_queries(q)
        a.set_subjects(s)
        out = torch.empty((nq, (ns + 63) // 64 * 64), dtype=a.out_dtype, device="cuda:0")
        a.score(out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(3):
            e0.record()
            a.score(out=out)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        cells = float(nq) * ns * length * length
        print(f"algo {algo} {length} bp: {best:.3f} ms = {cells / best / 1e6:.0f} GCUPS, checksum {int(out[:, :ns].to(torch.int64).sum())}", flush=True)


main()
