#!/usr/bin/env python3
"""Global vs semi-global Myers throughput at a few lengths (GPU box): python3 scripts/semi_perf.py [length ...]"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bgsa_amd as B
import oracle as O

sizes = {150: (2000, 262144), 700: (1000, 65536), 1000: (1000, 65536), 2500: (300, 65536)}
for slen in [int(x) for x in sys.argv[1:]] or sorted(sizes):
    nq, ns = sizes.get(slen, (300, 65536))
    q = O.gen_reads(1, nq, slen)
    s = O.gen_reads(2, ns, slen)
    for semi in (False, True):
        a = B.DeviceAligner(B.ALGO_MYERS, semi_global=semi)
        a.set_queries(q)
        a.set_subjects(s)
        out = a.score()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            a.score(out=out)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        print(slen, "semi" if semi else "global", a.kernel_name(), round(nq * ns * slen * slen / dt / 1e9), "GCUPS", flush=True)
