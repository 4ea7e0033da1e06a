#!/usr/bin/env python3
"""GPU box, round 5: what the sharded run's block size costs in KERNEL time, measured on the one card there is.

At N ranks a rank of `bench.py --gpus N` scores all 10,000 queries against its slice of the 1M-subject bucket block by block
(the tile of a block is handed to the streamed gather while the next block is scored).  This times one rank's kernels for the
slices of N = 1, 2, 4, 8 — one launch over all queries, blocks of 1,000 queries, blocks of 100 (the reference's
REF_BUCKET_COUNT) — no transfer: the question is only how much of the big launch's rate a block launch keeps.
    python3 scripts/r05_block_rows.py > gpurun_out/r05_block_rows.txt
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import bgsa_amd as B  # noqa: E402
from bgsa_amd.multi_gpu import plan_shards  # noqa: E402

dev = torch.device("cuda:0")
nq, ns_total, length = 10_000, 1_000_000, 150
gen = torch.Generator(device=dev)
gen.manual_seed(5)
letters = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
q_rows = letters[torch.randint(0, 4, (nq, length), generator=gen, device=dev)]
q_host = q_rows.cpu().numpy()
for world in (1, 2, 4, 8):
    ns = plan_shards(ns_total, world)[0].count
    ns_pad = (ns + 63) // 64 * 64
    s_rows = torch.full((ns_pad, length + 1), ord("\n"), dtype=torch.uint8, device=dev)
    s_rows[:, :length] = letters[torch.randint(0, 4, (ns_pad, length), generator=gen, device=dev)]
    a = B.DeviceAligner(B.ALGO_MYERS, "cuda:0", 0)
    a.set_queries(q_host)
    a.set_subject_rows_device(s_rows.reshape(-1), ns_pad, length, qlen=length)
    out = torch.empty((nq, ns_pad), dtype=a.out_dtype, device=dev)
    cells = float(nq) * ns * length * length
    row = [f"N={world}: slice {ns:>9,d} subjects"]
    for rows in (nq, 1000, 100):
        def step():
            for lo in range(0, nq, rows):
                a.score(lo, min(nq, lo + rows), out=out[lo:lo + rows])
        step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            step()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        row.append(f"{'one launch' if rows == nq else f'blocks of {rows}'}: {ms:8.2f} ms = {cells / ms / 1e6:9.0f} GCUPS")
    a.check_faults()
    print(" | ".join(row), flush=True)
    del a, out, s_rows
    torch.cuda.empty_cache()
