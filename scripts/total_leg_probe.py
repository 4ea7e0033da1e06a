#!/usr/bin/env python3
"""GPU box: where the wall time of bench.py's Total-GCUPS leg goes for config 3 (banded, int8 scores):
kernels only, copies only, both pipelined — per block of 100 queries x 1M subjects."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bgsa_amd as B

dev = torch.device("cuda:0")
nq, ns, length, k = int(os.environ.get("NQ", "10000")), 1000000, 150, 8
ns_pad = (ns + 63) // 64 * 64
g = torch.Generator(device=dev); g.manual_seed(1)
lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
q = lut[torch.randint(0, 4, (nq, length), device=dev, generator=g)]
s = torch.full((ns_pad, length + 1), 10, dtype=torch.uint8, device=dev)
s[:, :length] = lut[torch.randint(0, 4, (ns_pad, length), device=dev, generator=g)]
a = B.DeviceAligner(B.ALGO_BANDED, str(dev), k)
a.set_queries(q.cpu().numpy())
a.set_subject_rows_device(s.reshape(-1), ns_pad, length, qlen=length)
d_out = [torch.empty((100, ns_pad), dtype=torch.int8, device=dev) for _ in range(2)]
h_out = [torch.empty((100, ns_pad), dtype=torch.int8).pin_memory() for _ in range(2)]
copy_stream = torch.cuda.Stream()
nb = nq // 100


def run(kernels, copies):
    done = [torch.cuda.Event() for _ in range(2)]
    copied = [torch.cuda.Event() for _ in range(2)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    host = 0.0
    for b in range(nb):
        slot = b & 1
        if b >= 2 and copies:
            torch.cuda.current_stream().wait_event(copied[slot])
        if kernels:
            h0 = time.perf_counter()
            a.score(b * 100, b * 100 + 100, out=d_out[slot])
            host += time.perf_counter() - h0
        done[slot].record()
        if copies:
            with torch.cuda.stream(copy_stream):
                copy_stream.wait_event(done[slot])
                h_out[slot].copy_(d_out[slot], non_blocking=True)
                copied[slot].record()
    issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    return wall / nb * 1e3, issue / nb * 1e3, host / nb * 1e3


for name, kk, cc in (("kernels only", True, False), ("copies only", False, True), ("pipelined", True, True)):
    run(kk, cc)
    w, i, h = run(kk, cc)
    print(f"{name:13s}: {w:6.3f} ms/block wall, host issue {i:6.3f} ms/block (score() calls {h:6.3f})", flush=True)

# the bench's own leg on the same data, same process
import bench
r = bench.total_gcups_leg(B.ALGO_BANDED, k, None, q.cpu().numpy(), s, ns, ns_pad, length, dev)
print("bench.total_gcups_leg:", r["wall_ms"], "ms", r["stages_ms"], flush=True)
r = bench.total_gcups_leg(B.ALGO_BANDED, k, None, q.cpu().numpy(), s, ns, ns_pad, length, dev)
print("bench.total_gcups_leg again:", r["wall_ms"], "ms", r["stages_ms"], flush=True)
w, i, h = run(True, True)
print(f"pipelined (own loop, after the bench legs): {w:6.3f} ms/block wall, host issue {i:6.3f} ms/block", flush=True)
copy_stream = torch.cuda.Stream()
w, i, h = run(True, True)
print(f"pipelined (own loop, NEW copy stream): {w:6.3f} ms/block wall, host issue {i:6.3f} ms/block", flush=True)
h_out = [torch.empty((100, ns_pad), dtype=torch.int8).pin_memory() for _ in range(2)]
w, i, h = run(True, True)
print(f"pipelined (own loop, NEW pinned buffers): {w:6.3f} ms/block wall, host issue {i:6.3f} ms/block", flush=True)
d_out = [torch.empty((100, ns_pad), dtype=torch.int8, device=dev) for _ in range(2)]
w, i, h = run(True, True)
print(f"pipelined (own loop, NEW device buffers): {w:6.3f} ms/block wall, host issue {i:6.3f} ms/block", flush=True)
a = B.DeviceAligner(B.ALGO_BANDED, str(dev), k)
a.set_queries(q.cpu().numpy())
a.set_subject_rows_device(s.reshape(-1), ns_pad, length, qlen=length)
w, i, h = run(True, True)
print(f"pipelined (own loop, NEW aligner): {w:6.3f} ms/block wall, host issue {i:6.3f} ms/block", flush=True)
