#!/usr/bin/env python3
"""GPU box: pinned-host <-> device copy rates by transfer size, one stream and two streams in parallel.
The Total-GCUPS leg of bench.py (config 3: 10 GB of int8 scores down) is bound by the device-to-host rate."""
import time
import torch

dev = torch.device("cuda:0")
for mb in (8, 32, 100, 200, 1000):
    n = mb << 20
    d = torch.empty(n, dtype=torch.uint8, device=dev)
    h = torch.empty(n, dtype=torch.uint8).pin_memory()
    for name, fn in (("D2H", lambda: h.copy_(d, non_blocking=True)), ("H2D", lambda: d.copy_(h, non_blocking=True))):
        fn(); torch.cuda.synchronize()
        reps = max(3, 2000 // mb)
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"{name} {mb:5d} MB x{reps}: {n * reps / dt / 1e9:6.1f} GB/s", flush=True)
# two D2H streams at once
n = 200 << 20
d = [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(2)]
h = [torch.empty(n, dtype=torch.uint8).pin_memory() for _ in range(2)]
st = [torch.cuda.Stream() for _ in range(2)]
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    for i in range(2):
        with torch.cuda.stream(st[i]):
            h[i].copy_(d[i], non_blocking=True)
torch.cuda.synchronize()
print(f"D2H 2 streams x 200 MB x10: {2 * n * 10 / (time.perf_counter() - t0) / 1e9:6.1f} GB/s", flush=True)
# D2H while a kernel runs
import bgsa_amd as B
import numpy as np
rng = np.random.default_rng(1)
q = rng.integers(0, 4, (1000, 150)).astype(np.uint8); s = rng.integers(0, 4, (1000000 // 64 * 64, 150)).astype(np.uint8)
lut = np.frombuffer(b"ACGT", dtype=np.uint8)
a = B.DeviceAligner(B.ALGO_MYERS)
a.set_queries(lut[q]); a.set_subjects(lut[s])
out = a.score(); torch.cuda.synchronize()
side = torch.cuda.Stream()
t0 = time.perf_counter()
a.score(out=out)
with torch.cuda.stream(side):
    for _ in range(10):
        h[0].copy_(d[0], non_blocking=True)
torch.cuda.synchronize()
print(f"D2H 10 x 200 MB beside a 1k x 1M Myers kernel: {time.perf_counter() - t0:.3f} s wall (kernel alone ~0.105 s, copies alone ~{10 * n / 50e9:.3f} s at 50 GB/s)", flush=True)
