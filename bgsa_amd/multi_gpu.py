"""Subject-sharded multi-GPU driver: one process per GPU, torch.distributed (RCCL) for the two
exchanges the path has.

The reference's only multi-device code is the KNC backend: every device sees all queries
(`in(ref_content...)`, original/BGSA_KNC/cal_mic.c:121-128), each gets a contiguous slice of the
subject bucket sized by a ratio vector (`dispatch_task`, BGSA_KNC/global.c:374-431), and the host
concatenates per-device result blocks by offset (cal_mic.c:535-536), recording
`total_device_number` + `device_read_counts` in `.info` (cal_mic.c:476-478).  Here:

  * rank 0 broadcasts the mapped query buffer (C1 of SURVEY.md §2a) — one RCCL broadcast, ~1.5 MB;
  * every rank scores ALL queries against ITS subject slice with the single-GPU kernels — no
    collective inside the hot path, the (query x subject) grid is embarrassingly parallel;
  * optionally rank 0 gathers the [n_queries, slice] score tiles (C3) and lays them out either
    row-major over all subjects or in the reference's per-device block order.

`score_fn(queries, subjects) -> tensor [nq, ns]` is the only compute hook; the default is the HIP
path (bgsa_amd.DeviceAligner).  CPU tests inject a checker there to exercise the sharding and the
collectives under gloo — the product never does.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

V_NUM = 64


@dataclass
class Shard:
    start: int   # first subject of this rank
    count: int   # subjects of this rank (before padding to a multiple of 64)


def plan_shards(n_subjects: int, world: int, ratios=None) -> list[Shard]:
    """Contiguous slices, every slice but the last a multiple of 64 subjects.

    Equal ratios by default (homogeneous GPUs); `ratios` mirrors the reference's per-device ratio
    vector (BGSA_KNC/global.c:55-60,374-431).  The last rank takes the remainder, as the
    reference's last device does (global.c:420-428).
    """
    if world < 1:
        raise ValueError("world must be >= 1")
    ratios = [1.0] * world if ratios is None else [float(r) for r in ratios]
    if len(ratios) != world or min(ratios) <= 0:
        raise ValueError("need one positive ratio per rank")
    total = sum(ratios)
    shards, start = [], 0
    for r in range(world):
        if r == world - 1:
            count = n_subjects - start
        else:
            count = int(n_subjects * ratios[r] / total) // V_NUM * V_NUM
            count = min(count, n_subjects - start)
        shards.append(Shard(start, count))
        start += count
    return shards


class RatioBalancer:
    """Per-device work ratios re-estimated from the previous bucket's device times — the reference's
    `-D` dynamic mode (adjust_device_ratio3, original/BGSA_KNC/global.c:120-168): device 0 is the
    unit; device i's ratio is scaled by t0/ti, then smoothed by a weighted mean over the rounds so
    far in which round r weighs r.  Feed the result to plan_shards(ratios=...)."""

    def __init__(self, n_devices: int, ratios=None):
        self.ratios = [1.0] * n_devices if ratios is None else [float(r) for r in ratios]
        if len(self.ratios) != n_devices or min(self.ratios) <= 0:
            raise ValueError("need one positive ratio per device")
        self.history: list[list[float]] = []   # loop_device_ratio of the reference

    def update(self, times) -> list[float]:
        times = [float(x) for x in times]
        if len(times) != len(self.ratios) or min(times) <= 0:
            raise ValueError("need one positive time per device")
        new = list(self.ratios)
        new[0] = 1.0
        for i in range(1, len(new)):
            new[i] = self.ratios[i] * times[0] / times[i]
        rnd = len(self.history) + 1                      # time_index of the reference
        if rnd > 1:
            total = float(rnd)
            acc = [new[i] * rnd for i in range(len(new))]
            # the reference's loop starts at its second stored round (global.c:145: i = 1): round 1,
            # measured with the initial guess, does not enter the mean
            for r in range(1, rnd - 1):
                for i in range(1, len(new)):
                    acc[i] += self.history[r][i] * (r + 1)
                total += r + 1
            for i in range(1, len(new)):
                new[i] = acc[i] / total
        self.history.append(list(new))
        self.ratios = new
        return new


class ShardedAligner:
    def __init__(self, dist=None, device=None, score_fn=None, algo: int = 0, k: int = 0, scores=None,
                 semi_global: bool = False):
        """dist: the torch.distributed module with an initialised process group, or None for 1 rank.
        scores / semi_global: as bgsa_amd.DeviceAligner (BitPAl score set, generator -s)."""
        import torch
        self.torch = torch
        self.dist = dist
        self.rank = dist.get_rank() if dist is not None else 0
        self.world = dist.get_world_size() if dist is not None else 1
        self.device = torch.device(device) if device is not None else torch.device("cpu")
        self.algo, self.k, self.scores, self.semi_global = algo, k, scores, semi_global
        self.score_fn = score_fn if score_fn is not None else self._hip_score
        self._aligner = None

    # ---- default compute: the HIP path ----------------------------------------------------------
    def _hip_score(self, queries: np.ndarray, subjects: np.ndarray):
        import bgsa_amd as B
        if self._aligner is None:
            self._aligner = B.DeviceAligner(self.algo, str(self.device), self.k, self.scores, self.semi_global)
        a = self._aligner
        a.set_queries(queries)
        a.set_subjects(subjects)
        return a.score()[:, : a.ns_real]

    # ---- the two exchanges --------------------------------------------------------------------------
    def broadcast_queries(self, queries: np.ndarray | None, shape=None) -> np.ndarray:
        """Rank 0 passes the [nq, qlen] ASCII queries; every rank returns them."""
        torch = self.torch
        if self.dist is None:
            return np.ascontiguousarray(queries, dtype=np.uint8)
        meta = torch.zeros(2, dtype=torch.int64, device=self.device)
        if self.rank == 0:
            meta[0], meta[1] = queries.shape
        self.dist.broadcast(meta, src=0)
        nq, qlen = int(meta[0]), int(meta[1])
        buf = torch.empty((nq, qlen), dtype=torch.uint8, device=self.device)
        if self.rank == 0:
            buf.copy_(torch.from_numpy(np.ascontiguousarray(queries, dtype=np.uint8)))
        self.dist.broadcast(buf, src=0)
        return buf.cpu().numpy()

    def gather_scores(self, local, shards: list[Shard], layout: str = "row_major"):
        """Gather [nq, count_r] tiles on rank 0.

        layout "row_major": one [nq, n_subjects] tensor (subjects in file order);
        layout "device_blocks": the reference's result.txt order for one read bucket — device 0's
        [nq, count_0] block, then device 1's, ... (cal_mic.c:535-536) — as a flat tensor.
        """
        torch = self.torch
        if self.dist is None:
            return local if layout == "row_major" else local.reshape(-1)
        nq = local.shape[0]
        widest = max(s.count for s in shards)
        pad = torch.zeros((nq, widest), dtype=local.dtype, device=local.device)
        pad[:, : local.shape[1]] = local
        # scores travel as raw bytes: gloo has no int16/int8 gather, RCCL does not care
        raw = pad.view(torch.uint8)
        tiles = [torch.empty_like(raw) for _ in range(self.world)] if self.rank == 0 else None
        self.dist.gather(raw, tiles, dst=0)
        if self.rank != 0:
            return None
        parts = [t.view(local.dtype)[:, : s.count] for t, s in zip(tiles, shards)]
        if layout == "row_major":
            return torch.cat(parts, dim=1)
        return torch.cat([p.reshape(-1) for p in parts])

    # ---- one bucket end to end ------------------------------------------------------------------------
    def run(self, queries: np.ndarray | None, subjects_all: np.ndarray, gather: bool = True,
            layout: str = "row_major", ratios=None):
        """subjects_all: the whole bucket [ns, slen] (every rank holds or can read it, as the
        reference's host does); each rank slices its own shard.  Returns (result on rank 0 or the
        local tile when gather=False, shards)."""
        q = self.broadcast_queries(queries)
        shards = plan_shards(subjects_all.shape[0], self.world, ratios)
        mine = shards[self.rank]
        local = self.score_fn(q, subjects_all[mine.start: mine.start + mine.count])
        if not gather:
            return local, shards
        return self.gather_scores(local, shards, layout), shards
