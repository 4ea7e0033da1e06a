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


class ScoreGatherStream:
    """The per-block score gather of a sharded run, streamed beside the kernels (C3 of SURVEY.md §2a).

    The reference's KNC host downloads every device's [queries-of-the-block, its subjects] result tile
    and appends them to the result file device after device (cal_mic.c:139-147, 535-536).  Here every
    rank hands the tile of a finished query block to `submit()`; a side stream waits for the kernel
    that produced it, and the tile goes to rank 0 with grouped point-to-point operations — each peer
    straight into its own segment of root's block buffer, which IS the reference's per-device block
    layout, so nothing is staged `world` times and nothing is padded to the widest shard.  Two block
    buffers alternate: while block i travels, the compute stream is already scoring block i+1; the
    caller's stream never waits for a transfer.  xGMI is point to point: the seven peers of an 8-GPU
    node reach root over seven different links at once.

    layout "device_blocks": root's buffer for one block = device 0's [rows, count_0], device 1's
    [rows, count_1], ... flat (what `result` holds, with `.info` recording the counts);
    layout "row_major": root additionally copies the segments into one [rows, sum(counts)] tile.
    """

    def __init__(self, dist, device, counts, dtype, block_rows: int, layout: str = "device_blocks", depth: int = 2,
                 on_block=None):
        """on_block(index, tensor): called on rank 0, in submission order, once block `index` is complete in
        the chosen layout (the tensor is only valid during the call: a consumer copies it out, e.g. to the
        pinned buffer its writer thread drains)."""
        import torch
        self.on_block = on_block
        self.unflushed = {}    # slot -> (block index, rows)
        self.torch = torch
        self.dist = dist
        self.rank = dist.get_rank() if dist is not None else 0
        self.world = dist.get_world_size() if dist is not None else 1
        self.device = torch.device(device)
        self.counts = [int(c) for c in counts]
        if len(self.counts) != self.world:
            raise ValueError("need one subject count per rank")
        if layout not in ("device_blocks", "row_major"):
            raise ValueError("layout must be device_blocks or row_major")
        self.layout, self.dtype, self.block_rows, self.depth = layout, dtype, int(block_rows), int(depth)
        self.cuda = self.device.type == "cuda"
        self.side = torch.cuda.Stream(device=self.device) if self.cuda else None
        # gloo has no point-to-point operations on device tensors: tiles then travel through host memory (the
        # 2-rank rehearsal of this class on one GPU box; RCCL moves them device to device)
        self.via_host = self.cuda and dist is not None and dist.get_backend() == "gloo"
        self.host_parts = {}   # slot -> [(segment view on the device, host tensor)] to copy in once received
        total = sum(self.counts)
        self.offsets = [sum(self.counts[:r]) * self.block_rows for r in range(self.world)]
        if self.rank == 0:
            self.blocks = [torch.empty(self.block_rows * total, dtype=dtype, device=self.device) for _ in range(self.depth)]
            self.tiles = ([torch.empty((self.block_rows, total), dtype=dtype, device=self.device) for _ in range(self.depth)]
                          if layout == "row_major" else None)
        else:   # a peer only needs a contiguous copy of its tile when the caller's view is strided
            self.staging = [None] * self.depth
        self.pending = [[] for _ in range(self.depth)]   # outstanding transfers per slot
        self.held = [None] * self.depth                  # the caller's tile (and its contiguous copy) of the slot's block
        self.n_submitted = 0
        self.blocks_checked = 0
        self.last = None    # root: (slot, rows) of the most recent block

    def _wait_slot(self, slot):
        for w in self.pending[slot]:
            w.wait()
        self.pending[slot] = []
        self.held[slot] = None                       # the transfers that read the tile are done
        for seg, host in self.host_parts.pop(slot, []):
            seg.copy_(host.view(seg.dtype).view(seg.shape), non_blocking=False)
        if slot in self.unflushed:
            index, rows = self.unflushed.pop(slot)
            if self.on_block is not None:
                if self.cuda:
                    self.side.synchronize()      # the consumer reads the block from the host side
                self.on_block(index, self._view(slot, rows))

    def _view(self, slot, rows):
        if self.layout == "row_major":
            return self.tiles[slot][:rows]
        return self.blocks[slot][: rows * sum(self.counts)]

    def submit(self, tile):
        """tile: this rank's [rows <= block_rows, count_rank] scores of one query block (device tensor, may
        be a strided view).  Returns at once.  The side stream reads the tile later than this call returns, so the
        stream keeps it alive itself: a reference per slot until the slot's transfers are done, and
        `record_stream` so that the caching allocator does not hand the block to the compute stream's next
        kernel meanwhile — callers may pass temporaries (run_streamed does).  A caller that REUSES one buffer
        for every block must still leave it untouched until drain() or until `depth` further blocks have been
        submitted."""
        torch = self.torch
        rows = int(tile.shape[0])
        if rows > self.block_rows or int(tile.shape[1]) != self.counts[self.rank]:
            raise ValueError("tile does not match this rank's shard")
        slot = self.n_submitted % self.depth
        self.n_submitted += 1
        ready = None
        if self.cuda:
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(self.device))
        ctx = torch.cuda.stream(self.side) if self.cuda else _null()
        with ctx:
            if self.cuda:
                self.side.wait_event(ready)          # the kernel that wrote the tile
                tile.record_stream(self.side)
            self._wait_slot(slot)                    # the slot's previous block has left / arrived
            self.held[slot] = [tile]
            ops = []
            if self.rank == 0:
                buf = self.blocks[slot]
                seg = lambda r: buf[self.offsets[r] * rows // self.block_rows: self.offsets[r] * rows // self.block_rows + rows * self.counts[r]]
                seg(0).view(rows, self.counts[0]).copy_(tile, non_blocking=True)
                for r in range(1, self.world):
                    if not self.counts[r]:
                        continue
                    if self.via_host:
                        host = torch.empty(seg(r).numel() * seg(r).element_size(), dtype=torch.uint8)
                        self.host_parts.setdefault(slot, []).append((seg(r), host))
                        ops.append(self.dist.P2POp(self.dist.irecv, host, r))
                    else:
                        ops.append(self.dist.P2POp(self.dist.irecv, seg(r).view(torch.uint8), r))
                self.last = (slot, rows)
            elif self.counts[self.rank]:
                src = tile
                if not tile.is_contiguous():
                    if self.staging[slot] is None:
                        self.staging[slot] = torch.empty((self.block_rows, self.counts[self.rank]), dtype=self.dtype, device=self.device)
                    src = self.staging[slot][:rows]
                    src.copy_(tile, non_blocking=True)
                flat = src.reshape(-1).view(torch.uint8)
                self.held[slot].append(flat)
                ops.append(self.dist.P2POp(self.dist.isend, flat.cpu() if self.via_host else flat, 0))
            if ops:
                self.pending[slot] = self.dist.batch_isend_irecv(ops)
            if self.rank == 0 and self.layout == "row_major":
                for w in self.pending[slot]:
                    w.wait()
                self.pending[slot] = []
                for seg_dev, host in self.host_parts.pop(slot, []):
                    seg_dev.copy_(host.view(seg_dev.dtype).view(seg_dev.shape), non_blocking=False)
                col = 0
                for r in range(self.world):
                    c = self.counts[r]
                    start = self.offsets[r] * rows // self.block_rows
                    self.tiles[slot][:rows, col:col + c].copy_(self.blocks[slot][start:start + rows * c].view(rows, c), non_blocking=True)
                    col += c
            if self.rank == 0:
                self.unflushed[slot] = (self.n_submitted - 1, rows)

    def drain(self):
        """Waits for every outstanding block (host-side)."""
        torch = self.torch
        ctx = torch.cuda.stream(self.side) if self.cuda else _null()
        with ctx:
            order = sorted(range(self.depth), key=lambda sl: self.unflushed.get(sl, (1 << 60, 0))[0])
            for slot in order:                       # flush in submission order
                self._wait_slot(slot)
        if self.cuda:
            self.side.synchronize()
        self.blocks_checked = self.n_submitted

    def last_block(self):
        """Root, after drain(): the most recent block in the chosen layout."""
        if self.rank != 0 or self.last is None:
            return None
        slot, rows = self.last
        return self._view(slot, rows)


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


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
        if getattr(self, "_resident", None) is not subjects:   # the shard stays in HBM across query blocks
            a.set_subjects(subjects)
            self._resident = subjects
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

    def run_streamed(self, queries, subjects_all, block_rows: int = 100, layout: str = "device_blocks", ratios=None,
                     score_block=None):
        """The bucket scored block by block (block_rows queries, the reference's REF_BUCKET_COUNT) with the
        gather of block i streamed beside the scoring of block i+1 (ScoreGatherStream).  Returns, on rank 0,
        the list of per-block results in `layout` (host numpy arrays), else None; and the shards.
        score_block(q_block, subjects_slice) -> [rows, count] tensor defaults to score_fn."""
        torch = self.torch
        q = self.broadcast_queries(queries)
        shards = plan_shards(subjects_all.shape[0], self.world, ratios)
        mine = shards[self.rank]
        sub = subjects_all[mine.start: mine.start + mine.count]
        fn = score_block if score_block is not None else self.score_fn
        blocks = []
        first = fn(q[:min(block_rows, q.shape[0])], sub)
        gs = ScoreGatherStream(self.dist, first.device, [s.count for s in shards], first.dtype, block_rows, layout,
                               on_block=lambda i, t: blocks.append(t.cpu().numpy().copy()))
        gs.submit(first)
        for lo in range(block_rows, q.shape[0], block_rows):
            gs.submit(fn(q[lo:lo + block_rows], sub))
        gs.drain()
        return (blocks if self.rank == 0 else None), shards

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
        if self.dist is None or self.world == 1:
            # one rank: its tile IS the result — no block buffers, no copies (device_blocks = the one device's block, flat)
            return (local if layout == "row_major" else local.reshape(-1)), shards
        # the gather is the streamed one with a single block of all the queries: every peer's tile goes straight
        # into its segment of root's buffer (nothing padded to the widest shard, nothing staged `world` times).
        # depth 1: one block, one buffer; the stream object is dropped behind this call, so the buffer is handed out as is
        gs = ScoreGatherStream(self.dist, local.device, [s.count for s in shards], local.dtype, max(int(local.shape[0]), 1), layout,
                               depth=1)
        gs.submit(local)
        gs.drain()
        return gs.last_block(), shards
