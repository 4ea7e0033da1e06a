"""The N > 1 path on CPU: world_size-2 (and 3) gloo groups exercise the subject sharding, the
query broadcast and the score gather of bgsa_amd.multi_gpu.  The compute hook is the oracle here
(test infrastructure); on GPUs it is the HIP path."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from bgsa_amd.multi_gpu import RatioBalancer, ShardedAligner, plan_shards  # noqa: E402


def test_plan_shards_properties():
    for n, world in ((1_000_000, 8), (1000, 2), (130, 3), (64, 4), (5, 2), (0, 2)):
        shards = plan_shards(n, world)
        assert len(shards) == world and sum(s.count for s in shards) == n
        pos = 0
        for i, s in enumerate(shards):
            assert s.start == pos and s.count >= 0
            if i < world - 1:
                assert s.count % 64 == 0  # every slice but the last is whole wavefront groups
            pos += s.count
    eight = plan_shards(1_000_000, 8)
    assert max(s.count for s in eight) - min(s.count for s in eight) < 64 * 8
    skew = plan_shards(64 * 100, 2, ratios=[1, 3])
    assert skew[0].count == 64 * 25 and skew[1].count == 64 * 75
    with pytest.raises(ValueError):
        plan_shards(10, 2, ratios=[1, 0])


def test_ratio_balancer_follows_device_times():
    # a device that takes twice as long gets half the work; repeated identical measurements converge
    b = RatioBalancer(3)
    r1 = b.update([1.0, 2.0, 0.5])
    assert r1 == [1.0, 0.5, 2.0]
    for _ in range(6):
        # times measured with the new split: device i now needs ratio_i / speed_i
        speeds = [1.0, 0.5, 2.0]
        times = [b.ratios[i] / speeds[i] for i in range(3)]
        b.update(times)
    assert abs(b.ratios[1] - 0.5) < 1e-9 and abs(b.ratios[2] - 2.0) < 1e-9
    shards = plan_shards(64 * 700, 3, ratios=b.ratios)
    assert [s.count for s in shards] == [64 * 200, 64 * 100, 64 * 400]
    with pytest.raises(ValueError):
        b.update([1.0, 0.0, 1.0])


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, layout, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, str(ROOT))
    import oracle as O

    q = O.gen_reads(11, 7, 150) if rank == 0 else None       # only rank 0 owns the queries
    s = O.gen_reads(12, 200, 150)                            # every rank can read the bucket
    s[:20] = O.mutate(O.gen_reads(11, 7, 150)[np.arange(20) % 7], np.arange(20) % 9, 3)
    seen = []

    def score_fn(queries, subjects):
        seen.append((queries.shape, subjects.shape))
        return torch.from_numpy(O.myers64(queries, subjects))

    sa = ShardedAligner(dist=dist, score_fn=score_fn)
    result, shards = sa.run(q, s, gather=True, layout=layout)
    assert seen[0][0] == (7, 150) and seen[0][1][0] == shards[rank].count  # all queries, own slice
    if rank == 0:
        np.save(out_path, result.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,layout", [(2, "row_major"), (3, "row_major"), (2, "device_blocks")])
def test_sharded_run_under_gloo(tmp_path, oracle, world, layout):
    out = tmp_path / "scores.npy"
    mp.spawn(_worker, args=(world, _free_port(), layout, str(out)), nprocs=world, join=True)
    got = np.load(out)
    q = oracle.gen_reads(11, 7, 150)
    s = oracle.gen_reads(12, 200, 150)
    s[:20] = oracle.mutate(q[np.arange(20) % 7], np.arange(20) % 9, 3)
    want = oracle.myers64(q, s)
    if layout == "row_major":
        assert np.array_equal(got, want)
    else:  # the reference's per-device block order (cal_mic.c:535-536)
        shards = plan_shards(200, world)
        blocks = np.concatenate([want[:, sh.start: sh.start + sh.count].reshape(-1) for sh in shards])
        assert np.array_equal(got, blocks)


def test_single_rank_needs_no_process_group(oracle):
    q = oracle.gen_reads(1, 3, 50)
    s = oracle.gen_reads(2, 70, 50)
    sa = ShardedAligner(dist=None, score_fn=lambda a, b: torch.from_numpy(oracle.myers64(a, b)))
    result, shards = sa.run(q, s)
    assert len(shards) == 1 and np.array_equal(result.numpy(), oracle.myers64(q, s))


# ---- the streamed per-block gather (ScoreGatherStream): block i travels while block i+1 is scored ----------
def _stream_worker(rank, world, port, layout, out_path, ns=333):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, str(ROOT))
    import oracle as O

    q = O.gen_reads(21, 23, 150) if rank == 0 else None      # 23 queries in blocks of 5: a ragged last block
    s = O.gen_reads(22, ns, 150)                             # 333 subjects: ragged last shard
    sa = ShardedAligner(dist=dist, score_fn=lambda a, b: torch.from_numpy(O.myers64(a, b)))
    blocks, shards = sa.run_streamed(q, s, block_rows=5, layout=layout)
    if rank == 0:
        np.save(out_path, np.concatenate([b.reshape(-1) for b in blocks]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,layout,ns", [(2, "device_blocks", 333), (2, "row_major", 333), (3, "device_blocks", 333),
                                             (8, "device_blocks", 1500)])     # eight ranks: the shape of the driver's SCALE run
def test_streamed_gather_under_gloo(tmp_path, oracle, world, layout, ns):
    out = tmp_path / "blocks.npy"
    mp.spawn(_stream_worker, args=(world, _free_port(), layout, str(out), ns), nprocs=world, join=True)
    got = np.load(out)
    q = oracle.gen_reads(21, 23, 150)
    s = oracle.gen_reads(22, ns, 150)
    want = oracle.myers64(q, s)
    shards = plan_shards(ns, world)
    parts = []
    for lo in range(0, 23, 5):
        blk = want[lo:lo + 5]
        if layout == "row_major":
            parts.append(blk.reshape(-1))
        else:   # the reference's result order for one block: device 0's tile, then device 1's, ... (cal_mic.c:535-536)
            parts += [blk[:, sh.start: sh.start + sh.count].reshape(-1) for sh in shards]
    assert np.array_equal(got, np.concatenate(parts))


def test_streamed_gather_single_rank_and_validation(oracle):
    from bgsa_amd.multi_gpu import ScoreGatherStream
    seen = []
    gs = ScoreGatherStream(None, "cpu", [70], torch.int16, block_rows=4, on_block=lambda i, t: seen.append((i, t.clone())))
    tiles = [torch.arange(4 * 70, dtype=torch.int16).reshape(4, 70) + b for b in range(5)]
    for t in tiles[:4]:
        gs.submit(t)
    gs.submit(tiles[4][:3])                  # ragged last block
    gs.drain()
    assert [i for i, _ in seen] == [0, 1, 2, 3, 4]
    for (i, got), t in zip(seen, tiles):
        assert torch.equal(got.reshape(-1), (t if i < 4 else t[:3]).reshape(-1))
    with pytest.raises(ValueError):
        gs.submit(torch.zeros((5, 70), dtype=torch.int16))
    with pytest.raises(ValueError):
        ScoreGatherStream(None, "cpu", [70, 70], torch.int16, block_rows=4)
