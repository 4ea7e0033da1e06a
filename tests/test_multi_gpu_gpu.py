"""The N > 1 path with the REAL compute hook on a GPU: two ranks share this box's one card (the collectives go
through gloo, since RCCL refuses two ranks on one device), every rank scores its subject shard with the HIP
kernels, the per-block tiles travel to rank 0 through ScoreGatherStream while the next block is scored, and rank 0
checks the assembled blocks against the oracle.  What an 8-GPU node runs differs only in the transport (RCCL,
device to device)."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, algo, layout, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, str(ROOT))
    import torch
    import torch.distributed as dist
    import bgsa_amd as B
    import oracle as O
    from bgsa_amd.multi_gpu import ShardedAligner
    dist.init_process_group("gloo", rank=rank, world_size=world)
    q = O.gen_reads(31, 230, 150) if rank == 0 else None      # only rank 0 owns the queries
    s = O.gen_reads(32, 64 * 37 + 11, 150)                    # ragged last shard
    s[:40] = O.mutate(O.gen_reads(31, 230, 150)[:40], np.arange(40) % 9, 33)
    sa = ShardedAligner(dist=dist, device="cuda:0", algo=algo, k=8)
    blocks, shards = sa.run_streamed(q, s, block_rows=100, layout=layout)
    torch.cuda.synchronize()
    assert B.lib().bgsa_hip_stream_faults(1) == 0
    if rank == 0:
        np.save(out_path, np.concatenate([b.reshape(-1) for b in blocks]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("algo,layout", [(0, "device_blocks"), (0, "row_major"), (1, "device_blocks"), (2, "device_blocks")])
def test_two_ranks_hip_compute_and_streamed_gather(tmp_path, oracle, algo, layout):
    import torch.multiprocessing as mp
    from bgsa_amd.multi_gpu import plan_shards
    out = tmp_path / "blocks.npy"
    mp.spawn(_worker, args=(2, _free_port(), algo, layout, str(out)), nprocs=2, join=True)
    got = np.load(out)
    q = oracle.gen_reads(31, 230, 150)
    s = oracle.gen_reads(32, 64 * 37 + 11, 150)
    s[:40] = oracle.mutate(q[:40], np.arange(40) % 9, 33)
    want = {0: lambda: oracle.myers64(q, s), 1: lambda: oracle.banded64(q, s, 8), 2: lambda: oracle.bitpal(q, s)}[algo]()
    shards = plan_shards(s.shape[0], 2)
    parts = []
    for lo in range(0, 230, 100):
        blk = want[lo:lo + 100]
        if layout == "row_major":
            parts.append(blk.reshape(-1))
        else:
            parts += [blk[:, sh.start: sh.start + sh.count].reshape(-1) for sh in shards]
    assert np.array_equal(got, np.concatenate(parts))


def _bench_two_ranks(extra_env, extra_args=(), want_rc=0, ranks=2):
    """bench.py exactly as the driver starts it for N = 2, except that both ranks share this box's one card and the
    collectives go through gloo (BGSA_BENCH_SAME_GPU / BGSA_BENCH_BACKEND: a rehearsal of the code path, not a
    measurement)."""
    import json
    import subprocess
    env = dict(os.environ, BGSA_BENCH_SAME_GPU="1", BGSA_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(ROOT / "bench.py"), "--gpus", str(ranks), "--steps", "2", "--warmup", "1",
           "--nq", "300", "--ns", "64000", *extra_args]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=str(ROOT))
    assert (p.returncode == 0) if want_rc == 0 else (p.returncode != 0), p.stdout[-2000:] + p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:] + p.stderr[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("config", [2, 5])
def test_bench_line_for_two_ranks(config):
    r = _bench_two_ranks({}, ("--config", str(config)) + (("--length", "200") if config == 5 else ()))
    assert r["n_gpus"] == 2 and r["value"] > 0
    # the job is the SAME bucket at every N, cut by plan_shards; the streamed gather is inside the timed region
    assert r["scaling"] == "strong"
    assert r["config"]["subjects_total"] == 64000 and r["config"]["subjects_this_rank"] == 32000
    assert r["config"]["gather_block_rows"] == 100 and "SIZE OVERRIDDEN" in r["config"]["workload"]     # 300 queries: the reference's block
    ko = r["kernel_only"]
    assert ko["gcups"] > 0 and ko["ms_per_pass"] > 0
    assert r["value"] <= ko["gcups"] * 1.02, (r["value"], ko)          # the gather cannot make the job faster than its kernels
    g = r["gather"]
    assert "error" not in g, g
    assert g["block_rows"] == 100 and g["blocks_per_step"] == 3 and g["bytes_to_root_per_step"] == 32000 * 300 * 2
    assert g["root_blocks_received"] == 3 * 3 + 1                      # (warm-up + two steps) x three blocks + the content check's block
    assert g["content_check"]["segments_ok"] is True and g["content_check"]["rows"] == 100
    assert g["gcups_with_gather"] == r["value"]
    assert "gather_blocks_of_100" not in r          # the timed region already ran the reference's block size
    # rank 0 at N = 1 only — and the line says so instead of omitting the keys
    assert "skipped" in r["cpu_baseline"] and "skipped" in r["total_gcups"] and "skipped" in r["other_configs"]
    assert r["rccl_ok"] is None                    # gloo rehearsal: no RCCL byte moved, the line does not claim one did
    assert r["preflight"]["all_gather"]["content_ok"] is True and len(r["ranks"][1]["peer_access"]) >= 1
    if config == 2:
        # the default invocation at N > 1 also times BASELINE configs[4]: ONE bucket of 1000 bp reads cut by plan_shards
        st = r["config5_sharded"]
        assert "error" not in st, st
        assert st["scaling"] == "strong" and st["length_bp"] == 1000 and st["subjects_total"] == 64000
        assert [x["subjects"] for x in st["ranks"]] == [32000, 32000] and all(x["kernel_ms"] > 0 for x in st["ranks"])
        assert st["gcups"] > 0 and st["gather"]["gcups_with_gather"] > 0 and st["gather"]["root_blocks_checked"] >= 3
        # ... and the weak-scaling figure of rounds 1-4 as a side key: a full bucket per rank, kernels only
        wk = r["weak"]
        assert "error" not in wk, wk
        assert wk["scaling"] == "weak" and wk["subjects_per_rank"] == 64000 and wk["subjects_total"] == 128000 and wk["gcups"] > 0
    else:
        assert "config5_sharded" not in r and "weak" not in r
    # the line proves that two ranks ran: gathered through the process group, one entry per rank with its own kernel time
    assert r["gather_ok"] is True
    assert [x["rank"] for x in r["ranks"]] == [0, 1] and r["ranks_seen"] == 2
    assert all(x["kernel_ms"] > 0 and x["device"] for x in r["ranks"]) and r["ranks"][0]["pid"] != r["ranks"][1]["pid"]
    assert r["config"]["kernel_source_id"] and len(r["config"]["kernel_source_id"]) == 16
    assert all(name in r["roofline"] for name in ("issued_frac", "issued_frac_sustained", "valu_per_wave_row"))


def test_bench_line_for_four_ranks_one_bucket_in_four_slices():
    """Four ranks on the one card (the most this box allows beside the test process): ONE 64,000-subject bucket cut by
    plan_shards into four slices, three peers streaming their tiles to rank 0 inside the timed region, several blocks per step
    (BGSA_BENCH_BLOCK_ROWS=50: six blocks of the 300 queries); the config-5 leg cuts its bucket the same way."""
    r = _bench_two_ranks({"BGSA_BENCH_BLOCK_ROWS": "50"}, ("--config", "2"), ranks=4)
    assert r["n_gpus"] == 4 and r["ranks_seen"] == 4 and [x["rank"] for x in r["ranks"]] == [0, 1, 2, 3]
    assert r["scaling"] == "strong" and r["config"]["subjects_total"] == 64000 and r["gather_ok"] is True
    assert [x["subjects"] for x in r["ranks"]] == [16000] * 4
    assert r["value"] <= r["kernel_only"]["gcups"] * 1.02
    g = r["gather"]
    assert g["block_rows"] == 50 and g["blocks_per_step"] == 6 and g["bytes_to_root_per_block"] == 3 * 16000 * 50 * 2
    assert g["root_blocks_received"] == 3 * 6 + 1 and g["content_check"]["segments_ok"] is True
    assert r["gather_blocks_of_100"]["root_blocks_received"] >= 3 and r["gather_blocks_of_100"]["bytes_to_root_per_block"] == 3 * 16000 * 100 * 2
    st = r["config5_sharded"]
    assert "error" not in st, st
    assert [x["subjects"] for x in st["ranks"]] == [16000] * 4 and st["subjects_total"] == 64000
    assert st["gather"]["bytes_to_root_per_block"] == 3 * 16000 * 100 * 2 and st["gather"]["root_blocks_checked"] >= 3
    assert r["weak"]["subjects_total"] == 4 * 64000


def test_bench_line_survives_a_gather_that_never_finishes():
    """An interconnect problem is a hang, not an exception.  The gather is inside the timed region now, so a region that does
    not finish leaves no value: the watchdog prints a line with value = null that carries the kernels-only figure measured
    before it, gather_ok = false, and every rank leaves with a NON-ZERO exit code — a hung run is not a clean run."""
    r = _bench_two_ranks({"BGSA_BENCH_GATHER_TIMEOUT": "0.001"}, ("--config", "2"), want_rc=3)
    assert r["n_gpus"] == 2 and r["value"] is None and r["scaling"] == "strong"
    assert r["kernel_only"]["gcups"] > 0
    assert "did not finish" in r["gather"]["error"] and r["gather_ok"] is False and r["rccl_ok"] is False


def test_bench_line_under_torchrun_with_one_rank_goes_through_rccl():
    """What can be run of the RCCL path on a one-GPU box: bench.py exactly as the driver starts it (torch.distributed.run, backend
    nccl = RCCL, init_process_group with device_id), world size 1 — the process group's init, the barriers, the query broadcast,
    the pre-flight all_gather, all_gather_object and the all_reduce of the timing all execute on the RCCL backend (with one
    rank they move no byte between devices: that stays the SCALE run's to show).  The line must say rccl_ok = true."""
    import json
    import subprocess
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("BGSA_BENCH_BACKEND", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(ROOT / "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--nq", "300", "--ns", "64000", "--no-cpu-baseline", "--no-total", "--no-other-configs"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=str(ROOT))
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    r = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert r["rccl_ok"] is True and r["n_gpus"] == 1 and r["value"] > 0 and "watchdog" not in r
    assert r["ranks_seen"] == 1 and r["ranks"][0]["peer_access"] and r["ranks"][0]["peer_access"][0] is True
    assert r["preflight"]["all_gather"]["content_ok"] is True


def test_bench_line_when_a_rank_never_joins():
    """The likeliest first failure on a real node is a hang in RCCL's init.  One rank never reaches init_process_group:
    the whole-run watchdog (armed before the group exists) ends the run within its limit, rank 0 prints a parseable stub
    that names the stage, rccl_ok = false, exit status non-zero."""
    import time
    t0 = time.time()
    r = _bench_two_ranks({"BGSA_BENCH_TEST_HANG_RANK": "1", "BGSA_BENCH_TIMEOUT": "25"}, ("--config", "2"), want_rc=3)
    assert time.time() - t0 < 240
    assert r["value"] is None and r["rccl_ok"] is False and r["n_gpus"] == 2
    assert r["watchdog"]["fired"] and "init_process_group" in r["watchdog"]["stage"]


def test_bench_line_carries_the_other_baseline_configs():
    """N = 1, the default invocation: configs 3, 4 and 5 are timed after the timed region and land in the same line
    (here at reduced sizes), each with its kernel, time, rate and checksum; the whole run stays under its watchdog."""
    import json
    import subprocess
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, str(ROOT / "bench.py"), "--steps", "2", "--warmup", "1", "--nq", "300", "--ns", "64000",
           "--no-cpu-baseline", "--no-total"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=str(ROOT))
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    r = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    oc = r["other_configs"]
    assert set(oc) >= {"3", "4", "5"} and "watchdog" not in r and r["rccl_ok"] is None
    for cid, length, kern in (("3", 150, "banded"), ("4", 150, "bitpal"), ("5", 1000, "myers")):
        e = oc[cid]
        assert "error" not in e, e
        assert e["length_bp"] == length and kern in e["kernel"] and e["kernel_ms"] > 0 and e["gcups"] > 0
        assert "SIZE OVERRIDDEN" in e["workload"] and isinstance(e["checksum"], int)
    assert oc["3"]["banded_mix"]["name"] == "planted" and 0 < oc["3"]["banded_mix"]["fraction"] < 0.01
    assert oc["4"]["issued"]["frac"] and oc["5"]["issued"]["frac"]          # exact generator counts at any size
    for cid in ("3", "4", "5"):     # the same scalars in every entry, mirrored into `roofline` (power / hwmon clock where readable)
        e = oc[cid]
        assert all(name in e for name in ("issued_frac", "issued_frac_sustained", "valu_per_wave_row", "sustained_mhz", "watts_mean"))
        assert r["roofline"][f"cfg{cid}_gcups"] == e["gcups"] and r["roofline"][f"cfg{cid}_issued_frac"] == e["issued_frac"]
        assert r["roofline"][f"cfg{cid}_watts_mean"] == e["watts_mean"]


def test_bench_line_carries_the_sustained_clock_without_paying_for_it():
    """N = 1: the clock probes (eight sleeping one-wave workgroups on a stream of the library's own) run beside the timed
    kernels and are released BEFORE the closing device-wide synchronize — a first version left them to their time bound
    and the line's wall time was 4.6 x the kernel time.  The line must carry a plausible clock per XCD and a wall time
    per step that is the kernel time plus launch overhead."""
    import json
    import subprocess
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, str(ROOT / "bench.py"), "--steps", "4", "--warmup", "1", "--nq", "2000", "--ns", "256000",
           "--no-cpu-baseline", "--no-total", "--no-other-configs"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=str(ROOT))
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    r = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    c = r["clock"]
    assert c is not None and len(c["per_probe_mhz"]) == 8 and sorted(c["probe_xcc"]) == list(range(8))
    assert all(1000.0 < m < 2600.0 for m in c["per_probe_mhz"]), c
    assert r["ms_per_step"] < 1.15 * r["roofline"]["kernel_ms"] + 2.0, (r["ms_per_step"], r["roofline"]["kernel_ms"])
    assert c["probe_seconds"] < 2.0 * r["ms_per_step"] * 4 / 1e3 + 0.5
    assert r["roofline"]["issued"]["frac_at_sustained_clock"] >= r["roofline"]["issued"]["frac"] * 0.99
    assert r["ranks_seen"] == 1 and r["gather_ok"] is None and r["roofline"]["traffic_model"]["query_tile"] >= 1
    # the utilisation as scalars of `roofline`, the clock and (where the driver's hwmon files are readable) the power beside them
    ro = r["roofline"]
    assert ro["issued_frac"] == ro["issued"]["frac"] and ro["issued_frac_sustained"] == ro["issued"]["frac_at_sustained_clock"]
    assert ro["valu_per_wave_row"] and ro["sustained_mhz"] == c["sustained_mhz"] and r["scaling"] == "strong"
    assert r["power"] is None or (r["power"]["samples"] >= 1 and ro["watts_mean"] == r["power"]["watts_mean"] and 50 < ro["watts_mean"] < 2000)


def test_no_clock_probes_beside_a_kernel_that_fills_the_register_file():
    """The 30 / 32-word Myers kernels hold 253-255 VGPRs at two waves per SIMD: a probe wave finds no register granule beside them and
    would displace a workgroup (config 5: 3,953 -> 4,112 ms).  The line then says why it carries no clock instead of paying for one."""
    import json
    import subprocess
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, str(ROOT / "bench.py"), "--config", "5", "--steps", "2", "--warmup", "1", "--nq", "64", "--ns", "65536",
           "--no-cpu-baseline", "--no-total"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=str(ROOT))
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    r = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert r["config"]["kernel"].startswith("myers_global_asm_kernel<32, 1>") and r["value"] > 0
    assert r["clock"]["sustained_mhz"] is None and "register file" in r["clock"]["note"]
    ro = r["roofline"]
    assert ro["issued_frac"] and ro["valu_per_wave_row"] == 256
    # without probe waves the sustained clock is the driver's own reading of the card (hwmon), where its files are readable
    if r["power"] and r["power"]["sclk_mhz_mean"]:
        assert ro["sustained_mhz"] == r["power"]["sclk_mhz_mean"] and "hwmon" in ro["issued"]["sustained_clock_source"]
        assert ro["issued_frac"] * 0.95 <= ro["issued_frac_sustained"] <= ro["issued_frac"] * 2.5
    else:
        assert ro["issued_frac_sustained"] is None


def test_config5_eight_shards_one_after_another_equal_the_whole_bucket():
    """BASELINE configs[4] (1k x 1M x 1000 bp) in its 8-GPU shape, on the one card there is: each of plan_shards(1M, 8)'s
    subject slices is scored on its own, the way rank r would (own preprocess, own launch over 125k subjects: fewer
    subject groups per query tile than the whole bucket's launch), and must equal the whole bucket's columns."""
    import torch
    import bgsa_amd as B
    from bgsa_amd.multi_gpu import plan_shards
    dev = torch.device("cuda:0")
    nq, ns, length = 1000, 1_000_000, 1000
    gen = torch.Generator(device=dev)
    gen.manual_seed(55)
    letters = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    q_rows = letters[torch.randint(0, 4, (nq, length), generator=gen, device=dev)]
    s_rows = torch.full((ns, length + 1), ord("\n"), dtype=torch.uint8, device=dev)
    s_rows[:, :length] = letters[torch.randint(0, 4, (ns, length), generator=gen, device=dev)]
    planted = torch.randperm(ns, generator=gen, device=dev)[:nq]
    s_rows[planted, :length] = q_rows
    q_host = q_rows.cpu().numpy()

    whole = B.DeviceAligner(B.ALGO_MYERS, "cuda:0", 0)
    whole.set_queries(q_host)
    whole.set_subject_rows_device(s_rows.reshape(-1), ns, length, qlen=length)
    ref = whole.score()
    torch.cuda.synchronize()
    assert bool((ref[torch.arange(nq, device=dev), planted] == 0).all())
    tile_whole = B.lib().bgsa_hip_last_query_tile()

    shards = plan_shards(ns, 8)
    assert sum(s.count for s in shards) == ns and all(s.count % 64 == 0 for s in shards)
    tiles = []
    for sh in shards:
        a = B.DeviceAligner(B.ALGO_MYERS, "cuda:0", 0)
        a.set_queries(q_host)
        a.set_subject_rows_device(s_rows[sh.start:sh.start + sh.count].reshape(-1), sh.count, length, qlen=length)
        got = a.score()
        torch.cuda.synchronize()
        tiles.append(B.lib().bgsa_hip_last_query_tile())
        assert torch.equal(got, ref[:, sh.start:sh.start + sh.count]), sh
        del a, got
    assert B.lib().bgsa_hip_stream_faults(1) == 0
    print("query tile: whole bucket", tile_whole, "shards", tiles)
