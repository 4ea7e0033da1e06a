#!/usr/bin/env python3
"""bench.py — GCUPS of the all-pairs bit-parallel alignment hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 2|3|4|5]

One "step" = one pass of the hot path over the whole workload: every query against every
resident subject of this rank (BASELINE.json configs[1] by default: Myers unit-cost global,
10k queries x 1M subjects, 150 bp).  Inputs (mapped queries, Peq blocks) are resident in HBM
before the timed region; scores stay in HBM.  For N > 1 the driver launches one rank per GPU
(torch.distributed / RCCL); subjects are sharded by rank (weak scaling: every rank owns a full
1M-subject bucket), the query set is broadcast from rank 0 once, and there is no collective in
the timed region (the reference has none on this path either — SURVEY.md §2a).

Prints ONE JSON line (rank 0).  GCUPS = query_len * n_queries * subject_len * n_subjects /
seconds / 1e9, the reference's formula (original/BGSA_CPU/cal_cpu.c:472).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import bgsa_amd as B  # noqa: E402

# MI355X_MICROARCH.md: 256 CUs x 4 SIMD-32 x 2.4 GHz -> 32-bit lane-ops/s (= 157.3 TFLOP/s fp32 / 2)
VALU_PEAK_OPS = 256 * 4 * 32 * 2.4e9
HBM_PEAK = 8.0e12

CONFIGS = {
    # id: (algo, name, nq, ns per GPU, length, k, reference ALU ops per (row, word), data bits/word of that count)
    2: (B.ALGO_MYERS, "Myers unit-cost global, 10k queries x 1M subjects, 150 bp", 10_000, 1_000_000, 150, 0),
    3: (B.ALGO_BANDED, "Banded Myers e=8, 10k x 1M, 150 bp", 10_000, 1_000_000, 150, 8),
    4: (B.ALGO_BITPAL, "BitPAl packed M=2/I=-3/G=-5, 10k x 1M, 150 bp", 10_000, 1_000_000, 150, 0),
    5: (B.ALGO_MYERS, "Myers multi-word 1000 bp, 1k x 125k subjects per GPU", 1_000, 125_000, 1000, 0),
}


def algorithmic_ops_per_cell(algo: int, length: int, k: int) -> float:
    """the reference's own ALU-op count per DP cell (32-bit lanes, 31 data bits)."""
    wn31 = (length + 30) // 31
    if algo == B.ALGO_MYERS:
        return 24.0 * wn31 / length
    if algo == B.ALGO_BITPAL:
        return 194.0 * wn31 / length
    return 42.0 / length  # banded: 42 ops per row, nominal full-matrix cells


def algorithmic_bytes_per_pair(algo: int, length: int, wn: int, q_tile: int = 100) -> float:
    """score bytes + Peq bytes amortised over a query tile of REF_BUCKET_COUNT."""
    out = 1 if algo == B.ALGO_BANDED else 2
    peq = B.group_words(algo, wn, 8) * 4 / 64
    return out + peq / q_tile


def issued_valu_per_row(algo: int, wn: int, scores=None):
    """VALU instructions the shipped row body issues per (query row, wave), from the generator's
    own instruction lists (bgsa_amd/csrc/rows_ir.py); None for the compiler-scheduled kernels."""
    sys.path.insert(0, str(ROOT / "bgsa_amd" / "csrc"))
    try:
        import rows_ir as R
    except Exception:
        return None
    if algo == B.ALGO_MYERS and wn <= 24:      # Peq planes resident (myers_global_asm_kernel)
        nw = wn if wn <= 8 else next(n for n in range(10, 25, 2) if n >= wn)
        return R.myers_body(nw).valu_count()
    if algo == B.ALGO_MYERS and wn <= 32:      # 3-bit code planes (myers_global_planes_kernel)
        nw = next(n for n in range(26, 33, 2) if n >= wn)
        return R.myers_planes_body(nw).valu_count()
    if algo == B.ALGO_BITPAL and wn <= 8:
        return R.bitpal_body(wn, R.BitpalScores(*scores) if scores else R.BITPAL_DEFAULT).valu_count()
    return None


def pmc_traffic(config: int):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/*_pmc.csv): separate
    FETCH_SIZE / WRITE_SIZE runs of this same command; FETCH_SIZE doubled per the gfx950 note in
    MI355X_MICROARCH.md §HBM (it tallies 128-B requests at 64 B)."""
    import csv
    import glob
    files = sorted(glob.glob(str(ROOT / "profiles" / f"*cfg{config}_pmc.csv")))
    if not files:
        return None
    vals = {r["counter"]: float(r["value_per_launch"]) for r in csv.DictReader(open(files[-1]))}
    if "FETCH_SIZE" not in vals or "WRITE_SIZE" not in vals:
        return None
    return {"bytes": (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024, "source": Path(files[-1]).name}


def cpu_baseline(q_rows: np.ndarray, s_rows: np.ndarray, algo: int, k: int) -> dict:
    """Time the CPU path on a bounded sample of the same workload, on this node's host cores."""
    import oracle as O

    threads = os.cpu_count() or 1
    variant = {B.ALGO_MYERS: "original_sse", B.ALGO_BITPAL: "original_avx2", B.ALGO_BANDED: "banded_cpu"}[algo]
    nq, qlen = q_rows.shape
    ns, slen = s_rows.shape
    cells = float(nq) * ns * qlen * slen
    sample = f"first {nq} queries x first {ns} subjects of the bench workload, {slen} bp"
    if O.have_reference(variant):
        try:
            t0 = time.time()
            _, out = O.run_reference(variant, q_rows, s_rows, threads=threads, k=(k if algo == B.ALGO_BANDED else None),
                                     want_scores=False, tmp_root="/dev/shm" if Path("/dev/shm").is_dir() else None)
            rep = O.parse_gcups(out)
            if rep.get("cal_seconds", 0) > 0:
                base = {"value": cells / rep["cal_seconds"] / 1e9, "unit": "GCUPS", "cores": threads, "kind": "reference",
                        "impl": f"reference {variant}/aligner -N {threads} (cal GCUPS, its own timer)",
                        "total_gcups": rep.get("total_gcups"), "sample": sample, "wall_s": round(time.time() - t0, 2)}
                if algo == B.ALGO_MYERS:
                    # The reference's AVX2 Myers kernel is generator output that is not committed
                    # upstream (no JVM here): our own 8x32 AVX2 port of align_sse, for the record.
                    _, secs = O.myers_avx2_timed(q_rows, s_rows[: ns // 8 * 8], threads=threads)
                    base["avx2_port_gcups"] = round(float(nq) * (ns // 8 * 8) * qlen * slen / secs / 1e9, 2)
                return base
        except Exception as e:  # fall through to the port
            print(f"[bench] reference baseline failed ({e}); using the oracle port", file=sys.stderr)
    if algo == B.ALGO_MYERS:
        _, secs = O.myers_avx2_timed(q_rows, s_rows, threads=threads)
        impl = "oracle/bgsa_oracle.c bgsa_oracle_myers_avx2 (8x32 AVX2, OpenMP)"
    else:
        t0 = time.time()
        (O.bitpal if algo == B.ALGO_BITPAL else (lambda a, b, threads: O.banded64(a, b, k, threads)))(q_rows, s_rows, threads=threads)
        secs = time.time() - t0
        impl = "oracle/bgsa_oracle.c scalar restatement (OpenMP)"
    return {"value": cells / secs / 1e9, "unit": "GCUPS", "cores": threads, "kind": "port", "impl": impl, "sample": sample}


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS))
    ap.add_argument("--nq", type=int, default=None, help="override query count (not the BASELINE config)")
    ap.add_argument("--ns", type=int, default=None, help="override subjects per GPU (not the BASELINE config)")
    ap.add_argument("--length", type=int, default=None, help="override read length (not the BASELINE config)")
    ap.add_argument("--scores", type=str, default=None,
                    help="match,mismatch,gap for config 4 (BitPAl); any set other than 2,-3,-5 is not a BASELINE config")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=str, default="2000x100000", help="queries x subjects timed on the CPU")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}: launch with torch.distributed.run", file=sys.stderr)
        return 2
    if not torch.cuda.is_available():
        print("[bench] no GPU visible: the HIP path has no CPU fallback", file=sys.stderr)
        return 2
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:  # under torch.distributed.run: RCCL even for one rank
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    algo, cfg_name, nq, ns, length, k = CONFIGS[args.config]
    scores = tuple(int(x) for x in args.scores.split(",")) if args.scores else None
    if scores is not None and algo != B.ALGO_BITPAL:
        print("[bench] --scores only applies to --config 4", file=sys.stderr)
        return 2
    custom_scores = scores is not None and scores != (2, -3, -5)
    overridden = args.nq is not None or args.ns is not None or args.length is not None or custom_scores
    if custom_scores:
        cfg_name = f"BitPAl packed M={scores[0]}/I={scores[1]}/G={scores[2]}, 10k x 1M, 150 bp"
    nq = args.nq or nq
    ns = args.ns or ns
    length = args.length or length
    ns_pad = (ns + 63) // 64 * 64

    # ---- synthetic workload: uniform i.i.d. A/C/G/T, generated on the GPU -------------------------
    letters = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(0xB65A0000 + args.config)
    q_rows = letters[torch.randint(0, 4, (nq, length), generator=gen, device=dev)]
    if dist is not None:
        dist.broadcast(q_rows, src=0)  # C1 of SURVEY §2a: every device sees all queries
    gen.manual_seed(0xB65A1000 + args.config + 7919 * rank)
    s_rows = torch.full((ns_pad, length + 1), ord("\n"), dtype=torch.uint8, device=dev)
    s_rows[:, :length] = ord("N")  # padding reads, as the reference pads the last bucket (file.c:98-112)
    s_rows[:ns, :length] = letters[torch.randint(0, 4, (ns, length), generator=gen, device=dev)]

    aligner = B.DeviceAligner(algo, f"cuda:{local_rank}", k, scores if algo == B.ALGO_BITPAL else None)
    aligner.set_queries(q_rows.cpu().numpy())
    aligner.set_subject_rows_device(s_rows.reshape(-1), ns_pad, length, qlen=length)
    out = torch.empty((nq, ns_pad), dtype=aligner.out_dtype, device=dev)
    torch.cuda.synchronize()

    def step():
        aligner.score(0, nq, out=out)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev0[i].record()
        step()
        ev1[i].record()
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_ms = [a.elapsed_time(b) for a, b in zip(ev0, ev1)]
    kernel_s = float(np.mean(kernel_ms)) / 1e3

    # size-independent sanity on the full-size output: diagonal-free checksum properties
    checksum = int(out[:, :ns].to(torch.int64).sum().item()) if rank == 0 else 0

    # Outside the timed region, N > 1 only: the score gather of SURVEY §2a C3 — every rank's
    # [REF_BUCKET_COUNT, subjects] tile to rank 0 over xGMI (RCCL), timed once for the record.
    gather_info = None
    if dist is not None and world > 1:
        try:
            from bgsa_amd.multi_gpu import Shard, ShardedAligner
            sa = ShardedAligner(dist=dist, device=dev, score_fn=lambda *_: None, algo=algo, k=k)
            tile = out[:100, :ns].contiguous()
            shards = [Shard(r * ns, ns) for r in range(world)]
            fence()
            g0 = time.perf_counter()
            gathered = sa.gather_scores(tile, shards, layout="row_major")
            fence()
            g1 = time.perf_counter()
            if rank == 0:
                ok = bool((gathered[:, :ns] == tile).all())
                gather_info = {"what": "100-query score tiles of all ranks to rank 0 (RCCL gather, outside the timed steps)",
                               "bytes_per_rank": int(tile.numel() * tile.element_size()), "ms": round((g1 - g0) * 1e3, 3),
                               "rank0_tile_intact": ok}
            del gathered
        except Exception as e:  # never let the optional leg break the benchmark line
            gather_info = {"error": repr(e)}

    cells_per_step_rank = float(nq) * ns * length * length
    gcups = cells_per_step_rank * world * args.steps / elapsed / 1e9
    result = None
    if rank == 0:
        wn = aligner.wn
        ops_cell = algorithmic_ops_per_cell(algo, length, k)
        kernel_gcups = cells_per_step_rank / kernel_s / 1e9
        achieved_ops = kernel_gcups * 1e9 * ops_cell
        pairs_per_s = float(nq) * ns / kernel_s
        bpp = algorithmic_bytes_per_pair(algo, length, wn)
        vpr = issued_valu_per_row(algo, wn, scores)
        issued = None
        if vpr:
            issued_ops = vpr * 64.0 * (float(nq) * (ns_pad // 64) * length) / kernel_s
            issued = {"valu_per_row": vpr, "achieved": round(issued_ops / 1e12, 2), "unit": "Tops/s",
                      "frac": round(issued_ops / VALU_PEAK_OPS, 4)}
        traffic = pmc_traffic(args.config) if not overridden else None
        result = {
            "metric": "GCUPS (cell updates/sec) all-pairs Myers 150bp" if args.config == 2 else f"GCUPS ({cfg_name})",
            "value": round(gcups, 2),
            "unit": "GCUPS",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": cfg_name + (" [SIZE OVERRIDDEN]" if overridden else ""), "queries": nq,
                       "subjects_per_gpu": ns, "length_bp": length, "k": k, "parallelism": f"subject-sharded x{world}",
                       "kernel": aligner.kernel_name(), "word_num": wn},
            # The VALU issue roofline.  `achieved` / `frac` count the instructions the shipped row body really
            # issues (from the generator's own instruction lists) when that is known — the honest utilisation
            # of the chip; `algorithmic` is the figure of SURVEY §8(d): GCUPS x the REFERENCE's ALU operations
            # per cell, which exceeds the peak because the kernels need far fewer operations than it counts.
            "roofline": {
                "bound": "valu",
                "basis": "issued" if issued else "reference_op_count",
                "achieved": issued["achieved"] if issued else round(achieved_ops / 1e12, 3),
                "peak": round(VALU_PEAK_OPS / 1e12, 2),
                "unit": "Tops/s",
                "frac": issued["frac"] if issued else round(achieved_ops / VALU_PEAK_OPS, 4),
                "traffic": traffic["bytes"] if traffic else None,
                "traffic_source": traffic["source"] if traffic else None,
                "note": "32-bit integer VALU issue bound (SURVEY §8(d)), peak = 256 CU x 4 SIMD x 32 lanes x 2.4 GHz; "
                        "kernel time from HIP events on the launch stream",
                "issued": issued,
                "algorithmic": {"ops_per_cell": round(ops_cell, 4), "achieved": round(achieved_ops / 1e12, 3),
                                "frac": round(achieved_ops / VALU_PEAK_OPS, 4),
                                "note": "GCUPS x the reference's own ALU-op count per cell (SURVEY §8(d)); > 1 = fewer "
                                        "operations than the reference needs"},
                "kernel_ms": round(kernel_s * 1e3, 3),
                "kernel_gcups": round(kernel_gcups, 1),
                "hbm": {"bound": "hbm", "achieved": round(pairs_per_s * bpp / 1e9, 2), "peak": HBM_PEAK / 1e9,
                        "unit": "GB/s", "frac": round(pairs_per_s * bpp / HBM_PEAK, 6),
                        "bytes_per_pair": round(bpp, 3)},
            },
            "checksum": checksum,
        }
        if gather_info:
            result["gather"] = gather_info
        if not args.no_cpu_baseline and world == 1 and not custom_scores:  # the reference commits 2/-3/-5 only
            cq, cs = (int(x) for x in args.cpu_sample.split("x"))
            cq, cs = min(cq, nq), min(cs, ns) // 8 * 8
            result["cpu_baseline"] = cpu_baseline(q_rows[:cq].cpu().numpy(), s_rows[:cs, :length].cpu().numpy(), algo, k)
            result["cpu_baseline"]["gpu_over_cpu"] = round(gcups / result["cpu_baseline"]["value"], 1)
            result["cpu_baseline"]["value"] = round(result["cpu_baseline"]["value"], 2)
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
