#!/usr/bin/env python3
"""bench.py — GCUPS of the all-pairs bit-parallel alignment hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 2|3|4|5]

One "step" = one pass of the hot path over the whole workload: every query against every subject of the job's bucket
(BASELINE.json configs[1] by default: Myers unit-cost global, 10k queries x 1M subjects, 150 bp).  Inputs (mapped queries, Peq
blocks) are resident in HBM before the timed region.  The job is the SAME at every N ("scaling": "strong"):

  * N = 1: the whole bucket on the one GPU, one launch per step, scores stay in HBM (nothing to gather);
  * N > 1 (one rank per GPU, torch.distributed / RCCL): the bucket is cut into contiguous slices by plan_shards (the KNC
    backend's dispatch_task, BGSA_KNC/global.c:374-431), the query set is broadcast from rank 0 once, and a step is, per block of
    gather_block_rows(nq) queries (1,000 for the 10k-query configs), the kernel on the rank's slice + its tile handed to the streamed gather (ScoreGatherStream: a side
    stream sends it to rank 0 while the next block is scored; cal_mic.c:121-147, 535-536), then drain() — the gather is INSIDE the
    timed region, so `value` is what the node delivers to rank 0, not what its kernels could.  Beside it: `kernel_only` (one launch
    per pass, no transfer: the reference's "cal"), `gather_blocks_of_100` (the same mechanism in the reference's block size),
    `weak` (rounds 1-4's headline: a full bucket per rank, kernels only — linear in N by construction) and `config5_sharded`
    (BASELINE configs[4]: 1k x 1M x 1000 bp cut the same way, kernel-only and with the streamed gather).

`value` is the reference's formula over the max-over-ranks wall time of the timed steps (cal_cpu.c:472); beside it the line
carries `total_gcups` (N = 1: raw rows on the host -> scores on the host, the whole job in reference-sized blocks: H2D + GPU
preprocess + kernel + D2H, the reference's "Total GCUPS", cal_cpu.c:473-474).

The default invocation at N = 1 also carries the other BASELINE GPU configs — `other_configs`: configs 3 (planted mix), 4 and 5, one
warm-up + two passes each after the timed region, each with the reference's CPU path for that config timed on this node's host
cores (`cpu_baseline`, 64 threads) — and the utilisation figures as scalars (`roofline.issued_frac`, `.issued_frac_sustained`,
`.valu_per_wave_row`, `.cfg<N>_*`; `cpu_baseline.cfg<N>_value`).

The whole run sits under a watchdog that is armed BEFORE init_process_group: a hang (RCCL init, a collective, a
kernel) ends with rank 0 printing what it has, `"rccl_ok": false`, and every rank leaving with status 3.

Prints ONE JSON line (rank 0).  GCUPS = query_len * n_queries * subject_len * n_subjects / seconds /
1e9, the reference's formula (original/BGSA_CPU/cal_cpu.c:472).
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import threading
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
REF_BUCKET_COUNT = 100   # queries per block of the reference's pipeline (original/BGSA_CPU/config.h:13)

CONFIGS = {
    # id: (algo, name, nq, ns, length, k, scaling).  Every config is ONE job whatever N: at N > 1 its subject bucket is cut
    # by plan_shards over the ranks (the KNC backend's dispatch_task) — total work fixed as N grows = "strong".
    2: (B.ALGO_MYERS, "Myers unit-cost global, 10k queries x 1M subjects, 150 bp", 10_000, 1_000_000, 150, 0, "strong"),
    3: (B.ALGO_BANDED, "Banded Myers e=8, 10k x 1M, 150 bp", 10_000, 1_000_000, 150, 8, "strong"),
    4: (B.ALGO_BITPAL, "BitPAl packed M=2/I=-3/G=-5, 10k x 1M, 150 bp", 10_000, 1_000_000, 150, 0, "strong"),
    5: (B.ALGO_MYERS, "Myers multi-word 1000 bp, 1k queries x 1M subjects sharded over the GPUs", 1_000, 1_000_000, 1000, 0, "strong"),
}
# Query rows per block of the sharded run's streamed gather (N > 1).  The reference's pipeline moves REF_BUCKET_COUNT = 100 query rows
# per block (a constant of its config.h, sized for host memory); a device backend's config.h sets its own, and with 288 GB of HBM
# per GPU and point-to-point xGMI links the block is sized for the transfer: 1,000 rows = 250 MB per peer and block at N = 8
# (fewer, larger transfers; a launch of 1,000 x 125k pairs holds 60 tasks per wave slot where 100 rows hold six).  The line
# carries the reference-sized blocks beside it (`gather_blocks_of_100`).  BGSA_BENCH_BLOCK_ROWS overrides.
GATHER_BLOCK_ROWS_MAX = 1000


def gather_block_rows(nq: int) -> int:
    """Rows per block of the N > 1 timed region: BGSA_BENCH_BLOCK_ROWS if set; else 1,000 — but never fewer than ten blocks per step
    (the gather of block i travels beside the kernel of block i + 1: a step of one block would overlap nothing; config 5's 1,000
    queries run as ten blocks of the reference's 100) and never below the reference's REF_BUCKET_COUNT."""
    e = os.environ.get("BGSA_BENCH_BLOCK_ROWS")
    if e:
        return max(1, int(e))
    return min(GATHER_BLOCK_ROWS_MAX, max(REF_BUCKET_COUNT, -(-nq // 10)))


# Subject mixes of the banded filter (config 3).  Its work is data dependent: a wave stops as soon as every
# one of its 64 lanes is past the error limit, so the rate depends on how many pairs survive and how they
# are spread over waves.
BANDED_MIXES = {
    "planted": "SURVEY 8(d): uniform random reads, 1 % of the subjects near-duplicates (0..k edits) of one query each",
    "random": "uniform random pairs only: every wave stops at its first test",
    "dense1pct": "1 % of ALL PAIRS survive: every query is within 2 edits of one ancestor and 1 % of the subjects, "
                 "scattered, within k-3 edits of it — 1 - 0.99^64 = 47 % of the waves hold a surviving lane",
    "survivors": "every pair survives (all reads within a few edits of one ancestor): no early exit at all, "
                 "the nominal work of the full matrix",
}


def algorithmic_ops_per_cell(algo: int, length: int, k: int) -> float:
    """the reference's own ALU-op count per DP cell (32-bit lanes, 31 data bits)."""
    wn31 = (length + 30) // 31
    if algo == B.ALGO_MYERS:
        return 24.0 * wn31 / length
    if algo == B.ALGO_BITPAL:
        return 194.0 * wn31 / length
    return 42.0 / length  # banded: 42 ops per row, nominal full-matrix cells


def algorithmic_bytes_per_pair(algo: int, length: int, wn: int, q_tile: int = REF_BUCKET_COUNT) -> float:
    """score bytes + Peq bytes amortised over a query tile of REF_BUCKET_COUNT."""
    out = 1 if algo == B.ALGO_BANDED else 2
    peq = B.group_words(algo, wn, 8) * 4 / 64
    return out + peq / q_tile


def issued_valu_per_row(algo: int, wn: int, k: int = 0, scores=None):
    """VALU instructions the shipped row body issues per (query row, wave), from the generator's
    own instruction lists (bgsa_amd/csrc/rows_ir.py); None for the compiler-scheduled kernels."""
    sys.path.insert(0, str(ROOT / "bgsa_amd" / "csrc"))
    try:
        import rows_ir as R
    except Exception:
        return None
    peq_max = int(os.environ.get("BGSA_MYERS_PEQ_MAX_WORDS", "32"))     # myers_global.hip: myers_peq_max_words()
    if algo == B.ALGO_MYERS and wn <= min(peq_max, 32):      # Peq planes resident (myers_global_asm_kernel): 8 VALU per word at every width
        nw = wn if wn <= 8 or wn == 25 else next(n for n in list(range(10, 25, 2)) + [26, 28, 30, 32] if n >= wn)
        return R.myers_body(nw).valu_count()
    if algo == B.ALGO_MYERS and wn <= 32:      # 3-bit code planes (myers_global_planes_kernel)
        nw = next(n for n in range(26, 33, 2) if n >= wn)
        return R.myers_planes_body(nw).valu_count()
    if algo == B.ALGO_BITPAL and wn <= 8:
        return R.bitpal_body(wn, R.BitpalScores(*scores) if scores else R.BITPAL_DEFAULT).valu_count()
    if algo == B.ALGO_BANDED:                  # per row that is actually run (early exit: fewer rows than nominal)
        impl = os.environ.get("BGSA_BANDED_IMPL", "")[:1]
        if k <= 15 and R.banded_phase_rows(k) and impl == "p":
            return R.banded_phase_body().valu_count()     # the band held in place (A/B alternative, k <= 11)
        if impl == "" and R.banded_cut_rows(k):
            return R.banded_cut_body(1).valu_count()      # one-word windows (per group and row, whatever the groups per wave)
        if k <= 15:
            return R.banded_body().valu_count()
        one_shift = impl == "" and os.environ.get("BGSA_BANDED_PAIR_LOOP", "1") == "1"    # the default loop's pair row (round 4)
        return (R.banded_body64_sh64() if one_shift else R.banded_body64()).valu_count()
    return None


ISSUED_SCALARS = ("issued_frac", "issued_frac_sustained", "valu_per_wave_row")


def flat_issued(issued: dict | None) -> dict:
    """The utilisation figures of an `issued` object as scalars, under the names every entry of the line uses (the headline's
    `roofline` and each of `other_configs`): a reader — or a parser — that keeps only the scalars of an object still has the
    figure that is <= 1.  issued_frac = VALU instructions issued x 64 lanes / kernel time / peak at the nominal 2.4 GHz,
    issued_frac_sustained = the same against the clock the chip held, valu_per_wave_row = instructions per (query row, wave)."""
    if not issued:
        return {name: None for name in ISSUED_SCALARS} | {"issued_source": None}
    return {"issued_frac": issued.get("frac"), "issued_frac_sustained": issued.get("frac_at_sustained_clock"),
            "valu_per_wave_row": issued.get("valu_per_nominal_wave_row", issued.get("valu_per_row")),
            "issued_source": issued.get("source")}


KERNEL_SOURCES = {   # what defines the scoring kernel of an algorithm: its id stamps PMC passes and the bench line
    B.ALGO_MYERS: ("myers_global.hip", "myers_rows_gen.inc", "bgsa_common.h"),
    B.ALGO_BANDED: ("banded.hip", "banded_rows_gen.inc", "bgsa_common.h"),
    B.ALGO_BITPAL: ("bitpal.hip", "bitpal_kernels.inl", "bitpal_rows_gen.inc", "bgsa_common.h"),
}


def kernel_source_id(algo: int) -> str:
    """sha256 (16 hex digits) over the sources that define this algorithm's scoring kernels, in the tree the library was
    built from.  A PMC pass is only comparable with a run of the SAME kernel: collect_profiles.py stamps every
    profiles/*_pmc.csv with the id the profiled bench.py printed, pmc_values() refuses a file whose id differs."""
    import hashlib
    h = hashlib.sha256()
    for name in KERNEL_SOURCES[algo]:
        h.update(name.encode() + b"\0" + (ROOT / "bgsa_amd" / "csrc" / name).read_bytes())
    return h.hexdigest()[:16]


def pmc_values(config: int, tag: str = "", source_id: str | None = None):
    """Per-launch counter values of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/*cfg<N><tag>_pmc.csv: separate --pmc runs of this same command, scripts/collect_profiles.py).
    Returns (values, file name), or (None, reason) when there is no pass or the newest one was collected from another
    kernel (its `_meta,kernel_source_id` row differs from source_id, or is missing)."""
    import csv
    import glob
    files = sorted(glob.glob(str(ROOT / "profiles" / f"*cfg{config}{tag}_pmc.csv")))
    if not files:
        return None, None
    rows = list(csv.DictReader(open(files[-1])))
    meta = {r["counter"]: r["value_per_launch"] for r in rows if r.get("pass") == "_meta"}
    if source_id is not None and meta.get("kernel_source_id") != source_id:
        return None, (f"{Path(files[-1]).name} refused: collected from kernel source id {meta.get('kernel_source_id', 'unstamped')}, "
                      f"this build is {source_id} — re-collect the PMC passes")
    vals = {r["counter"]: float(r["value_per_launch"]) for r in rows if r.get("pass") != "_meta"}
    return vals, Path(files[-1]).name


def cpu_baseline(q_rows: np.ndarray, s_rows: np.ndarray, algo: int, k: int, thread_counts=None, budget_s: float | None = None) -> dict:
    """Time the CPU path on a bounded sample of the same workload, on this node's host cores.

    Myers: north_star asks for "BGSA's own AVX2 CPU path".  Upstream commits the Myers kernel as its SSE instance only
    (the AVX2 one is generator output, and there is no JVM here); oracle/_ref/original_avx2_myers is the generator's AVX2
    instance of that file — the token mapping of AVX2Arch / AVX2Intrinsics applied at build time, compiled with
    original/BGSA_AVX2's own host files (oracle/derive_avx2_myers.py), score-identical to the scalar
    oracle on the golden fixtures (tests/test_oracle.py).  It is the reported `value` (kind "reference-derived avx2"),
    the committed SSE build's figure stands beside it."""
    import oracle as O

    threads = os.cpu_count() or 1
    variants = {B.ALGO_MYERS: ["original_avx2_myers", "original_sse"], B.ALGO_BITPAL: ["original_avx2"],
                B.ALGO_BANDED: ["banded_cpu"]}[algo]
    kinds = {"original_avx2_myers": "reference-derived avx2", "original_sse": "reference", "original_avx2": "reference",
             "banded_cpu": "reference"}
    nq, qlen = q_rows.shape
    ns, slen = s_rows.shape
    cells = float(nq) * ns * qlen * slen
    sample = f"first {nq} queries x first {ns} subjects of the bench workload, {slen} bp"
    budget = float(os.environ.get("BGSA_BENCH_CPU_SECONDS", "45")) if budget_s is None else budget_s
    t0 = time.time()
    found = []
    for variant in variants:
        if not O.have_reference(variant):
            continue
        try:
            # the reference's guided OpenMP grid need not peak with every hardware thread busy: the best of three
            # thread counts on the same sample, each by the reference's own cal timer (a quarter of the hardware
            # threads won on every box so far: it goes first, so that a tight budget still sees it)
            tried, best = {}, None
            for n_thr in (thread_counts or sorted({max(1, threads // 4), max(1, threads // 2), threads})):
                _, out = O.run_reference(variant, q_rows, s_rows, threads=n_thr, k=(k if algo == B.ALGO_BANDED else None),
                                         want_scores=False, tmp_root="/dev/shm" if Path("/dev/shm").is_dir() else None)
                r = O.parse_gcups(out)
                if r.get("cal_seconds", 0) > 0:
                    tried[n_thr] = round(cells / r["cal_seconds"] / 1e9, 2)
                    if best is None or r["cal_seconds"] < best[1]["cal_seconds"]:
                        best = (n_thr, r)
                if time.time() - t0 > budget * (len(found) + 1) / len(variants):     # keep the default run bounded
                    break
            if best is not None:
                n_thr, rep = best
                found.append({"value": cells / rep["cal_seconds"] / 1e9, "unit": "GCUPS", "cores": n_thr, "kind": kinds[variant],
                              "impl": f"oracle/_ref/{variant}/aligner -N {n_thr} (cal GCUPS, its own timer; best of the thread counts tried)",
                              "gcups_by_threads": tried, "total_gcups": rep.get("total_gcups"), "sample": sample})
        except Exception as e:
            print(f"[bench] reference baseline {variant} failed ({e})", file=sys.stderr)
    if found:
        base = max(found, key=lambda b: b["value"]) if algo != B.ALGO_MYERS else found[0]
        if algo == B.ALGO_MYERS:
            sse = next((b for b in found if "original_sse" in b["impl"]), None)
            if sse is not None and sse is not base:
                base["sse"] = {"value": round(sse["value"], 2), "cores": sse["cores"], "kind": "reference", "impl": sse["impl"],
                               "gcups_by_threads": sse["gcups_by_threads"]}
                base["avx2_over_sse"] = round(base["value"] / sse["value"], 2)
                if base["value"] < sse["value"]:
                    base["note"] = ("the AVX2 instance is slower than the SSE build on this host (published ratio on a Xeon W-2123: "
                                    "about 1.2x, README.md:84-92): its eight-lane groups leave the guided OpenMP grid fewer, larger "
                                    "chunks per thread on this sample")
            base["derivation"] = ("original/BGSA_SSE/align_core.c:19-152 with _mm_ -> _mm256_, si128 -> si256, SSE_ -> AVX_, "
                                  "sse_ -> avx_ (AVX2Arch.java:23-60, MyersGenerator.java:225-401), built with original/BGSA_AVX2's "
                                  "host files at build time; score-identical to original/BGSA_CPU on fixtures F1/F2/F4/F5/F9") \
                if base["kind"] != "reference" else None
        base["wall_s"] = round(time.time() - t0, 2)
        return base
    if algo == B.ALGO_MYERS:
        _, secs = O.myers_avx2_timed(q_rows, s_rows, threads=threads)
        impl = "oracle/bgsa_oracle.c bgsa_oracle_myers_avx2 (8x32 AVX2, OpenMP)"
    else:
        t0 = time.time()
        (O.bitpal if algo == B.ALGO_BITPAL else (lambda a, b, threads: O.banded64(a, b, k, threads)))(q_rows, s_rows, threads=threads)
        secs = time.time() - t0
        impl = "oracle/bgsa_oracle.c scalar restatement (OpenMP)"
    return {"value": cells / secs / 1e9, "unit": "GCUPS", "cores": threads, "kind": "port", "impl": impl, "sample": sample}


# ---- synthetic reads ---------------------------------------------------------------------------------

def random_reads(n, length, gen, dev):
    letters = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    return letters[torch.randint(0, 4, (n, length), generator=gen, device=dev)]


def mutate_reads(rows, max_edits, gen, dev):
    """Up to max_edits substitutions per read (uniform count 0..max_edits), on the GPU.  Substitutions
    keep the length — what the banded kernel needs (equal lengths only) — and bound the edit distance."""
    n, length = rows.shape
    out = rows.clone()
    letters = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    n_edits = torch.randint(0, max_edits + 1, (n,), generator=gen, device=dev)
    for e in range(max_edits):
        pos = torch.randint(0, length, (n,), generator=gen, device=dev)
        new = letters[torch.randint(0, 4, (n,), generator=gen, device=dev)]
        idx = torch.nonzero(n_edits > e).squeeze(1)
        out[idx, pos[idx]] = new[idx]
    return out


def make_workload(config, algo, nq, ns, length, k, mix, rank, dev, dist):
    """Queries [nq, length] and subject rows [ns_pad, length+1] (uint8 device tensors)."""
    ns_pad = (ns + 63) // 64 * 64
    gen = torch.Generator(device=dev)
    gen.manual_seed(0xB65A0000 + config)
    q_rows = random_reads(nq, length, gen, dev)
    ancestor = None
    if algo == B.ALGO_BANDED and mix in ("dense1pct", "survivors"):
        ancestor = random_reads(1, length, gen, dev)
        q_rows = mutate_reads(ancestor.expand(nq, length), 2, gen, dev)
    if dist is not None:
        dist.broadcast(q_rows, src=0)  # C1 of SURVEY §2a: every device sees all queries
        if ancestor is not None:
            dist.broadcast(ancestor, src=0)
    gen.manual_seed(0xB65A1000 + config + 7919 * rank)
    s_rows = torch.full((ns_pad, length + 1), ord("\n"), dtype=torch.uint8, device=dev)
    s_rows[:, :length] = ord("N")  # padding reads, as the reference pads the last bucket (file.c:98-112)
    s_rows[:ns, :length] = random_reads(ns, length, gen, dev)
    if algo == B.ALGO_BANDED and mix != "random":
        if mix == "survivors":
            s_rows[:ns, :length] = mutate_reads(ancestor.expand(ns, length), max(k - 3, 1), gen, dev)
        else:
            n_planted = ns // 100
            where = torch.randperm(ns, generator=gen, device=dev)[:n_planted]
            if mix == "planted":   # near-duplicates of individual queries
                src = q_rows[torch.randint(0, nq, (n_planted,), generator=gen, device=dev)]
                s_rows[where, :length] = mutate_reads(src, k, gen, dev)
            else:                  # dense1pct: near the common ancestor of all queries
                s_rows[where, :length] = mutate_reads(ancestor.expand(n_planted, length), max(k - 3, 1), gen, dev)
    return q_rows, s_rows, ns_pad


# ---- Total GCUPS: host rows -> host scores for reference-sized blocks ---------------------------------

def total_gcups_leg(algo, k, scores, q_host, s_rows_dev, ns, ns_pad, length, dev, n_blocks=1 << 30):
    """Raw subject rows in pinned host memory -> H2D -> GPU preprocess -> per block of REF_BUCKET_COUNT
    queries: kernel on one stream, D2H of the previous block's scores on another (two result buffers) ->
    scores in pinned host memory (two buffers, overwritten as a writer thread would drain them).  The whole
    query set by default.  Wall time of all of it = the reference's Total GCUPS (cal_cpu.c:473-474) without the
    file I/O (the reference keeps that outside too: its I/O threads)."""
    L = B.lib()
    nq = min(q_host.shape[0], n_blocks * REF_BUCKET_COUNT)
    n_blocks = (nq + REF_BUCKET_COUNT - 1) // REF_BUCKET_COUNT
    esz = 1 if algo == B.ALGO_BANDED else 2
    h_rows = torch.empty(s_rows_dev.numel(), dtype=torch.uint8).pin_memory()
    h_rows.copy_(s_rows_dev.reshape(-1))
    h_out = [torch.empty((REF_BUCKET_COUNT, ns_pad), dtype=torch.int8 if esz == 1 else torch.int16).pin_memory() for _ in range(2)]
    a = B.DeviceAligner(algo, str(dev), k, scores)
    d_rows = torch.empty_like(s_rows_dev.reshape(-1))
    d_out = [torch.empty((REF_BUCKET_COUNT, ns_pad), dtype=a.out_dtype, device=dev) for _ in range(2)]
    copy_stream = torch.cuda.Stream(device=dev)
    done = [torch.cuda.Event() for _ in range(2)]
    copied = [torch.cuda.Event() for _ in range(2)]
    a.set_queries(q_host[:nq])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    ev[0].record()
    d_rows.copy_(h_rows, non_blocking=True)
    a.set_subject_rows_device(d_rows, ns_pad, length, qlen=length)
    ev[1].record()
    for b in range(n_blocks):
        slot = b & 1
        lo, hi = b * REF_BUCKET_COUNT, min(nq, (b + 1) * REF_BUCKET_COUNT)
        if b >= 2:
            torch.cuda.current_stream(dev).wait_event(copied[slot])   # the buffer's previous copy-out is done
        a.score(lo, hi, out=d_out[slot][: hi - lo])
        done[slot].record()
        with torch.cuda.stream(copy_stream):
            copy_stream.wait_event(done[slot])
            h_out[slot][: hi - lo].copy_(d_out[slot][: hi - lo], non_blocking=True)
            copied[slot].record()
    issued = time.perf_counter() - t0
    with torch.cuda.stream(copy_stream):
        ev[2].record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    a.check_faults()
    cells = float(nq) * ns * length * length
    return {"value": round(cells / wall / 1e9, 1), "unit": "GCUPS", "wall_ms": round(wall * 1e3, 2),
            "stages_ms": {"h2d_and_preprocess": round(ev[0].elapsed_time(ev[1]), 2), "blocks": round(ev[1].elapsed_time(ev[2]), 2),
                          "host_issue": round(issued * 1e3, 2)},
            "what": f"{n_blocks} blocks of {REF_BUCKET_COUNT} queries x {ns_pad} subjects: pinned host rows -> H2D -> GPU "
                    f"preprocess -> kernel -> D2H (double-buffered on a second stream) -> pinned host scores; "
                    f"formula of cal_cpu.c:473-474 without the file I/O",
            "h2d_bytes": int(h_rows.numel()), "d2h_bytes": int(nq * ns_pad * esz)}


class PowerSampler:
    """The GPU's power and shader clock as the driver's hwmon files report them (/sys/class/drm/card*/device/hwmon/hwmon*/
    power1_average | power1_input in microwatts, freq1_input in Hz), read on the host about five times a second while the timed
    region runs: plain file reads from a thread — nothing is launched on the GPU and NO program is started (rocm-smi is a script
    whose interpreter would have to be exec'ed from a process that has initialised the GPU, which the GPU boxes refuse, rightly).
    The headline kernel is power-bound (LABNOTES §9.5): the clock the chip holds under it is the box-to-box spread of the headline
    number, and this puts the watts beside it.  Best effort: where the files are absent or unreadable the line carries `null`."""

    def __init__(self, period_s: float = 0.2):
        import glob
        self.period, self.samples, self._stop, self._thread = period_s, [], threading.Event(), None
        self.dirs = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))

    @staticmethod
    def _read(path):
        try:
            with open(path) as f:
                return float(f.read().strip())
        except Exception:
            return None

    def _once(self):
        cards = []
        for d in self.dirs:
            w = self._read(d + "/power1_average")
            if w is None:
                w = self._read(d + "/power1_input")
            f = self._read(d + "/freq1_input")
            cards.append((w / 1e6 if w is not None else None, f / 1e6 if f is not None else None))
        return cards if any(w is not None for w, _ in cards) else None

    def start(self):
        if not self.dirs:
            return self

        def loop():
            while not self._stop.is_set():
                one = self._once()
                if one:
                    self.samples.append(one)
                self._stop.wait(self.period)
        self._thread = threading.Thread(target=loop, daemon=True)
        self._thread.start()
        return self

    def stop(self):
        self._stop.set()
        if self._thread is not None:
            self._thread.join(timeout=2)
        if not self.samples:
            return None
        # a box may list idle neighbours (eight cards on the host, one of them ours): the card under the kernel is the one that
        # drew the most power over the region; its clock file reads the shader clock of that card
        n = len(self.samples[0])
        mean_w = [float(np.mean([s_[i][0] for s_ in self.samples if s_[i][0] is not None] or [0.0])) for i in range(n)]
        busy = int(np.argmax(mean_w))
        watts = [s_[busy][0] for s_ in self.samples if s_[busy][0] is not None]
        sclk = [s_[busy][1] for s_ in self.samples if s_[busy][1] is not None]
        return {"samples": len(self.samples), "cards_seen": n, "card": self.dirs[busy].split("/")[4],
                "watts_mean": round(float(np.mean(watts)), 1) if watts else None,
                "watts_max": round(max(watts), 1) if watts else None,
                "sclk_mhz_mean": round(float(np.mean(sclk)), 1) if sclk else None,
                "source": "hwmon power1_average / freq1_input of the busiest /sys/class/drm/card*, read on the host during the timed region"}


class RunWatchdog:
    """The whole run under one time limit, armed before the process group exists.

    An interconnect or GPU problem shows as a hang, not an exception: RCCL's init, the query broadcast, a barrier
    or a kernel that never returns.  When the limit passes, rank 0 prints ONE parseable line — the measured line
    if the timed region is already behind it, a stub that names the stage otherwise — with `rccl_ok: false`, and
    every rank leaves with exit status 3 (`os._exit`: a process that has touched the GPU is ended, never
    replaced).  `leg()` arms a second, shorter limit around an optional leg; its expiry keeps the measured value."""

    def __init__(self, rank: int, world: int, args, limit: float | None = None):
        self.rank, self.world, self.args = rank, world, args
        self.limit = float(os.environ.get("BGSA_BENCH_TIMEOUT", "900")) if limit is None else limit
        self.stage = "start"
        self.t0 = time.time()
        self.lock = threading.Lock()        # the one JSON line is printed once, by main() or by a timer
        self.printed = False
        self.result = None                  # main() publishes its line here as soon as the timed region is done
        self.extra = {}                     # what a firing timer adds to the line
        self._timers = []
        self._arm(self.limit, f"whole run did not finish within {self.limit:.0f} s")

    def _arm(self, seconds, why, patch=None):
        t = threading.Timer(seconds, self._fire, args=(why, patch))
        t.daemon = True
        t.start()
        self._timers.append(t)
        return t

    def leg(self, seconds, why, patch):
        """A limit for one optional leg; patch(line) marks the leg as timed out in the printed line."""
        return self._arm(seconds, why, patch)

    def stub(self):
        a = self.args
        return {"metric": "GCUPS (cell updates/sec) all-pairs Myers 150bp" if a.config == 2 else f"GCUPS (config {a.config})",
                "value": None, "unit": "GCUPS", "n_gpus": self.world, "steps": a.steps, "warmup": a.warmup,
                "ms_per_step": None, "higher_is_better": True, "scaling": CONFIGS[a.config][6], "vs_baseline": None,
                "dtype": "u32", "data": "synthetic", "config": {"workload": CONFIGS[a.config][1]}}

    def _fire(self, why, patch):
        if self.rank != 0:
            time.sleep(2.0)         # let rank 0 get its line out before peers start to disappear under it
        with self.lock:
            if self.rank == 0 and not self.printed:
                line = dict(self.result) if self.result is not None else self.stub()
                line.update(self.extra)
                line["rccl_ok"] = False
                line["watchdog"] = {"fired": True, "why": why, "stage": self.stage,
                                    "after_s": round(time.time() - self.t0, 1)}
                if patch is not None:
                    patch(line)
                print(json.dumps(line), flush=True)
                self.printed = True
            os._exit(3)             # also ends a rank that printed its line and then hung in the closing barrier

    def cancel(self):
        for t in self._timers:
            t.cancel()


def checksum_int64(out, ns, rows_per_block: int = 256) -> int:
    """Sum of all scores as int64 without an int64 copy of the matrix (10^10 scores would be an 80 GB temporary)."""
    total = 0
    for lo in range(0, out.shape[0], rows_per_block):
        total += int(out[lo:lo + rows_per_block, :ns].sum(dtype=torch.int64).item())
    return total


def preflight(dist, dev, rank, world, local_rank):
    """Cheap evidence about the node before anything is timed: which peers this rank's device can address
    directly (hipDeviceCanAccessPeer), and one all_gather of BGSA_BENCH_PREFLIGHT_MB (64) per rank."""
    info = {}
    try:
        n_dev = torch.cuda.device_count()
        info["peer_access"] = [bool(j == dev.index or torch.cuda.can_device_access_peer(dev.index, j)) for j in range(n_dev)]
    except Exception as e:
        info["peer_access_error"] = repr(e)
    if dist is not None:      # also with one rank under torchrun: the collective still runs on the RCCL backend
        mb = float(os.environ.get("BGSA_BENCH_PREFLIGHT_MB", "64"))
        n = max(1, int(mb * (1 << 20)))
        try:
            src = torch.full((n,), rank & 0xFF, dtype=torch.uint8, device=dev)
            dst = [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(world)]
            dist.all_gather(dst, src)            # first one also pays the transport's connection set-up
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            dist.all_gather(dst, src)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            ok = all(int(dst[r][0].item()) == (r & 0xFF) and int(dst[r][-1].item()) == (r & 0xFF) for r in range(world))
            info["all_gather"] = {"mb_per_rank": mb, "ms": round(dt * 1e3, 3), "content_ok": ok,
                                  "recv_gbps": round((world - 1) * n / dt / 1e9, 2)}
        except Exception as e:
            info["all_gather"] = {"error": repr(e)}
    return info


# CPU samples of the other configs' reference baselines (queries x subjects of the config's own workload), sized for <= 10 s each at
# 64 threads on the GPU box's host: banded ~2,100 GCUPS, BitPAl ~165, Myers 1000 bp ~1,900 (AVX2 instance) / ~950 (SSE)
OTHER_CPU_SAMPLES = {3: (2000, 200_000), 4: (400, 100_000), 5: (100, 50_000)}


def side_config(cfg_id, dev, L, nq=None, ns=None, passes=2, mix="planted", rank=0, dist=None, keep=False, cpu=False):
    """One more BASELINE config inside the same line: workload resident in HBM, one warm-up pass, `passes` passes
    timed with HIP events on the launch stream.  Returns the entry (and, with keep, the live objects for a
    further leg)."""
    algo, name, cfg_nq, cfg_ns, length, k, _ = CONFIGS[cfg_id]
    nq, ns = nq or cfg_nq, ns or cfg_ns
    mix = mix if algo == B.ALGO_BANDED else None
    q_rows, s_rows, ns_pad = make_workload(cfg_id, algo, nq, ns, length, k, mix, rank, dev, dist)
    a = B.DeviceAligner(algo, str(dev), k, None)
    a.set_queries(q_rows.cpu().numpy())
    a.set_subject_rows_device(s_rows.reshape(-1), ns_pad, length, qlen=length)
    out = torch.empty((nq, ns_pad), dtype=a.out_dtype, device=dev)
    a.score(0, nq, out=out)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(passes)]
    # power and clock of the card while the passes run (hwmon files, no probe waves here): one GPU, rank 0 only
    sampler = PowerSampler(0.05).start() if (dist is None and rank == 0 and os.environ.get("BGSA_BENCH_POWER", "1") != "0") else None
    for e0, e1 in ev:
        e0.record()
        a.score(0, nq, out=out)
        e1.record()
    torch.cuda.synchronize()
    side_power = sampler.stop() if sampler else None
    a.check_faults()
    kernel_s = float(np.mean([e0.elapsed_time(e1) for e0, e1 in ev])) / 1e3
    cells = float(nq) * ns * length * length
    entry = {"workload": name + (" [SIZE OVERRIDDEN]" if (nq, ns) != (cfg_nq, cfg_ns) else ""),
             "queries": nq, "subjects": ns, "length_bp": length, "k": k,
             "kernel": a.kernel_name(), "kernel_ms": round(kernel_s * 1e3, 3), "passes": passes,
             "gcups": round(cells / kernel_s / 1e9, 1),
             "checksum": checksum_int64(out, ns)}
    if algo == B.ALGO_BANDED:
        alive = int((out[:, :ns] != 127).sum().item())
        entry["banded_mix"] = {"name": mix, "pairs_not_rejected": alive, "fraction": alive / (float(nq) * ns)}
    src_id = kernel_source_id(algo)
    entry["kernel_source_id"] = src_id
    sized = (nq, ns) == (cfg_nq, cfg_ns)
    pmc, pmc_src = pmc_values(cfg_id, f"_{mix}" if mix else "", src_id) if sized else (None, None)
    vpr = issued_valu_per_row(algo, a.wn, k, None)
    exact = not (algo == B.ALGO_BANDED and mix != "survivors")
    if pmc and "SQ_INSTS_VALU" in pmc:
        entry["issued"] = {"frac": round(pmc["SQ_INSTS_VALU"] * 64.0 / kernel_s / VALU_PEAK_OPS, 4),
                           "source": f"SQ_INSTS_VALU, {pmc_src}"}
    elif vpr and exact:
        entry["issued"] = {"frac": round(vpr * 64.0 * float(nq) * (ns_pad // 64) * length / kernel_s / VALU_PEAK_OPS, 4),
                           "source": "generator instruction lists (rows_ir.py) x rows x waves", "pmc_file": pmc_src}
    else:
        entry["issued"] = {"frac": None, "source": pmc_src or "no PMC pass for this config and mix"}
    entry["roofline_frac_reference_ops"] = round(entry["gcups"] * 1e9 * algorithmic_ops_per_cell(algo, length, k) / VALU_PEAK_OPS, 4)
    # the same three scalars as the headline's `roofline` object
    if pmc and "SQ_INSTS_VALU" in pmc:
        entry["issued"]["valu_per_nominal_wave_row"] = round(pmc["SQ_INSTS_VALU"] / (float(nq) * (ns_pad // 64) * length), 3)
    elif vpr and exact:
        entry["issued"]["valu_per_row"] = vpr
    entry["power"] = side_power
    if side_power and side_power.get("sclk_mhz_mean") and entry["issued"].get("frac"):
        # the clock probes run beside the headline's timed region only; here the sustained clock is the driver's own reading of the
        # card's shader clock, sampled on the host during the passes (about 1.5 % above what the probe waves measure under config 2)
        entry["issued"]["frac_at_sustained_clock"] = round(entry["issued"]["frac"] * 2400.0 / side_power["sclk_mhz_mean"], 4)
        entry["issued"]["sustained_clock_source"] = "hwmon freq1_input, mean over the timed passes"
    entry.update(flat_issued(entry["issued"]))
    entry["sustained_mhz"] = side_power.get("sclk_mhz_mean") if side_power else None
    entry["watts_mean"] = side_power.get("watts_mean") if side_power else None
    if cpu:
        # the reference's own CPU path for this config on this node's host cores, in the same run (north_star): 64 threads —
        # the count that won every thread sweep of rounds 3-4 on these hosts — by the reference's cal timer (cal_cpu.c:111-118,472)
        try:
            cq, cs = OTHER_CPU_SAMPLES[cfg_id]
            cq, cs = min(cq, nq), min(cs, ns) // 8 * 8
            threads = min(64, os.cpu_count() or 1)
            base = cpu_baseline(q_rows[:cq].cpu().numpy(), s_rows[:cs, :length].cpu().numpy(), algo, k, thread_counts=[threads], budget_s=12.0)
            base["gpu_over_cpu"] = round(entry["gcups"] / base["value"], 1)
            base["value"] = round(base["value"], 2)
            entry["cpu_baseline"] = base
        except Exception as e:
            entry["cpu_baseline"] = {"error": repr(e)}
    if keep:
        return entry, (a, out, q_rows, s_rows, ns_pad)
    del a, out, q_rows, s_rows
    torch.cuda.empty_cache()
    return entry


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS))
    ap.add_argument("--nq", type=int, default=None, help="override query count (not the BASELINE config)")
    ap.add_argument("--ns", type=int, default=None, help="override subject count (not the BASELINE config)")
    ap.add_argument("--length", type=int, default=None, help="override read length (not the BASELINE config)")
    ap.add_argument("--k", type=int, default=None, help="override the banded threshold (not the BASELINE config)")
    ap.add_argument("--scores", type=str, default=None,
                    help="match,mismatch,gap for config 4 (BitPAl); any set other than 2,-3,-5 is not a BASELINE config")
    ap.add_argument("--banded-mix", type=str, default="planted", choices=sorted(BANDED_MIXES),
                    help="subject mix of config 3 (default: SURVEY 8(d)'s planted stratum)")
    ap.add_argument("--banded-variants", type=str, default="random,dense1pct,survivors",
                    help="config 3, one GPU: other mixes timed after the main one ('' = none)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-total", action="store_true", help="skip the Total-GCUPS leg")
    ap.add_argument("--cpu-sample", type=str, default="1000x100000", help="queries x subjects timed on the CPU (once per thread count tried)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="N = 1, config 2: do not time configs 3 / 4 / 5 after the timed region (set it under a profiler)")
    ap.add_argument("--no-strong", action="store_true",
                    help="N > 1, config 2: do not run the config-5 leg (its 1000 bp bucket cut by plan_shards: `config5_sharded`)")
    ap.add_argument("--no-weak", action="store_true",
                    help="N > 1, config 2: do not run the weak-scaling side leg (a full bucket per rank, kernels only: `weak`)")
    ap.add_argument("--no-clock-probe", action="store_true",
                    help="do not run the sustained-clock probe waves beside the timed kernels (set it under rocprofv3 --pmc, "
                         "which may serialise kernels: the probes then delay the launch they are meant to observe)")
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
    # Armed before anything can hang: RCCL's init is the likeliest first failure on a real node.
    wd = RunWatchdog(rank, world, args)
    if os.environ.get("BGSA_BENCH_TEST_HANG_RANK") == str(rank):      # test hook: this rank never joins the group
        wd.stage = "test hook: this rank sleeps before init_process_group"
        time.sleep(1e6)
    # Rehearsal of the N > 1 code on a one-GPU box (not a measurement): BGSA_BENCH_SAME_GPU=1 puts every rank on
    # device 0 and BGSA_BENCH_BACKEND=gloo replaces RCCL, which refuses two ranks on one device.
    if os.environ.get("BGSA_BENCH_SAME_GPU") == "1":
        local_rank = 0
    backend = os.environ.get("BGSA_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:  # under torch.distributed.run: RCCL even for one rank
        import torch.distributed as dist
        wd.stage = f"init_process_group({backend})"
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        wd.stage = "first barrier"
        dist.barrier()
    wd.stage = "preflight"
    pre = preflight(dist, dev, rank, world, local_rank)

    algo, cfg_name, nq, ns_total, length, k, scaling = CONFIGS[args.config]
    scores = tuple(int(x) for x in args.scores.split(",")) if args.scores else None
    if scores is not None and algo != B.ALGO_BITPAL:
        print("[bench] --scores only applies to --config 4", file=sys.stderr)
        return 2
    custom_scores = scores is not None and scores != (2, -3, -5)
    overridden = any(x is not None for x in (args.nq, args.ns, args.length, args.k)) or custom_scores
    if custom_scores:
        cfg_name = f"BitPAl packed M={scores[0]}/I={scores[1]}/G={scores[2]}, 10k x 1M, 150 bp"
    nq = args.nq or nq
    ns_total = args.ns or ns_total
    length = args.length or length
    k = args.k if args.k is not None else k
    mix = args.banded_mix if algo == B.ALGO_BANDED else None

    # ---- this rank's subjects -----------------------------------------------------------------------------
    from bgsa_amd.multi_gpu import ScoreGatherStream, plan_shards
    shards = plan_shards(ns_total, world)            # ONE bucket, contiguous slices, multiples of 64 (dispatch_task, BGSA_KNC/global.c:374-431)
    ns = shards[rank].count
    wd.stage = "workload + query broadcast"
    q_rows, s_rows, ns_pad = make_workload(args.config, algo, nq, ns, length, k, mix, rank, dev, dist)

    aligner = B.DeviceAligner(algo, str(dev), k, scores if algo == B.ALGO_BITPAL else None)
    q_host = q_rows.cpu().numpy()
    aligner.set_queries(q_host)
    aligner.set_subject_rows_device(s_rows.reshape(-1), ns_pad, length, qlen=length)
    out = torch.empty((nq, ns_pad), dtype=aligner.out_dtype, device=dev)
    torch.cuda.synchronize()

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # The sustained shader clock of the timed region (probe.hip): eight sleeping one-wave workgroups, one per XCD, read
    # the shader-clock and the reference-clock counters beside the timed kernels.  Off under a profiler.
    profiled = any(name.startswith(("ROCPROF", "ROCP_")) for name in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")
    use_probe = not args.no_clock_probe and not profiled
    probe_note = None
    if use_probe and algo == B.ALGO_MYERS and aligner.wn >= 25 and aligner.wn <= 32:
        # these kernels hold 253-255 VGPRs at two waves per SIMD: a probe wave finds no register granule beside them and displaces a
        # workgroup on its CU (25 words: 9,091 -> 9,664 ms with the probes, LABNOTES 5.1; 32 words: 3,953 -> 4,112, round 5)
        use_probe = False
        probe_note = "no clock probes beside a kernel that fills the register file (253-255 VGPRs): a probe wave would displace a workgroup"
    L = B.lib()

    def probe_start(expected_s):
        nonlocal use_probe
        if not use_probe:
            return False
        return L.bgsa_hip_clock_probe_start(8, int(min(600000, max(2000, expected_s * 3e3 + 5000))),
                                            ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)) == 0

    def probe_stop():
        mhz, xcc = (ctypes.c_double * 16)(), (ctypes.c_int * 16)()
        n, secs = ctypes.c_int(0), ctypes.c_double(0)
        if L.bgsa_hip_clock_probe_stop(mhz, xcc, 16, ctypes.byref(n), ctypes.byref(secs)) != 0 or n.value == 0:
            return None
        per = sorted((int(xcc[i]), round(float(mhz[i]), 1)) for i in range(n.value))
        return {"sustained_mhz": round(float(np.mean([m for _, m in per])), 1), "nominal_mhz": 2400.0,
                "per_probe_mhz": [m for _, m in per], "probe_xcc": [x for x, _ in per], "probe_seconds": round(secs.value, 3),
                "method": "s_memtime / s_memrealtime deltas of sleeping one-wave workgroups (one per XCD, a stream of their own) "
                          "that run beside the timed kernels from the opening fence to the closing one"}

    def timed(step_fn, steps, warmup, probe=False):
        """W warm-up steps, then exactly K steps bracketed by barrier + synchronize on both sides; wall time =
        max over ranks; kernel time = HIP events on the launch stream around each step."""
        t_w = time.perf_counter()
        for _ in range(warmup):
            step_fn()
        fence()
        per_step = (time.perf_counter() - t_w) / max(warmup, 1)
        probing = probe and probe_start(per_step * steps)
        ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
        ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
        t0 = time.perf_counter()
        for i in range(steps):
            ev0[i].record()
            step_fn()
            ev1[i].record()
        if probing:
            # the probes sleep on a stream of their own until told to stop: the closing fence's device-wide synchronize would
            # wait for their time bound.  So: wait for the timed stream alone, release the probes (tens of microseconds),
            # then the fence of the contract.
            torch.cuda.current_stream(dev).synchronize()
            clock_box[0] = probe_stop()
        else:
            clock_box[0] = None
        fence()
        elapsed = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed, float(np.mean([a.elapsed_time(b) for a, b in zip(ev0, ev1)])) / 1e3

    clock_box = [None]
    power = None
    one_launch = lambda: aligner.score(0, nq, out=out)     # noqa: E731  every query against the rank's resident subjects
    kernel_only = None
    gs = None
    GATHER_BLOCK_ROWS = gather_block_rows(nq)
    n_blocks_step = -(-nq // GATHER_BLOCK_ROWS)
    if world > 1:
        # ---- N > 1.  The job is the SAME bucket as at N = 1, cut over the ranks; what has to be true at the end of a step is
        # what the reference's multi-device host has (cal_mic.c:121-147, 535-536): every device's result tiles of every query
        # block on the host device, rank 0.  So the timed step is: per block of GATHER_BLOCK_ROWS queries the kernel on the
        # rank's slice, its tile handed to the streamed gather (ScoreGatherStream.submit: a side stream sends it to rank 0
        # while the next block is scored), and drain() at the end — the gather is INSIDE the timed region.
        # First, kernels only (the reference's `cal`: one launch over all queries, nothing moved): the side key `kernel_only`.
        wd.stage = "kernel-only leg"
        ko_elapsed, kernel_s = timed(one_launch, 2, 1)
        aligner.check_faults()
        kernel_only = {"what": "one launch per pass over all queries on every rank's slice, no transfer (the reference's cal timer); max over ranks",
                       "passes": 2, "ms_per_pass": round(ko_elapsed / 2 * 1e3, 3),
                       "gcups": round(float(nq) * ns_total * length * length * 2 / ko_elapsed / 1e9, 2)}
        wd.extra["kernel_only"] = kernel_only          # a watchdog that fires in the gathered region still prints this
        q_tile_used = int(L.bgsa_hip_last_query_tile())
        wd.stage = "gather set-up"
        setup_error = None
        try:
            gs = ScoreGatherStream(dist, dev, [sh.count for sh in shards], aligner.out_dtype, block_rows=GATHER_BLOCK_ROWS)
        except Exception as e:      # can only fail locally (allocation): agree on it before anyone enters a transfer the others would wait for
            setup_error = repr(e)
        ok = torch.tensor([0 if setup_error else 1], dtype=torch.int32, device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            print(f"[bench] rank {rank}: gather set-up failed ({setup_error or 'on another rank'})", file=sys.stderr)
            if rank == 0:       # the contract is ONE parseable line whatever happens: the kernels-only figure and what went wrong
                with wd.lock:
                    line = wd.stub()
                    line.update(kernel_only=kernel_only, gather={"error": f"set-up failed: {setup_error or 'on another rank'}"},
                                gather_ok=False, rccl_ok=False)
                    print(json.dumps(line), flush=True)
                    wd.printed = True
            wd.cancel()
            return 4

        def gathered_step():
            for lo in range(0, nq, GATHER_BLOCK_ROWS):
                hi = min(nq, lo + GATHER_BLOCK_ROWS)
                aligner.score(lo, hi, out=out[lo:hi])
                gs.submit(out[lo:hi, :ns])
            gs.drain()

        # An interconnect problem shows as a hang, not an exception: if the gathered region has not finished in time, rank 0 prints
        # a line without a value (the job's figure does not exist) that carries `kernel_only`, gather_ok = false, and every rank
        # leaves with a NON-ZERO status.
        g_limit = float(os.environ.get("BGSA_BENCH_GATHER_TIMEOUT", "300"))

        def mark_gather(line):
            line["gather"] = {"error": f"the timed region (kernels + streamed gather) did not finish within {g_limit:.0f} s; "
                                       "`kernel_only` is what was measured"}
            line["gather_ok"] = False

        wd.stage = "timed region (kernels + streamed gather to rank 0)"
        gather_timer = wd.leg(g_limit, f"the gathered timed region did not finish within {g_limit:.0f} s", mark_gather)
        elapsed, step_s = timed(gathered_step, args.steps, args.warmup, probe=True)
        gather_timer.cancel()
        clock = clock_box[0]
    else:
        wd.stage = "timed region"
        sampler = PowerSampler().start() if (rank == 0 and os.environ.get("BGSA_BENCH_POWER", "1") != "0") else None
        elapsed, kernel_s = timed(one_launch, args.steps, args.warmup, probe=True)
        power = sampler.stop() if sampler else None
        clock = clock_box[0]
        # The probes must not cost anything.  If the wall time of the timed region is not the kernels' time (events on the launch
        # stream) plus launch overhead, something held the launches back — the probes, on a box where they do not run beside the
        # caller's stream after all — and the region is timed again without them; the line then says so instead of carrying a clock.
        if clock is not None:
            slow = elapsed > 1.25 * kernel_s * args.steps + 0.05
            if dist is not None:
                flag = torch.tensor([1 if slow else 0], dtype=torch.int32, device=dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MAX)
                slow = bool(flag.item())
            if slow:
                probe_note = (f"timed region took {elapsed:.3f} s with the clock probes against {kernel_s * args.steps:.3f} s of kernel time: "
                              "timed again without them")
                use_probe = False
                elapsed, kernel_s = timed(one_launch, args.steps, 0, probe=False)
                clock = None
        q_tile_used = int(L.bgsa_hip_last_query_tile())
    aligner.check_faults()
    wd.stage = "after the timed region"

    # ---- N > 1: did the blocks arrive?  One more block after the timed region, checked end to end: every rank sums the tile it
    # sent, rank 0 sums the segment that arrived in its block buffer (the reference's per-device block layout).
    gather_check = None
    if gs is not None:
        wd.stage = "gather content check"
        hi = min(nq, GATHER_BLOCK_ROWS)
        aligner.score(0, hi, out=out[:hi])
        gs.submit(out[:hi, :ns])
        gs.drain()
        sent = [None] * world
        dist.all_gather_object(sent, int(out[:hi, :ns].sum(dtype=torch.int64).item()))
        if rank == 0:
            blk = gs.last_block()
            got, off = [], 0
            for sh in shards:
                got.append(int(blk[off:off + hi * sh.count].sum(dtype=torch.int64).item()))
                off += hi * sh.count
            gather_check = {"segments_ok": got == sent, "rows": hi, "segment_sums_sent": sent if got != sent else None,
                            "segment_sums_at_root": got if got != sent else None,
                            "what": "one block after the timed region: per rank the int64 sum of the tile it sent against the sum of its "
                                    "segment of rank 0's block buffer"}

    # ---- HBM traffic of one launch, modelled from the launch geometry: every query tile re-reads the rank's Peq / Mext
    # blocks once, every score is written once, every stream read once per subject workgroup column (L2-resident: not
    # counted).  Where a PMC pass exists (N = 1) the counters check the model; per rank at N > 1 the model is what
    # there is (config 5: does the ratio grow when eight GPUs' slices shrink? — it depends on the tile, not the slice).
    traffic_model = None
    if q_tile_used > 0 and "blocked" not in aligner.kernel_name():
        peq_bytes = B.group_words(algo, aligner.wn, k) * 4 * (ns_pad // 64)
        model_bytes = -(-nq // q_tile_used) * peq_bytes + float(nq) * ns_pad * out.element_size()
        traffic_model = {"query_tile": q_tile_used, "bytes": round(model_bytes),
                         "ratio_to_algorithmic": round(model_bytes / (float(nq) * ns_pad * algorithmic_bytes_per_pair(algo, length, aligner.wn)), 2)}

    # ---- evidence that N ranks on N devices ran: gathered through the process group itself ----------------------
    props = torch.cuda.get_device_properties(dev)
    me = {"rank": rank, "local_rank": local_rank, "device_index": dev.index, "device": props.name,
          "uuid": str(getattr(props, "uuid", "")), "pci_bus_id": getattr(props, "pci_bus_id", None),
          "host": os.uname().nodename, "pid": os.getpid(), "kernel_ms": round(kernel_s * 1e3, 3),
          "sustained_mhz": clock["sustained_mhz"] if clock else None, "subjects": ns, "traffic_model": traffic_model,
          "peer_access": pre.get("peer_access"), "preflight_all_gather": pre.get("all_gather")}
    ranks_info = [me]
    if dist is not None:
        ranks_info = [None] * world
        dist.all_gather_object(ranks_info, me)

    # size-independent sanity on the full-size output
    checksum = checksum_int64(out, ns) if rank == 0 else 0
    survivors = int((out[:, :ns] != 127).sum().item()) if (rank == 0 and algo == B.ALGO_BANDED) else None

    n_subjects_job = ns_total
    cells_per_step_rank = float(nq) * ns * length * length
    cells_per_step_job = float(nq) * n_subjects_job * length * length
    gcups = cells_per_step_job * args.steps / elapsed / 1e9
    if rank == 0:
        wn = aligner.wn
        ops_cell = algorithmic_ops_per_cell(algo, length, k)
        kernel_gcups = cells_per_step_rank / kernel_s / 1e9
        achieved_ops = kernel_gcups * 1e9 * ops_cell
        pairs_per_launch = float(nq) * ns_pad
        bpp = algorithmic_bytes_per_pair(algo, length, wn)
        vpr = issued_valu_per_row(algo, wn, k, scores)
        src_id = kernel_source_id(algo)
        pmc, pmc_src = pmc_values(args.config, f"_{mix}" if mix else "", src_id) if not overridden else (None, None)
        wave_rows = float(nq) * (ns_pad // 64) * length          # nominal (query row, wave) pairs per launch
        issued = None
        # the generator's own count (rows_ir.py instruction lists x rows x waves): exact when no wave exits early, an
        # upper bound of the work when some do (banded mixes other than `survivors`) — always printed beside the counter
        gen_ops = vpr * 64.0 * wave_rows / kernel_s if vpr else None
        gen = ({"valu_per_row": vpr, "frac": round(gen_ops / VALU_PEAK_OPS, 4),
                "exact": not (algo == B.ALGO_BANDED and mix != "survivors")} if vpr else None)
        if pmc and "SQ_INSTS_VALU" in pmc:                       # measured: the counter of the same command, same kernel source
            ops = pmc["SQ_INSTS_VALU"] * 64.0 / kernel_s
            issued = {"source": f"SQ_INSTS_VALU, {pmc_src}", "valu_insts_per_launch": pmc["SQ_INSTS_VALU"],
                      "valu_per_nominal_wave_row": round(pmc["SQ_INSTS_VALU"] / wave_rows, 3),
                      "achieved": round(ops / 1e12, 2), "unit": "Tops/s", "frac": round(ops / VALU_PEAK_OPS, 4)}
        elif gen and gen["exact"]:
            issued = {"source": "generator instruction lists (rows_ir.py) x rows x waves", "valu_per_row": vpr,
                      "achieved": round(gen_ops / 1e12, 2), "unit": "Tops/s", "frac": round(gen_ops / VALU_PEAK_OPS, 4)}
        if issued is not None:
            issued["generator_count"] = gen
            issued["pmc_file"] = pmc_src
            if clock:    # the same instructions against the clock this box really held: tells a slow box from a slower kernel
                issued["frac_at_sustained_clock"] = round(issued["frac"] * 2400.0 / clock["sustained_mhz"], 4)
            elif power and power.get("sclk_mhz_mean"):   # no probe waves beside this kernel: the driver's own reading of the card's clock
                issued["frac_at_sustained_clock"] = round(issued["frac"] * 2400.0 / power["sclk_mhz_mean"], 4)
                issued["sustained_clock_source"] = "hwmon freq1_input, mean over the timed region"
        traffic = None
        if pmc and "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
            # FETCH_SIZE doubled per the gfx950 note in MI355X_MICROARCH.md §HBM (128-B requests tallied at 64 B); KB units
            traffic = (2 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024
        algorithmic_bytes = pairs_per_launch * bpp
        frac = achieved_ops / VALU_PEAK_OPS
        result = {
            "metric": "GCUPS (cell updates/sec) all-pairs Myers 150bp" if args.config == 2 else f"GCUPS ({cfg_name})",
            "value": round(gcups, 2),
            "unit": "GCUPS",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": cfg_name + (" [SIZE OVERRIDDEN]" if overridden else ""), "queries": nq,
                       "subjects_total": n_subjects_job, "subjects_this_rank": ns, "length_bp": length, "k": k,
                       "parallelism": (f"one bucket cut by plan_shards over {world} ranks; per block of {GATHER_BLOCK_ROWS} queries the kernel on "
                                       f"the rank's slice + its tile streamed to rank 0 (inside the timed region)") if world > 1
                                      else "one GPU: the whole bucket, one launch per step",
                       "gather_block_rows": GATHER_BLOCK_ROWS if world > 1 else None,
                       "kernel": aligner.kernel_name(), "word_num": wn, "kernel_source_id": src_id},
            # N > 1: `value` includes the streamed gather; this is the kernels alone (one launch per pass, max over ranks)
            "kernel_only": kernel_only,
            "clock": clock if clock is not None else ({"sustained_mhz": None, "note": probe_note} if probe_note else None),
            "power": power,
            "ranks_seen": len({(r["host"], r["uuid"] or r["pci_bus_id"] or r["device_index"], r["pid"]) for r in ranks_info}),
            "ranks": ranks_info,
            "gather_ok": None,
            # true once every collective of the run has returned under RCCL; false from the watchdog; null without a group
            "rccl_ok": None,
            "preflight": pre,
            # The roofline that binds this path is the 32-bit integer VALU issue rate.  `achieved` /
            # `frac` are §8(d)'s figure: GCUPS x the REFERENCE's own ALU operations per cell.  It may exceed 1: the
            # kernels need fewer operations per cell than the reference counts (`ops_note`); `issued` is the
            # utilisation of the chip — VALU instructions really issued x 64 lanes / time / peak — and is <= 1.
            "roofline": {
                "bound": "valu",
                "basis": "reference_op_count (SURVEY 8(d))",
                "achieved": round(achieved_ops / 1e12, 3),
                "peak": round(VALU_PEAK_OPS / 1e12, 2),
                "unit": "Tops/s",
                "frac": round(frac, 4),
                "ops_per_cell_reference": round(ops_cell, 4),
                "ops_note": None,
                # the utilisation figures as scalars of this object (a parser that keeps only scalars keeps these): VALU instructions
                # issued x 64 lanes / kernel time / peak, at the nominal and at the sustained clock; instructions per (query row, wave)
                **flat_issued(issued),
                # the clock the chip held under the timed kernels: probe waves; the driver's hwmon reading where no probe runs
                "sustained_mhz": clock["sustained_mhz"] if clock else (power.get("sclk_mhz_mean") if power else None),
                "watts_mean": power["watts_mean"] if power else None,            # and the power it drew (hwmon files, host-side sampling)
                "issued": issued,
                "traffic": traffic,
                "traffic_source": pmc_src if (traffic or pmc is None) else None,
                "algorithmic_bytes": round(algorithmic_bytes),
                "traffic_ratio": round(traffic / algorithmic_bytes, 2) if traffic else None,
                "traffic_model": traffic_model,
                "kernel_ms": round(kernel_s * 1e3, 3),
                "kernel_gcups": round(kernel_gcups, 1),
                "note": "peak = 256 CU x 4 SIMD x 32 lanes x 2.4 GHz; kernel time from HIP events on the launch stream; "
                        "traffic_ratio = counter bytes / SURVEY's algorithmic bytes (its model: one Peq block per 100 queries; the counter launches tile up to 128)",
                "hbm": {"bound": "hbm", "achieved": round(algorithmic_bytes / kernel_s / 1e9, 2), "peak": HBM_PEAK / 1e9,
                        "unit": "GB/s", "frac": round(algorithmic_bytes / kernel_s / HBM_PEAK, 6),
                        "bytes_per_pair": round(bpp, 3),
                        "measured_gbps": round(traffic / kernel_s / 1e9, 2) if traffic else None},
            },
            "checksum": checksum,
        }
        if issued and frac > 1:
            per_cell = issued["achieved"] * 1e12 / (kernel_gcups * 1e9)
            result["roofline"]["ops_note"] = (f"frac > 1 is not skipped work: the kernel issues {per_cell:.3f} lane-ops per cell where the "
                                              f"reference counts {ops_cell:.3f} (v_bitop3 + VCC carry chains + 32 data bits per word)")
        if algo == B.ALGO_BANDED:
            result["config"]["banded_mix"] = {"name": mix, "what": BANDED_MIXES[mix],
                                              "pairs_not_rejected": survivors, "fraction": survivors / (float(nq) * ns)}

    # ---- N > 1: the same steps with the score gather of SURVEY §2a C3 streamed beside them -------------------
    gather_info = None
    print_lock = wd.lock
    if rank == 0:
        wd.result = result          # from here on a firing watchdog prints the measured line, not a stub
    # ---- N > 1: what the timed region moved, and the same steps in the reference's own block size beside it ----------------
    if world > 1:
        gather_info = None
        if rank == 0:
            gather_info = {"what": f"inside the timed region: {n_blocks_step} blocks of {GATHER_BLOCK_ROWS} queries per step — kernel per block on every "
                                   "rank's slice, the tiles of all ranks to rank 0 in the reference's per-device block layout (cal_mic.c:535-536), "
                                   "grouped send/recv on a side stream beside the next block's kernel, two block buffers",
                           "block_rows": GATHER_BLOCK_ROWS, "blocks_per_step": n_blocks_step,
                           "bytes_to_root_per_block": int(sum(sh.count for sh in shards[1:]) * min(nq, GATHER_BLOCK_ROWS) * out.element_size()),
                           "bytes_to_root_per_step": int(sum(sh.count for sh in shards[1:]) * nq * out.element_size()),
                           "root_blocks_received": gs.n_submitted,
                           "gcups_with_gather": round(gcups, 2), "gcups_kernels_only": kernel_only["gcups"],
                           "content_check": gather_check}
        small_wanted = os.environ.get("BGSA_BENCH_GATHER", "1") != "0" and GATHER_BLOCK_ROWS != REF_BUCKET_COUNT
        small_info = None
        if small_wanted:
            # The reference moves REF_BUCKET_COUNT = 100 query rows per block: ten such blocks, with and without the gather (an
            # optional leg under a limit of its own: its expiry keeps the measured value and marks the leg)
            limit = float(os.environ.get("BGSA_BENCH_SMALL_GATHER_TIMEOUT", "120"))
            small_timer = wd.leg(limit, f"the reference-sized gather leg did not finish within {limit:.0f} s",
                                 lambda line: line.__setitem__("gather_blocks_of_100", {"error": f"did not finish within {limit:.0f} s"}))
            wd.stage = "gather leg, reference-sized blocks"
            try:
                gs100 = ScoreGatherStream(dist, dev, [sh.count for sh in shards], aligner.out_dtype, block_rows=REF_BUCKET_COUNT)
                nq_g = min(nq, 10 * REF_BUCKET_COUNT)

                def blocks100(submit):
                    for lo in range(0, nq_g, REF_BUCKET_COUNT):
                        hi = min(nq_g, lo + REF_BUCKET_COUNT)
                        aligner.score(lo, hi, out=out[lo:hi])
                        if submit:
                            gs100.submit(out[lo:hi, :ns])
                    if submit:
                        gs100.drain()

                g_elapsed, _ = timed(lambda: blocks100(True), 1, 1)
                k_elapsed, _ = timed(lambda: blocks100(False), 1, 1)
                cells_g = float(nq_g) * ns_total * length * length
                small_info = {"what": f"{-(-nq_g // REF_BUCKET_COUNT)} blocks of {REF_BUCKET_COUNT} queries (the reference's REF_BUCKET_COUNT), same "
                                      "mechanism, after the timed region",
                              "gcups_with_gather": round(cells_g / g_elapsed / 1e9, 1),
                              "gcups_kernels_only_same_blocks": round(cells_g / k_elapsed / 1e9, 1),
                              "bytes_to_root_per_block": int(sum(sh.count for sh in shards[1:]) * REF_BUCKET_COUNT * out.element_size()),
                              "root_blocks_received": gs100.n_submitted}
                del gs100
            except Exception as e:      # an optional leg never takes the measured line with it
                small_info = {"error": repr(e)}
            small_timer.cancel()
        if rank == 0:
            with print_lock:
                result["gather"] = gather_info
                result["gather_ok"] = bool(gather_check and gather_check["segments_ok"])
                if small_info is not None:
                    result["gather_blocks_of_100"] = small_info

    # ---- N > 1, the default invocation: the weak-scaling figure of rounds 1-4 as a side key — every rank a full bucket of its
    # own (config 2: 1M subjects per rank, job = N x 1M), kernels only.  It is linear in N by construction; it says what the
    # ranks do when nothing is shared, nothing about the gather.
    if dist is not None and world > 1 and args.config == 2 and not args.no_weak and args.length is None:
        wd.stage = "weak leg (a full bucket per rank)"
        weak, err = None, None
        weak_limit = float(os.environ.get("BGSA_BENCH_WEAK_TIMEOUT", "180"))
        weak_timer = wd.leg(weak_limit, f"weak leg did not finish within {weak_limit:.0f} s",
                            lambda line: line.__setitem__("weak", {"error": f"did not finish within {weak_limit:.0f} s"}))
        try:
            del out
            aligner = None
            gs = None
            torch.cuda.empty_cache()
            w_entry, (wa, wout, _wq, _ws, _) = side_config(2, dev, L, nq=args.nq, ns=args.ns, passes=1, rank=rank, dist=dist, keep=True)
            w_el, w_ks = timed(lambda: wa.score(0, nq, out=wout), 2, 0)
            wa.check_faults()
            weak = {"what": "every rank scores all queries against a full bucket of its own (rounds 1-4's headline at N > 1): kernels only, "
                            "no transfer; job = N buckets",
                    "scaling": "weak", "subjects_per_rank": ns_total, "subjects_total": ns_total * world, "passes": 2,
                    "ms_per_pass": round(w_el / 2 * 1e3, 3),
                    "gcups": round(float(nq) * ns_total * world * length * length * 2 / w_el / 1e9, 1)}
            del wa, wout, _wq, _ws
            torch.cuda.empty_cache()
        except Exception as e:
            err = repr(e)
        weak_timer.cancel()
        if rank == 0:
            with print_lock:
                result["weak"] = weak if weak is not None else {"error": err}

    # ---- N > 1, the default invocation: BASELINE configs[4] — ONE 1M-subject bucket of 1000 bp reads cut by plan_shards
    # over the ranks (dispatch_task, BGSA_KNC/global.c:374-431; per-device offload + result download, cal_mic.c:121-147) —
    # kernel-only and with the streamed per-block gather, so that a SCALE run times the north star's sharded config too.
    if dist is not None and world > 1 and args.config == 2 and not args.no_strong and args.length is None:
        wd.stage = "config 5 sharded"
        strong, err = None, None
        strong_limit = float(os.environ.get("BGSA_BENCH_STRONG_TIMEOUT", "300"))
        # as for the gather leg: a rank that fails here while its peers wait in a collective shows as a hang, and the
        # measured line must still go out (marked), with every rank leaving non-zero
        strong_timer = wd.leg(strong_limit, f"strong leg did not finish within {strong_limit:.0f} s",
                              lambda line: line.__setitem__("config5_sharded", {"error": f"did not finish within {strong_limit:.0f} s"}))
        try:
            out = None
            aligner = None
            gs = None
            torch.cuda.empty_cache()
            algo5, name5, cfg_nq5, cfg_ns5, len5, _, _ = CONFIGS[5]
            nq5, ns5 = args.nq or cfg_nq5, args.ns or cfg_ns5
            shards5 = plan_shards(ns5, world)
            mine5 = shards5[rank].count
            entry, (a5, out5, _q5, _s5, _) = side_config(5, dev, L, nq=nq5, ns=mine5, passes=1, rank=rank, dist=dist, keep=True)
            el5, ks5 = timed(lambda: a5.score(0, nq5, out=out5), 2, 0)
            gs5 = ScoreGatherStream(dist, dev, [s.count for s in shards5], a5.out_dtype, block_rows=REF_BUCKET_COUNT)

            def blocks5(submit):
                for lo in range(0, nq5, REF_BUCKET_COUNT):
                    hi = min(nq5, lo + REF_BUCKET_COUNT)
                    a5.score(lo, hi, out=out5[lo:hi])
                    if submit:
                        gs5.submit(out5[lo:hi, :mine5])
                if submit:
                    gs5.drain()

            g_el5, _ = timed(lambda: blocks5(True), 1, 1)
            k_el5, _ = timed(lambda: blocks5(False), 1, 1)
            a5.check_faults()
            cells5 = float(nq5) * ns5 * len5 * len5
            per_rank = [None] * world
            dist.all_gather_object(per_rank, {"rank": rank, "subjects": mine5, "kernel_ms": round(ks5 * 1e3, 3),
                                              "query_tile": int(L.bgsa_hip_last_query_tile())})
            strong = {"workload": name5 + (" [SIZE OVERRIDDEN]" if (nq5, ns5) != (cfg_nq5, cfg_ns5) else ""),
                      "scaling": "strong", "queries": nq5, "subjects_total": ns5, "length_bp": len5,
                      "partition": "plan_shards: contiguous subject slices, multiples of 64 (BGSA_KNC/global.c:374-431)",
                      "kernel": entry["kernel"], "passes": 2,
                      "ms_per_pass": round(el5 / 2 * 1e3, 3), "gcups": round(cells5 * 2 / el5 / 1e9, 1),
                      "ranks": per_rank,
                      "gather": {"what": f"{(nq5 + REF_BUCKET_COUNT - 1) // REF_BUCKET_COUNT} blocks of {REF_BUCKET_COUNT} queries, every "
                                         "rank's tile to rank 0 on a side stream beside the next block's kernel (cal_mic.c:139-147, 535-536)",
                                 "gcups_with_gather": round(cells5 / g_el5 / 1e9, 1),
                                 "gcups_kernels_only_same_blocks": round(cells5 / k_el5 / 1e9, 1),
                                 "bytes_to_root_per_block": int(sum(s.count for s in shards5[1:]) * REF_BUCKET_COUNT * out5.element_size()),
                                 "root_blocks_checked": gs5.blocks_checked}}
            del a5, out5, _q5, _s5, gs5
            torch.cuda.empty_cache()
        except Exception as e:      # an optional leg never takes the measured line with it
            err = repr(e)
        strong_timer.cancel()
        if rank == 0:
            with print_lock:
                result["config5_sharded"] = strong if strong is not None else {"error": err}

    # ---- config 3, one GPU: the other subject mixes (same kernel, same sizes) ---------------------------------
    if algo == B.ALGO_BANDED and world == 1 and args.banded_variants and not overridden:
        variants = {}
        for name in [v for v in args.banded_variants.split(",") if v and v != mix]:
            del s_rows
            torch.cuda.empty_cache()
            vq, s_rows, _ = make_workload(args.config, algo, nq, ns, length, k, name, rank, dev, None)
            aligner.set_queries(vq.cpu().numpy())
            aligner.set_subject_rows_device(s_rows.reshape(-1), ns_pad, length, qlen=length)
            _, v_kernel_s = timed(lambda: aligner.score(0, nq, out=out), args.steps, 1)
            aligner.check_faults()
            v = {"what": BANDED_MIXES[name], "kernel_ms": round(v_kernel_s * 1e3, 3),
                 "gcups": round(cells_per_step_rank / v_kernel_s / 1e9, 1),
                 "pairs_not_rejected_fraction": float((out[:, :ns] != 127).sum().item()) / (float(nq) * ns)}
            vp, vsrc = pmc_values(args.config, f"_{name}")
            if vp and "SQ_INSTS_VALU" in vp:
                v["issued_frac"] = round(vp["SQ_INSTS_VALU"] * 64.0 / v_kernel_s / VALU_PEAK_OPS, 4)
                v["issued_source"] = f"SQ_INSTS_VALU, {vsrc}"
            elif name == "survivors":
                v["issued_frac"] = round(issued_valu_per_row(algo, aligner.wn, k) * 64.0 * float(nq) * (ns_pad // 64) * length
                                         / v_kernel_s / VALU_PEAK_OPS, 4)
                v["issued_source"] = "generator instruction lists x all rows (no wave exits early in this mix)"
            variants[name] = v
        result["banded_variants"] = variants
        # restore the main mix for the legs below
        q_rows, s_rows, _ = make_workload(args.config, algo, nq, ns, length, k, mix, rank, dev, None)
        q_host = q_rows.cpu().numpy()

    if rank == 0 and world > 1:
        result["total_gcups"] = {"skipped": "measured by rank 0 at N = 1 only"}
        result["cpu_baseline"] = {"skipped": "measured by rank 0 at N = 1 only"}
        result["other_configs"] = {"skipped": "N = 1 only; at N > 1 the line carries `config5_sharded`"}
    if rank == 0:
        if not args.no_total and world == 1:
            wd.stage = "total_gcups leg"
            try:
                del out
                torch.cuda.empty_cache()
                runs = [total_gcups_leg(algo, k, scores if algo == B.ALGO_BITPAL else None, q_host, s_rows, ns, ns_pad, length, dev)]
                if runs[0]["wall_ms"] < 2000:   # a short leg is at the mercy of the shared host's scheduling: take the better of two
                    runs.append(total_gcups_leg(algo, k, scores if algo == B.ALGO_BITPAL else None, q_host, s_rows, ns, ns_pad,
                                                length, dev))
                result["total_gcups"] = max(runs, key=lambda r: r["value"])
                result["total_gcups"]["runs_wall_ms"] = [r["wall_ms"] for r in runs]
            except Exception as e:
                result["total_gcups"] = {"error": repr(e)}
        if world == 1 and args.config == 2 and not args.no_other_configs and args.length is None and args.k is None:
            # ---- the other BASELINE GPU configs in the line the driver runs: 3 (planted mix), 4, 5 -----------------
            wd.stage = "other_configs leg"
            try:
                del out
            except NameError:
                pass
            aligner = None
            torch.cuda.empty_cache()
            others = {}
            t_leg = time.perf_counter()
            for cid in (3, 4, 5):
                wd.stage = f"other_configs leg: config {cid}"
                try:
                    others[str(cid)] = side_config(cid, dev, L, nq=args.nq, ns=args.ns, cpu=not args.no_cpu_baseline)
                except Exception as e:
                    others[str(cid)] = {"error": repr(e)}
            others["what"] = ("BASELINE configs 3 (SURVEY 8(d)'s planted mix), 4 and 5 on this GPU after the timed region: workload "
                              "resident in HBM, one warm-up + two passes each, HIP events on the launch stream")
            others["wall_s"] = round(time.perf_counter() - t_leg, 1)
            result["other_configs"] = others
            # ... and as scalars of the two objects every reader of the line keeps: `roofline` and `cpu_baseline`
            for cid in ("3", "4", "5"):
                e = others[cid]
                if "error" in e:
                    continue
                result["roofline"][f"cfg{cid}_gcups"] = e["gcups"]
                result["roofline"][f"cfg{cid}_kernel_ms"] = e["kernel_ms"]
                result["roofline"][f"cfg{cid}_issued_frac"] = e["issued_frac"]
                result["roofline"][f"cfg{cid}_issued_frac_sustained"] = e["issued_frac_sustained"]
                result["roofline"][f"cfg{cid}_valu_per_wave_row"] = e["valu_per_wave_row"]
                result["roofline"][f"cfg{cid}_watts_mean"] = e["watts_mean"]
        if not args.no_cpu_baseline and world == 1 and not custom_scores:  # the reference commits 2/-3/-5 only
            wd.stage = "cpu_baseline leg"
            cq, cs = (int(x) for x in args.cpu_sample.split("x"))
            cq, cs = min(cq, nq), min(cs, ns) // 8 * 8
            result["cpu_baseline"] = cpu_baseline(q_rows[:cq].cpu().numpy(), s_rows[:cs, :length].cpu().numpy(), algo, k)
            result["cpu_baseline"]["gpu_over_cpu"] = round(gcups / result["cpu_baseline"]["value"], 1)
            result["cpu_baseline"]["value"] = round(result["cpu_baseline"]["value"], 2)
            if isinstance(result["cpu_baseline"].get("sse"), dict):     # the committed SSE build's figure as a scalar beside the derived AVX2 one
                result["cpu_baseline"]["sse_value"] = result["cpu_baseline"]["sse"]["value"]
            for cid in ("3", "4", "5"):         # the other configs' baselines (measured in the other_configs leg) as scalars here too
                b = (result.get("other_configs") or {}).get(cid, {}).get("cpu_baseline") if isinstance(result.get("other_configs"), dict) else None
                if b and "value" in b:
                    result["cpu_baseline"][f"cfg{cid}_value"] = b["value"]
                    result["cpu_baseline"][f"cfg{cid}_kind"] = b["kind"]
                    result["cpu_baseline"][f"cfg{cid}_cores"] = b["cores"]
                    if isinstance(b.get("sse"), dict):
                        result["cpu_baseline"][f"cfg{cid}_sse_value"] = b["sse"]["value"]
        with print_lock:
            if dist is not None:
                result["rccl_ok"] = True if backend == "nccl" else None    # every collective up to here has returned
            print(json.dumps(result), flush=True)
            wd.printed = True
    wd.stage = "closing barrier"
    if dist is not None:
        try:
            dist.barrier()
        except Exception as e:       # the line is out; a peer that left early is not this rank's failure
            print(f"[bench] closing barrier: {e!r}", file=sys.stderr)
        wd.cancel()
        try:
            dist.destroy_process_group()
        except Exception:
            pass
    wd.cancel()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
