"""oracle — TEST INFRASTRUCTURE ONLY.

CPU checker for the HIP hot path: a ctypes view of oracle/bgsa_oracle.c (our restatement of the
reference's Myers / banded Myers / BitPAl kernels) and a runner for the REAL reference binaries
compiled into oracle/_ref/ by `make -C oracle ref`.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product (bgsa_amd, libbgsa_hip.so) never does.

Parity status: pinned — tests/test_oracle.py checks every restated kernel against fixtures
minted from the compiled reference (scripts/make_golden.py -> tests/golden/*.npz).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
import tempfile
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
LIB_PATH = HERE / "_build" / "libbgsa_oracle.so"
REF_DIR = HERE / "_ref"

_lib = None


def build(ref: bool | None = None) -> None:
    """Compile the restatement (always) and the real reference (when /root/reference exists)."""
    subprocess.run(["make", "-C", str(HERE), "all"], check=True, stdout=subprocess.DEVNULL)
    if ref is None:
        ref = Path(os.environ.get("BGSA_REFERENCE", "/root/reference")).is_dir()
    if ref:
        subprocess.run(["make", "-C", str(HERE), "ref"], check=True, stdout=subprocess.DEVNULL)


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            build(ref=False)
        L = ctypes.CDLL(str(LIB_PATH))
        c_p, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
        common = [c_p, i64, i32, c_p, i64, i32]
        for name in ("myers64", "myers31", "bitpal", "dp_edit", "dp_edit_semiglobal"):
            f = getattr(L, "bgsa_oracle_" + name)
            f.argtypes = common + [c_p, i32]
            f.restype = None
        L.bgsa_oracle_banded64.argtypes = common + [i32, c_p, i32]
        L.bgsa_oracle_banded64.restype = None
        L.bgsa_oracle_dp_banded.argtypes = common + [i32, c_p, i32]
        L.bgsa_oracle_dp_banded.restype = None
        L.bgsa_oracle_dp_nw.argtypes = common + [i32, i32, i32, c_p, i32]
        L.bgsa_oracle_dp_nw.restype = None
        L.bgsa_oracle_dp_semiglobal.argtypes = common + [i32, i32, i32, c_p, i32]
        L.bgsa_oracle_dp_semiglobal.restype = None
        L.bgsa_oracle_myers_avx2.argtypes = common + [c_p, i32]
        L.bgsa_oracle_myers_avx2.restype = ctypes.c_double
        _lib = L
    return _lib


# --------------------------------------------------------------------------------------------
# Row buffers: the reference's in-memory / on-disk sequence format (len bytes + '\n' per row).
# --------------------------------------------------------------------------------------------

def rows_to_buffer(rows: np.ndarray) -> np.ndarray:
    """[n, len] uint8 ASCII -> flat [n*(len+1)] uint8 with '\\n' terminators."""
    rows = np.ascontiguousarray(rows, dtype=np.uint8)
    n, length = rows.shape
    buf = np.full((n, length + 1), ord("\n"), dtype=np.uint8)
    buf[:, :length] = rows
    return buf.reshape(-1)


def _ptr(a: np.ndarray) -> ctypes.c_void_p:
    return ctypes.c_void_p(a.ctypes.data)


def _run(name: str, queries: np.ndarray, subjects: np.ndarray, dtype, extra=(), threads: int = 0):
    q = np.ascontiguousarray(queries, dtype=np.uint8)
    s = np.ascontiguousarray(subjects, dtype=np.uint8)
    nq, qlen = q.shape
    ns, slen = s.shape
    qb, sb = rows_to_buffer(q), rows_to_buffer(s)
    out = np.zeros((nq, ns), dtype=dtype)
    f = getattr(lib(), "bgsa_oracle_" + name)
    r = f(_ptr(qb), nq, qlen, _ptr(sb), ns, slen, *extra, _ptr(out), threads)
    return out, r


def myers64(q, s, threads=0):
    return _run("myers64", q, s, np.int16, threads=threads)[0]


def myers31(q, s, threads=0):
    return _run("myers31", q, s, np.int16, threads=threads)[0]


def bitpal(q, s, threads=0):
    return _run("bitpal", q, s, np.int16, threads=threads)[0]


def banded64(q, s, k, threads=0):
    return _run("banded64", q, s, np.int8, extra=(int(k),), threads=threads)[0]


def dp_edit(q, s, threads=0):
    return _run("dp_edit", q, s, np.int16, threads=threads)[0]


def dp_edit_semiglobal(q, s, threads=0):
    """-(lowest edit distance of the whole subject against any query substring ending anywhere):
    the generator's Myers semi-global mode."""
    return _run("dp_edit_semiglobal", q, s, np.int16, threads=threads)[0]


def dp_nw(q, s, match=2, mismatch=-3, gap=-5, threads=0):
    return _run("dp_nw", q, s, np.int16, extra=(match, mismatch, gap), threads=threads)[0]


def dp_semiglobal(q, s, match=2, mismatch=-3, gap=-5, threads=0):
    """Query end to end, free subject overhangs: max over the last DP row (generator option -s)."""
    return _run("dp_semiglobal", q, s, np.int16, extra=(match, mismatch, gap), threads=threads)[0]


def dp_banded(q, s, k, threads=0):
    return _run("dp_banded", q, s, np.int8, extra=(int(k),), threads=threads)[0]


def myers_avx2_timed(q, s, threads=0):
    """Returns (scores, seconds in the scoring loop).  ns must be a multiple of 8."""
    out, secs = _run("myers_avx2", q, s, np.int16, threads=threads)
    if secs < 0:
        raise ValueError("myers_avx2 needs a subject count that is a multiple of 8")
    return out, secs


# --------------------------------------------------------------------------------------------
# Deterministic synthetic reads: splitmix64 -> "ACGT"[x >> 62]
# --------------------------------------------------------------------------------------------

def splitmix64(seed: int, n: int) -> np.ndarray:
    with np.errstate(over="ignore"):
        idx = np.arange(1, n + 1, dtype=np.uint64)
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def gen_reads(seed: int, n: int, length: int) -> np.ndarray:
    """[n, length] uint8 ASCII over ACGT."""
    x = splitmix64(seed, n * length)
    return np.frombuffer(b"ACGT", dtype=np.uint8)[(x >> np.uint64(62)).astype(np.int64)].reshape(n, length)


def mutate(rows: np.ndarray, edits, seed: int) -> np.ndarray:
    """Apply `edits[i]` random substitutions / insertions / deletions to row i, keeping length."""
    rng = np.random.default_rng(seed)
    rows = np.array(rows, dtype=np.uint8, copy=True)
    n, length = rows.shape
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    out = np.empty_like(rows)
    for i in range(n):
        seq = list(rows[i])
        for _ in range(int(edits[i])):
            kind = rng.integers(0, 3)
            pos = int(rng.integers(0, len(seq)))
            if kind == 0:
                others = [b for b in acgt if b != seq[pos]]
                seq[pos] = others[int(rng.integers(0, len(others)))]
            elif kind == 1:
                seq.insert(pos, acgt[rng.integers(0, 4)])
            elif len(seq) > 1:
                del seq[pos]
        while len(seq) < length:
            seq.append(acgt[rng.integers(0, 4)])
        out[i] = np.array(seq[:length], dtype=np.uint8)
    return out


# --------------------------------------------------------------------------------------------
# The real reference (compiled binaries under oracle/_ref/)
# --------------------------------------------------------------------------------------------

REF_VARIANTS = {
    # name          : (dir,             result dtype)
    "original_cpu": ("original_cpu", np.int16),   # Myers scalar 64-bit — THE oracle (BASELINE.json)
    "original_sse": ("original_sse", np.int16),   # Myers SSE 4x32
    "original_avx2": ("original_avx2", np.int16),  # BitPAl (2,-3,-5) AVX2 8x32
    "banded_cpu": ("banded_cpu", np.int8),        # banded Myers scalar 64-bit
    # Myers AVX2 8x32: the generator's AVX2 instance of original/BGSA_SSE/align_core.c (derive_avx2_myers.py) on
    # original/BGSA_AVX2's host files — reference-derived, used as the Myers cpu_baseline
    "original_avx2_myers": ("original_avx2_myers", np.int16),
}


def have_reference(variant: str = "original_cpu") -> bool:
    d = REF_DIR / REF_VARIANTS[variant][0]
    return (d / "aligner").exists() and (d / "convert").exists()


def run_reference(variant: str, queries: np.ndarray, subjects: np.ndarray, threads: int = 1,
                  k: int | None = None, want_scores: bool = True, tmp_root: str | None = None):
    """Run the reference `aligner` (+ `convert -r`) on the given reads.

    Returns (scores [nq, ns] or None, stdout text of the aligner).  The result is read from
    `convert -r` text, the canonical order (query-major, subjects in file order, padding
    removed) — reference original/BGSA_CPU/convert.c:167-277.
    """
    d = REF_DIR / REF_VARIANTS[variant][0]
    if not have_reference(variant):
        raise FileNotFoundError(f"reference binaries missing under {d}; run `make -C oracle ref`")
    q = np.ascontiguousarray(queries, dtype=np.uint8)
    s = np.ascontiguousarray(subjects, dtype=np.uint8)
    with tempfile.TemporaryDirectory(dir=tmp_root) as tmp:
        tmp = Path(tmp)
        rows_to_buffer(q).tofile(tmp / "query.txt")
        rows_to_buffer(s).tofile(tmp / "subject.txt")
        cmd = [str(d / "aligner"), "-q", "query.txt", "-d", "subject.txt", "-f", "result.txt",
               "-N", str(threads)]
        if k is not None:
            cmd += ["-k", str(k)]
        env = dict(os.environ, OMP_NUM_THREADS=str(threads))
        p = subprocess.run(cmd, cwd=tmp, env=env, check=True, capture_output=True, text=True)
        scores = None
        if want_scores:
            subprocess.run([str(d / "convert"), "-r", "result.txt", "-o", "scores.txt"], cwd=tmp,
                           check=True, capture_output=True)
            flat = np.loadtxt(tmp / "scores.txt", dtype=np.int64, ndmin=1)
            scores = flat.reshape(q.shape[0], s.shape[0]).astype(REF_VARIANTS[variant][1])
        return scores, p.stdout


def parse_gcups(stdout: str) -> dict:
    """Pull `cal GCUPS` / `Total GCUPS` / `cal_total_times` out of the aligner's report
    (reference original/BGSA_CPU/cal_cpu.c:459-475)."""
    out = {}
    for line in stdout.splitlines():
        line = " ".join(line.split())  # BGSA_SSE prints "cal   GCUPS is"
        if line.startswith("cal GCUPS is"):
            out["cal_gcups"] = float(line.split()[-1])
        elif line.startswith("Total GCUPS is"):
            out["total_gcups"] = float(line.split()[-1])
        elif line.startswith("cal_total_times"):
            out["cal_seconds"] = float(line.split()[-1].rstrip("s"))
    return out
