"""bgsa_amd — Python view of the MI355X backend for BGSA's all-pairs bit-parallel alignment.

The product is the C-ABI shared library `libbgsa_hip.so` (include/bgsa_hip.h, built from
bgsa_amd/csrc/*.hip for gfx950).  This module only loads it with ctypes and adds thin helpers
that hold device memory in torch tensors (plumbing: allocation, streams, torch.distributed).
There is no CPU fallback: if the library is missing, importing the compute API raises.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
# BGSA_HIP_LIB: another build of the same library (A/B measurements of generator variants)
LIB_PATH = Path(os.environ["BGSA_HIP_LIB"]) if os.environ.get("BGSA_HIP_LIB") else HERE / "libbgsa_hip.so"
# the A/B flavour of the same library (`make -C bgsa_amd/csrc ab`): additionally every measured-and-not-adopted kernel a
# measurement knob can select and three more BitPAl score sets; the knob tests and A/B scripts load it through BGSA_HIP_LIB
LIB_AB_PATH = HERE / "libbgsa_hip_ab.so"
INCLUDE = HERE.parent / "include" / "bgsa_hip.h"

ALGO_MYERS, ALGO_BANDED, ALGO_BITPAL = 0, 1, 2
V_NUM = 64

_lib = None


class Params(ctypes.Structure):
    """bgsa_hip_params_t of include/bgsa_hip.h: everything a scoring call reads, as one value."""
    _fields_ = [("algo", ctypes.c_int), ("alignment", ctypes.c_int), ("match", ctypes.c_int),
                ("mismatch", ctypes.c_int), ("gap", ctypes.c_int), ("k", ctypes.c_int)]


class BgsaHipError(RuntimeError):
    pass


def build_library(verbose: bool = False) -> Path:
    """Compile libbgsa_hip.so in-tree (hipcc --offload-arch=gfx950).  Works without a GPU."""
    out = None if verbose else subprocess.DEVNULL
    subprocess.run(["make", "-C", str(HERE / "csrc"), "-j8", "all", "ab"], check=True, stdout=out)
    subprocess.run(["make", "-C", str(HERE / "host")], check=True, stdout=out)  # aligner, convert (C)
    return LIB_PATH


def declared_symbols() -> list[str]:
    """Function names declared in include/bgsa_hip.h (used by the symbol-export test)."""
    import re
    text = INCLUDE.read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"^\s*#.*$", "", text, flags=re.M)  # preprocessor lines
    names = re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{]*\)\s*;", text)
    return sorted(set(n for n in names if n not in ("defined",)))


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise BgsaHipError(
            f"{LIB_PATH} is missing — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C bgsa_amd/csrc`.  There is no CPU fallback for the HIP path.")
    # torch first: it bundles its own ROCm runtime, and the process must end up with ONE HIP/HSA
    # runtime.  Loaded in the other order (this library pulling in /opt/rocm's libamdhip64 before torch
    # brings its libhsa-runtime64) the runtime finds no device.
    import torch  # noqa: F401
    L = ctypes.CDLL(str(LIB_PATH))
    vp, i32, i64, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_size_t
    L.bgsa_hip_last_error.restype = ctypes.c_char_p
    L.bgsa_hip_select_algorithm.argtypes = [i32]
    L.bgsa_hip_select_scores.argtypes = [i32, i32, i32]
    L.bgsa_hip_select_alignment.argtypes = [i32]
    ip = ctypes.POINTER(i32)
    L.bgsa_hip_score_set.argtypes = [i32, ip, ip, ip, ip]
    L.bgsa_hip_word_num.argtypes = [i32, i32, i32, i32]
    L.bgsa_hip_group_words.argtypes = [i32, i32, i32]
    L.bgsa_hip_group_words.restype = sz
    L.bgsa_hip_handle_reads_dev.argtypes = [i32, vp, i64, i32, i64, i32, i32, vp, vp]
    L.bgsa_hip_map_queries_dev.argtypes = [vp, i64, vp]
    L.bgsa_hip_cal_align_score_dev.argtypes = [i32, vp, vp, vp, i32, i32, i64, i32, i32, i32, i32, vp, sz, vp]
    L.bgsa_hip_workspace_bytes.argtypes = [i32, i32, i32, i32]
    L.bgsa_hip_workspace_bytes.restype = sz
    pp = ctypes.POINTER(Params)
    L.bgsa_hip_current_params.argtypes = [pp]
    L.bgsa_hip_workspace_bytes_ex.argtypes = [pp, i32, i32, i32]
    L.bgsa_hip_workspace_bytes_ex.restype = sz
    L.bgsa_hip_cal_align_score_ex.argtypes = [pp, vp, vp, vp, i32, i32, i64, i32, i32, i32, vp, sz, vp]
    L.bgsa_hip_stream_faults.argtypes = [i32]
    L.bgsa_hip_debug_inject_stream_fault.argtypes = [i32]
    L.bgsa_hip_set_auto_resident.argtypes = [i32]
    L.bgsa_hip_set_strict_resident.argtypes = [i32]
    L.bgsa_hip_stale_ranges.argtypes = [ctypes.POINTER(ctypes.c_uint64)]
    L.bgsa_hip_bucket_resident.argtypes = [vp, sz, i32]
    L.bgsa_hip_bucket_release.argtypes = [vp]
    u64p = ctypes.POINTER(ctypes.c_uint64)
    L.bgsa_hip_seam_stats.argtypes = [u64p, u64p, u64p]
    L.bgsa_hip_row_cache_stats.argtypes = [u64p, u64p]
    L.bgsa_hip_event_create.argtypes = [ctypes.POINTER(vp)]
    L.bgsa_hip_event_destroy.argtypes = [vp]
    L.bgsa_hip_event_record.argtypes = [vp, vp]
    L.bgsa_hip_event_synchronize.argtypes = [vp]
    L.bgsa_hip_event_elapsed_ms.argtypes = [vp, vp, ctypes.POINTER(ctypes.c_float)]
    L.bgsa_hip_stream_wait_event.argtypes = [vp, vp]
    L.bgsa_hip_query_stream.argtypes = [i32, vp, i32, i32, vp, i32]
    L.bgsa_hip_kernel_name.argtypes = [i32, i32]
    L.bgsa_hip_kernel_name.restype = ctypes.c_char_p
    L.bgsa_hip_malloc.argtypes = [ctypes.POINTER(vp), sz]
    L.bgsa_hip_free.argtypes = [vp]
    L.bgsa_hip_malloc_host.argtypes = [ctypes.POINTER(vp), sz]
    L.bgsa_hip_free_host.argtypes = [vp]
    L.bgsa_hip_memcpy_h2d.argtypes = [vp, vp, sz, vp]
    L.bgsa_hip_memcpy_d2h.argtypes = [vp, vp, sz, vp]
    L.bgsa_hip_memset.argtypes = [vp, i32, sz, vp]
    L.bgsa_hip_stream_synchronize.argtypes = [vp]
    L.bgsa_hip_stream_create.argtypes = [ctypes.POINTER(vp)]
    L.bgsa_hip_stream_destroy.argtypes = [vp]
    L.bgsa_hip_set_device.argtypes = [i32]
    # host-buffer BGSA surface
    L.hip_handle_reads.argtypes = [vp, vp, i32, i64, i64]
    L.hip_handle_reads.restype = None
    L.hip_cal_align_score.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]
    L.hip_cal_align_score.restype = None
    L.align_hip.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp, vp]
    L.align_hip.restype = None
    L.init_mapping_table.restype = None
    L.malloc_mem.argtypes = [ctypes.c_uint64]
    L.malloc_mem.restype = vp
    L.free_mem.argtypes = [vp]
    L.free_mem.restype = None
    _lib = L
    return L


def check(rc: int, what: str = "bgsa_hip") -> None:
    if rc != 0:
        raise BgsaHipError(f"{what}: rc={rc}: {lib().bgsa_hip_last_error().decode()}")


class SeqT(ctypes.Structure):
    """seq_t of include/bgsa_hip.h (reference original/BGSA_CPU/global.h:9-16)."""
    _fields_ = [("len", ctypes.c_int), ("size", ctypes.c_int64), ("count", ctypes.c_int64),
                ("extra_size", ctypes.c_int), ("extra_count", ctypes.c_int),
                ("content", ctypes.c_void_p)]


def score_sets() -> list[tuple[int, int, int]]:
    """The (match, mismatch, gap) sets BitPAl kernels were compiled for (Makefile BITPAL_SETS)."""
    out = []
    for i in range(lib().bgsa_hip_score_set_count()):
        m, x, g = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        check(lib().bgsa_hip_score_set(i, ctypes.byref(m), ctypes.byref(x), ctypes.byref(g), None), "score_set")
        out.append((m.value, x.value, g.value))
    return out


def word_num(algo: int, qlen: int, slen: int, k: int = 0) -> int:
    return int(lib().bgsa_hip_word_num(algo, qlen, slen, k))


def group_words(algo: int, wn: int, k: int = 0) -> int:
    return int(lib().bgsa_hip_group_words(algo, wn, k))


def pad_rows(rows: np.ndarray, multiple: int = V_NUM) -> tuple[np.ndarray, int]:
    """Pad the subject set to a multiple of 64 with all-'N' reads, as get_read_from_file does for
    the final bucket (reference original/BGSA_CPU/file.c:98-112).  Returns (rows, extra_count)."""
    rows = np.ascontiguousarray(rows, dtype=np.uint8)
    n, length = rows.shape
    extra = (-n) % multiple
    if extra:
        rows = np.concatenate([rows, np.full((extra, length), ord("N"), dtype=np.uint8)])
    return rows, extra


def rows_to_buffer(rows: np.ndarray) -> np.ndarray:
    """[n, len] ASCII -> the reference's row buffer (len bytes + '\\n' per row), flat uint8."""
    rows = np.ascontiguousarray(rows, dtype=np.uint8)
    n, length = rows.shape
    buf = np.full((n, length + 1), ord("\n"), dtype=np.uint8)
    buf[:, :length] = rows
    return buf.reshape(-1)


# ------------------------------------------------------------------------------------------------
# Device-resident driver (torch tensors hold the HBM buffers)
# ------------------------------------------------------------------------------------------------

class DeviceAligner:
    """One subject bucket resident in HBM, scored against query tiles.

    Mirrors what cal_on_<arch> does per read bucket (reference original/BGSA_CPU/cal_cpu.c:
    252-401): preprocess the bucket once, then loop over query buckets calling the grid.
    """

    def __init__(self, algo: int = ALGO_MYERS, device: str = "cuda:0", k: int = 0, scores=None,
                 semi_global: bool = False):
        """scores: (match, mismatch, gap) for ALGO_BITPAL; None = the reference's 2 / -3 / -5.
        For ALGO_MYERS (0, 1, 1) reports +distance (generator -m 1) instead of -distance.
        semi_global (generator -s): ALGO_BITPAL — query end to end, free subject overhangs;
        ALGO_MYERS — subject end to end inside the query (the generator's orientations differ)."""
        import torch
        self.torch = torch
        self.algo, self.k = algo, int(k)
        self.scores = tuple(int(x) for x in scores) if scores is not None else None
        self.semi_global = bool(semi_global)
        if self.scores is not None and algo != ALGO_BITPAL and not (algo == ALGO_MYERS and self.scores in ((0, 1, 1), (0, -1, -1))):
            raise BgsaHipError("scores apply to ALGO_BITPAL; ALGO_MYERS only knows (0, -1, -1) and (0, 1, 1) = +distance")
        if self.semi_global and algo == ALGO_BANDED:
            raise BgsaHipError("semi_global is not defined for the banded filter")
        self.device = torch.device(device)
        if not torch.cuda.is_available():
            raise BgsaHipError("no GPU visible: the HIP path has no CPU fallback")
        torch.cuda.set_device(self.device)
        check(lib().bgsa_hip_set_device(self.device.index or 0), "set_device")
        self.out_dtype = torch.int8 if algo == ALGO_BANDED else torch.int16

    def _stream(self):
        return ctypes.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def set_queries(self, queries: np.ndarray) -> None:
        """queries: [nq, qlen] uint8 ASCII.  Uploads the row buffer and maps it to 0..4 in place."""
        torch = self.torch
        q = np.ascontiguousarray(queries, dtype=np.uint8)
        self.nq, self.qlen = q.shape
        buf = rows_to_buffer(q)
        # 8 spare bytes: the kernel's scalar dword fetch of the last characters stays in bounds
        self.d_content = torch.zeros(buf.size + 8, dtype=torch.uint8, device=self.device)
        self.d_content[: buf.size].copy_(torch.from_numpy(buf))
        check(lib().bgsa_hip_map_queries_dev(self.d_content.data_ptr(), buf.size, self._stream()), "map_queries")

    def set_subjects(self, subjects: np.ndarray, qlen: int | None = None) -> None:
        """subjects: [ns, slen] uint8 ASCII (padded here to a multiple of 64 with 'N' reads)."""
        torch = self.torch
        s, self.extra = pad_rows(subjects)
        self.ns_real = subjects.shape[0]
        self.ns, self.slen = s.shape
        buf = rows_to_buffer(s)
        d_rows = torch.from_numpy(buf).to(self.device)
        self.set_subject_rows_device(d_rows, self.ns, self.slen, qlen)

    def set_subject_rows_device(self, d_rows, ns: int, slen: int, qlen: int | None = None) -> None:
        """d_rows: uint8 device tensor holding ns rows of slen+1 bytes; ns % 64 == 0."""
        torch = self.torch
        self.ns, self.slen = int(ns), int(slen)
        qlen = self.qlen if qlen is None else qlen
        self.wn = word_num(self.algo, qlen, self.slen, self.k)
        n_words = group_words(self.algo, self.wn, self.k) * (self.ns // V_NUM)
        self.d_peq = torch.empty(n_words, dtype=torch.int32, device=self.device)
        check(lib().bgsa_hip_handle_reads_dev(self.algo, d_rows.data_ptr(), d_rows.numel(), self.slen,
                                              self.ns, self.wn, self.k, self.d_peq.data_ptr(),
                                              self._stream()), "handle_reads_dev")

    def params(self) -> Params:
        """This aligner's own scoring parameters (the *_ex entry points take them explicitly, so two
        aligners with different scores or modes never meet in the C ABI's process-global ints)."""
        if self.algo == ALGO_BITPAL:
            m, x, g = self.scores or (2, -3, -5)
        elif self.algo == ALGO_MYERS and self.scores == (0, 1, 1):
            m, x, g = 0, 1, 1            # generator -m 1: +distance
        else:
            m, x, g = 0, -1, -1
        return Params(self.algo, 1 if self.semi_global else 0, m, x, g, self.k)

    def score(self, ref_start: int = 0, ref_end: int | None = None, out=None):
        """Scores queries [ref_start, ref_end) against the resident bucket -> [nq_tile, ns] tensor."""
        torch = self.torch
        ref_end = self.nq if ref_end is None else ref_end
        if out is None:
            out = torch.empty((ref_end - ref_start, self.ns), dtype=self.out_dtype, device=self.device)
        p = self.params()
        need = int(lib().bgsa_hip_workspace_bytes_ex(ctypes.byref(p), self.qlen, self.slen, ref_end - ref_start))
        if getattr(self, "d_work", None) is None or self.d_work.numel() < need:
            self.d_work = torch.empty(max(need, 8), dtype=torch.uint8, device=self.device)
        check(lib().bgsa_hip_cal_align_score_ex(ctypes.byref(p), self.d_content.data_ptr(), self.d_peq.data_ptr(),
                                                out.data_ptr(), self.qlen, self.slen, self.ns, ref_start,
                                                ref_end, self.wn, self.d_work.data_ptr(),
                                                self.d_work.numel(), self._stream()), "cal_align_score_ex")
        return out

    def check_faults(self) -> None:
        """Synchronises and raises if a kernel reported a stream fault (bgsa_hip_stream_faults)."""
        self.torch.cuda.synchronize(self.device)
        flags = int(lib().bgsa_hip_stream_faults(1))
        if flags:
            raise BgsaHipError(f"stream fault: {lib().bgsa_hip_last_error().decode()}")

    def _select(self) -> None:
        # the process-global selection of the C ABI (the reference's ints), for the entry points that read it
        if self.algo == ALGO_BITPAL:
            check(lib().bgsa_hip_select_scores(*(self.scores or (2, -3, -5))), "select_scores")
        elif self.algo == ALGO_MYERS and self.scores == (0, 1, 1):
            check(lib().bgsa_hip_select_scores(0, 1, 1), "select_scores")       # generator -m 1: +distance
        else:   # also resets the score ints a BitPAl / +distance selection left behind
            check(lib().bgsa_hip_select_algorithm(self.algo), "select_algorithm")
        check(lib().bgsa_hip_select_alignment(1 if self.semi_global else 0), "select_alignment")

    def kernel_name(self) -> str:
        self._select()
        name = lib().bgsa_hip_kernel_name(self.algo, self.wn).decode()
        lib().bgsa_hip_select_alignment(0)
        return name


def align_all_pairs(queries: np.ndarray, subjects: np.ndarray, algo: int = ALGO_MYERS, k: int = 0,
                    device: str = "cuda:0", scores=None, semi_global: bool = False) -> np.ndarray:
    """Convenience: scores[nq, ns] for small inputs, through the device-resident C ABI."""
    a = DeviceAligner(algo, device, k, scores, semi_global)
    a.set_queries(queries)
    a.set_subjects(subjects)
    out = a.score()
    a.check_faults()
    return out[:, : a.ns_real].cpu().numpy()
