"""The `aligner` / `convert` command-line programs (bgsa_amd/host, plain C on the C ABI) against the
golden fixtures: same files in, same `convert -r` text out as the reference's own binaries."""
import os
import subprocess
from pathlib import Path

import numpy as np
import pytest

import bgsa_amd as B
from conftest import load_golden

pytestmark = pytest.mark.gpu

HOST = Path(B.__file__).resolve().parent / "host"
ROOT = HOST.parent.parent
ALGO_FLAG = {"original_cpu": "myers", "original_avx2": "bitpal", "banded_cpu": "banded"}


def _run_cli(tmp_path, g, bucket_bytes=None, converter=None, extra_args=(), env_extra=None):
    (tmp_path / "query.txt").write_bytes(B.rows_to_buffer(g["queries"]).tobytes())
    (tmp_path / "subject.txt").write_bytes(B.rows_to_buffer(g["subjects"]).tobytes())
    cmd = [str(HOST / "aligner"), "-q", "query.txt", "-d", "subject.txt", "-f", "result.txt",
           "-a", ALGO_FLAG[g["variant"]]]
    if g["k"] >= 0:
        cmd += ["-k", str(g["k"])]
    cmd += list(extra_args)
    env = dict(os.environ, **(env_extra or {}))
    if bucket_bytes:
        env["BGSA_READ_BUCKET_SIZE"] = str(bucket_bytes)
    p = subprocess.run(cmd, cwd=tmp_path, env=env, capture_output=True, text=True)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "cal GCUPS is" in p.stdout and "Total GCUPS is" in p.stdout
    conv = converter or (HOST / "convert")
    subprocess.run([str(conv), "-r", "result.txt", "-o", "scores.txt"], cwd=tmp_path, check=True, capture_output=True)
    flat = np.loadtxt(tmp_path / "scores.txt", dtype=np.int64, ndmin=1)
    return flat.reshape(g["queries"].shape[0], g["subjects"].shape[0]), p.stdout


@pytest.mark.parametrize("name", ["f1_myers_150", "f6_myers_ns100", "f2_myers_1000", "f7_bitpal_150", "f8_banded_k8_150",
                                  "f9_myers_140x150"])
def test_cli_matches_golden(tmp_path, name):
    g = load_golden(name)
    got, _ = _run_cli(tmp_path, g)
    assert np.array_equal(got, g["scores"])


def test_cli_many_read_buckets_and_query_blocks(tmp_path, oracle):
    # 5 read buckets (the last one padded with 'N' reads) x 3 query blocks of REF_BUCKET_COUNT
    q = oracle.gen_reads(91, 230, 150)
    s = oracle.gen_reads(92, 300, 150)
    g = {"queries": q, "subjects": s, "variant": "original_cpu", "k": -1}
    got, report = _run_cli(tmp_path, g, bucket_bytes=64 * 151 + 10)
    assert np.array_equal(got, oracle.myers64(q, s))
    assert "subject_count is 320" in report  # 300 reads + 20 padding reads, as the reference counts them
    info = np.fromfile(tmp_path / "result.txt.info", dtype=np.uint8)
    assert int(np.frombuffer(info[:4].tobytes(), dtype=np.int32)[0]) == 5


@pytest.mark.parametrize("resident", ["1", "0"])
@pytest.mark.parametrize("launch_blocks,devices,mode", [("1", "0", None), ("4", "0", None), ("16", "0", None), ("4", "0,0,0", None),
                                                        ("3", "0,0", "mmap"), ("4", "0", "mmap+populate"), ("2", "0,0", "mmap+falloc")])
def test_cli_launch_blocks_and_writer_modes_give_the_same_file(tmp_path, oracle, launch_blocks, devices, mode, resident):
    """A launch scores BGSA_LAUNCH_BLOCKS query blocks of REF_BUCKET_COUNT at once; the file must not notice: with one device
    the blocks of a bucket are one contiguous piece, with several every block is still device 0's tile, device 1's, ...
    (cal_mic.c:535-536).  470 queries = four full blocks and a ragged fifth, two read buckets, the last one padded.  The same
    with the alternative writers (the mapped file; measured slower on tmpfs, kept as BGSA_WRITER_MODE), and with both pipelines:
    the bucket's scores resident in HBM with the copy-out draining behind the kernels (the default), or two launches in flight
    through the ring (BGSA_RESULT_RESIDENT=0)."""
    q = oracle.gen_reads(191, 470, 150)
    s = oracle.gen_reads(192, 700, 150)
    g = {"queries": q, "subjects": s, "variant": "original_cpu", "k": -1}
    env = {"BGSA_LAUNCH_BLOCKS": launch_blocks, "BGSA_RESULT_RESIDENT": resident}
    if mode:
        env["BGSA_WRITER_MODE"] = mode
    got, report = _run_cli(tmp_path, g, bucket_bytes=448 * 151 + 10, extra_args=["-g", devices], env_extra=env)
    assert np.array_equal(got, oracle.myers64(q, s))
    assert "pipeline_busy_time" in report and "cal_total_times" in report


@pytest.mark.parametrize("env,want", [({"BGSA_RESULT_RESIDENT_GB": "0.000001"}, "ring"),
                                      ({"BGSA_RESULT_RESIDENT_GB": "100000", "BGSA_DEVICE_FREE_GB": "0.5"}, "ring"),
                                      ({"BGSA_RESULT_RESIDENT_GB": "100000"}, "resident"),
                                      ({"BGSA_RESULT_RESIDENT_GB": "100000", "BGSA_DEVICE_FREE_GB": "0.5", "GPUS": "0,0"}, "ring")])
def test_cli_resident_results_only_when_they_fit_the_device(tmp_path, oracle, env, want):
    """Resident results are the default — when the bucket's scores fit BGSA_RESULT_RESIDENT_GB AND the memory the device
    really has free (hipMemGetInfo, with headroom): a limit set too high for the card must end in the ring pipeline with the
    same file, not in a failed allocation.  BGSA_DEVICE_FREE_GB stands in for a small or busy card."""
    q = oracle.gen_reads(195, 230, 150)
    s = oracle.gen_reads(196, 700, 150)
    g = {"queries": q, "subjects": s, "variant": "original_cpu", "k": -1}
    env = dict(env)
    gpus = env.pop("GPUS", "0")
    got, report = _run_cli(tmp_path, g, bucket_bytes=448 * 151 + 10, extra_args=["-g", gpus], env_extra=env)
    assert np.array_equal(got, oracle.myers64(q, s))
    assert f"result_pipeline     is {want}" in report


def test_reference_convert_reads_our_result_files(tmp_path):
    # the result / .info pair is the reference's format: its own `convert -r` must decode it
    ref_convert = ROOT / "oracle" / "_ref" / "original_cpu" / "convert"
    if not ref_convert.exists():
        pytest.skip("reference binaries not built (oracle/_ref)")
    g = load_golden("f1_myers_150")
    got, _ = _run_cli(tmp_path, g, bucket_bytes=128 * 151, converter=ref_convert)
    assert np.array_equal(got, g["scores"])


# ---- several GPUs from one process (-n / -g list / -R): this box has one card, so the same card is
# listed more than once — every listed entry gets its own slice, buffers and streams, which is the
# whole multi-device code path ------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["f1_myers_150", "f7_bitpal_150", "f8_banded_k8_150"])
def test_cli_two_devices_matches_golden(tmp_path, name):
    g = load_golden(name)
    got, report = _run_cli(tmp_path, g, extra_args=["-g", "0,0"])
    assert np.array_equal(got, g["scores"])
    assert "gpu_count     is 2" in report
    info = (tmp_path / "result.txt.info").read_bytes()
    assert np.frombuffer(info[4:8], dtype=np.int32)[0] == 2


def test_cli_three_devices_ratios_buckets_and_blocks(tmp_path, oracle):
    # 3 devices with ratios 1 : 2.5 : 0.5, 3 read buckets (the last padded), 3 query blocks
    q = oracle.gen_reads(93, 230, 150)
    s = oracle.gen_reads(94, 1000, 150)
    (tmp_path / "ratio.txt").write_text("1\n2.5\n0.5\n")
    g = {"queries": q, "subjects": s, "variant": "original_cpu", "k": -1}
    got, report = _run_cli(tmp_path, g, bucket_bytes=384 * 151 + 10, extra_args=["-g", "0,0,0", "-R", "ratio.txt"])
    assert np.array_equal(got, oracle.myers64(q, s))
    info = (tmp_path / "result.txt.info").read_bytes()
    n_buckets, n_dev = np.frombuffer(info[:8], dtype=np.int32)
    assert (n_buckets, n_dev) == (3, 3)
    rec = 8 * 3 + 4
    counts = [np.frombuffer(info[16 + b * rec: 16 + b * rec + 24], dtype=np.int64) for b in range(3)]
    assert [int(c.sum()) for c in counts] == [384, 384, 256]            # 232 reads + 24 padding reads
    assert all(int(x) % 64 == 0 for c in counts for x in c)
    assert counts[0][1] > counts[0][0] > counts[0][2]                    # follows the ratios
    assert np.frombuffer(info[16 + 2 * rec + 24: 16 + 3 * rec], dtype=np.int32)[0] == 24


def test_cli_dynamic_ratios_follow_device_times(tmp_path, oracle):
    """-D (the KNC backend's dynamic mode, adjust_device_ratio3, BGSA_KNC/global.c:120-168): three device
    entries, the middle one made four times slower (it scores every block four times), six read buckets:
    after the first bucket its slice shrinks to about a quarter, the scores do not change."""
    q = oracle.gen_reads(193, 150, 150)
    s = oracle.gen_reads(194, 6 * 3840, 150)
    g = {"queries": q, "subjects": s, "variant": "original_cpu", "k": -1}
    got, report = _run_cli(tmp_path, g, bucket_bytes=3840 * 151 + 10, extra_args=["-g", "0,0,0", "-D"],
                           env_extra={"BGSA_DEBUG_DEVICE_DELAY": "0,3,0"})
    assert np.array_equal(got, oracle.myers64(q, s))
    assert "-> ratios" in report
    info = (tmp_path / "result.txt.info").read_bytes()
    n_buckets, n_dev = np.frombuffer(info[:8], dtype=np.int32)
    assert (n_buckets, n_dev) == (6, 3)
    rec = 8 * 3 + 4
    counts = np.array([np.frombuffer(info[16 + b * rec: 16 + b * rec + 24], dtype=np.int64) for b in range(6)])
    assert (counts.sum(axis=1) == 3840).all() and (counts % 64 == 0).all()
    assert counts[0][0] == counts[0][1] == counts[0][2]                   # bucket 0: the initial equal split
    for b in range(2, 6):                                                  # from then on the slow device gets far less
        assert counts[b][1] < 0.6 * counts[b][0] and counts[b][1] < 0.6 * counts[b][2], counts
    # without -D nothing moves
    got2, report2 = _run_cli(tmp_path, g, bucket_bytes=3840 * 151 + 10, extra_args=["-g", "0,0,0"],
                             env_extra={"BGSA_DEBUG_DEVICE_DELAY": "0,3,0"})
    assert np.array_equal(got2, got) and "-> ratios" not in report2


def test_cli_more_devices_than_groups(tmp_path, oracle):
    # 100 subjects = 2 groups on 3 devices: one device gets nothing and the files still decode
    q = oracle.gen_reads(95, 7, 150)
    s = oracle.gen_reads(96, 100, 150)
    g = {"queries": q, "subjects": s, "variant": "original_cpu", "k": -1}
    got, _ = _run_cli(tmp_path, g, extra_args=["-g", "0,0,0"])
    assert np.array_equal(got, oracle.myers64(q, s))


def test_reference_knc_convert_reads_multi_device_result(tmp_path):
    ref_convert = ROOT / "oracle" / "_ref" / "original_knc" / "convert"
    if not ref_convert.exists():
        pytest.skip("reference KNC converter not built (oracle/_ref)")
    g = load_golden("f6_myers_ns100")
    got, _ = _run_cli(tmp_path, g, converter=ref_convert, extra_args=["-n", "2", "-g", "0,0"])
    assert np.array_equal(got, g["scores"])


def test_cli_rejects_missing_gpu(tmp_path):
    g = load_golden("f1_myers_150")
    (tmp_path / "query.txt").write_bytes(B.rows_to_buffer(g["queries"]).tobytes())
    (tmp_path / "subject.txt").write_bytes(B.rows_to_buffer(g["subjects"]).tobytes())
    p = subprocess.run([str(HOST / "aligner"), "-q", "query.txt", "-d", "subject.txt", "-g", "63"],
                       cwd=tmp_path, capture_output=True, text=True)
    assert p.returncode != 0 and "does not exist" in p.stdout


def test_cli_bitpal_with_other_scores(tmp_path, oracle):
    sets = [x for x in B.score_sets() if x != (2, -3, -5)]
    if not sets:
        pytest.skip("only the default score set is compiled in")
    m, x, g_ = sets[-1]
    q = oracle.gen_reads(97, 9, 150)
    s = oracle.gen_reads(98, 200, 150)
    s[:9] = oracle.mutate(q, np.arange(9) * 2, 99)
    g = {"queries": q, "subjects": s, "variant": "original_avx2", "k": -1}
    got, report = _run_cli(tmp_path, g, extra_args=["-M", str(m), "-I", str(x), "-G", str(g_)])
    assert np.array_equal(got, oracle.dp_nw(q, s, m, x, g_))
    assert f"score is {m}, {x}, {g_}" in report


def test_cli_refuses_uncompiled_scores(tmp_path):
    g = load_golden("f1_myers_150")
    (tmp_path / "query.txt").write_bytes(B.rows_to_buffer(g["queries"]).tobytes())
    (tmp_path / "subject.txt").write_bytes(B.rows_to_buffer(g["subjects"]).tobytes())
    p = subprocess.run([str(HOST / "aligner"), "-q", "query.txt", "-d", "subject.txt", "-M", "9", "-I", "-9", "-G", "-9"],
                       cwd=tmp_path, capture_output=True, text=True)
    assert p.returncode != 0 and "BITPAL_SETS" in p.stdout


def test_cli_semiglobal(tmp_path, oracle):
    q = oracle.gen_reads(61, 5, 60)
    s = oracle.gen_reads(62, 100, 150)
    for r in range(10):
        s[r, 9 * r: 9 * r + 60] = q[r % 5]
    g = {"queries": q, "subjects": s, "variant": "original_avx2", "k": -1}
    got, _ = _run_cli(tmp_path, g, extra_args=["-s"])     # variant original_avx2 -> -a bitpal
    assert np.array_equal(got, oracle.dp_semiglobal(q, s))
    assert (got[np.arange(10) % 5, np.arange(10)] == 120).all()     # exact copies: 60 matches x 2


def test_cli_semiglobal_myers(tmp_path, oracle):
    q = oracle.gen_reads(63, 5, 200)
    s = oracle.gen_reads(64, 100, 60)
    for r in range(10):
        s[r] = q[r % 5, 11 * r: 11 * r + 60]
    g = {"queries": q, "subjects": s, "variant": "original_cpu", "k": -1}
    got, _ = _run_cli(tmp_path, g, extra_args=["-s"])
    assert np.array_equal(got, oracle.dp_edit_semiglobal(q, s))
    assert (got[np.arange(10) % 5, np.arange(10)] == 0).all()     # exact windows of the query


# ---- the drop-in boundary end to end: the REFERENCE's own main.c / file.c / thread.c / cal_cpu.c,
# unmodified, compiled against libbgsa_hip.so in place of its global.c + align_core.c
# (oracle/Makefile: _ref/original_hip, examples/BGSA_HIP/config_hip.h) -------------------------------------
REF_HIP = ROOT / "oracle" / "_ref" / "original_hip"


def _run_reference_host(tmp_path, g, binary, threads=4, env_extra=None):
    (tmp_path / "query.txt").write_bytes(B.rows_to_buffer(g["queries"]).tobytes())
    (tmp_path / "subject.txt").write_bytes(B.rows_to_buffer(g["subjects"]).tobytes())
    env = dict(os.environ, **(env_extra or {}))
    p = subprocess.run([str(REF_HIP / binary), "-q", "query.txt", "-d", "subject.txt", "-f", "result.txt", "-N", str(threads)],
                       cwd=tmp_path, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    conv = ROOT / "oracle" / "_ref" / "original_cpu" / "convert"
    subprocess.run([str(conv), "-r", "result.txt", "-o", "scores.txt"], cwd=tmp_path, check=True, capture_output=True)
    flat = np.loadtxt(tmp_path / "scores.txt", dtype=np.int64, ndmin=1)
    return flat.reshape(g["queries"].shape[0], g["subjects"].shape[0]), p.stdout


@pytest.mark.parametrize("name,binary", [("f1_myers_150", "aligner"), ("f6_myers_ns100", "aligner"),
                                         ("f9_myers_140x150", "aligner"), ("f7_bitpal_150", "aligner_bitpal")])
def test_reference_host_pipeline_on_the_gpu_library(tmp_path, name, binary):
    if not (REF_HIP / binary).exists():
        pytest.skip("oracle/_ref/original_hip not built (needs /root/reference at build time)")
    g = load_golden(name)
    got, report = _run_reference_host(tmp_path, g, binary)
    assert np.array_equal(got, g["scores"])          # the reference's scores, through its own host code
    assert "GCUPS" in report


@pytest.mark.parametrize("binary", ["aligner_block64", "aligner_block250"])
@pytest.mark.parametrize("threads,ahead", [(8, None), (3, "7"), (16, "256")])
def test_reference_host_pipeline_other_query_blocks(tmp_path, oracle, binary, threads, ahead):
    """The reference's host files built with REF_BUCKET_COUNT = 64 and 250 instead of 100 (oracle/Makefile): the align_hip seam's
    read-ahead (100 rows by default, the reference's own block) then no longer lines up with the caller's blocks, and with 250
    the OpenMP threads start beyond the rows read ahead.  That may cost time, never results: 700 queries = several blocks of
    either size with a ragged last one."""
    if not (REF_HIP / binary).exists():
        pytest.skip("oracle/_ref/original_hip not built (needs /root/reference at build time)")
    q = oracle.gen_reads(171, 700, 150)
    s = oracle.gen_reads(172, 64 * 9 + 5, 150)
    s[:40] = oracle.mutate(q[np.arange(40) * 17 % 700], np.arange(40) % 11, 173)
    g = {"queries": q, "subjects": s}
    got, report = _run_reference_host(tmp_path, g, binary, threads=threads, env_extra={"BGSA_HIP_ROW_AHEAD": ahead} if ahead else None)
    assert np.array_equal(got, oracle.myers64(q, s))
    assert "GCUPS" in report


# ---- the COARSE seam as a drop-in ("where a device launch belongs"): the reference's main.c /
# file.c / thread.c unmodified and its cal_cpu.c minus the definition of cpu_cal_align_score (examples/BGSA_HIP/derive_cal_hip.py,
# applied at build time to the reference file where it lies) — cpu_cal's call lands in the library's hip_cal_align_score: one device
# launch per block of REF_BUCKET_COUNT queries, no align_hip, no row cache (oracle/Makefile: _ref/original_hip_coarse) --------------
REF_HIP_COARSE = ROOT / "oracle" / "_ref" / "original_hip_coarse"


def _run_coarse_host(tmp_path, g, binary, threads=4, env_extra=None):
    (tmp_path / "query.txt").write_bytes(B.rows_to_buffer(g["queries"]).tobytes())
    (tmp_path / "subject.txt").write_bytes(B.rows_to_buffer(g["subjects"]).tobytes())
    env = dict(os.environ, BGSA_HIP_SEAM_STATS="1", **(env_extra or {}))
    p = subprocess.run([str(REF_HIP_COARSE / binary), "-q", "query.txt", "-d", "subject.txt", "-f", "result.txt", "-N", str(threads)],
                       cwd=tmp_path, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    conv = ROOT / "oracle" / "_ref" / "original_cpu" / "convert"
    subprocess.run([str(conv), "-r", "result.txt", "-o", "scores.txt"], cwd=tmp_path, check=True, capture_output=True)
    flat = np.loadtxt(tmp_path / "scores.txt", dtype=np.int64, ndmin=1)
    return flat.reshape(g["queries"].shape[0], g["subjects"].shape[0]), p.stdout + p.stderr


def test_coarse_seam_binary_binds_the_grid_not_the_row():
    """What the two drop-ins import from the library: the fine one align_hip (57,900 calls per 100-query block of 1M subjects,
    served by the row cache), the coarse one hip_cal_align_score and nothing of the fine seam."""
    if not (REF_HIP_COARSE / "aligner").exists():
        pytest.skip("oracle/_ref/original_hip_coarse not built (needs /root/reference at build time)")
    undefined = lambda path: {ln.split()[-1] for ln in subprocess.run(["nm", "-D", "--undefined-only", str(path)], capture_output=True,   # noqa: E731
                                                                      text=True, check=True).stdout.splitlines() if ln.strip()}
    coarse, fine = undefined(REF_HIP_COARSE / "aligner"), undefined(REF_HIP / "aligner")
    assert {"hip_cal_align_score", "hip_handle_reads", "malloc_mem", "init_mapping_table"} <= coarse and "align_hip" not in coarse
    assert {"align_hip", "hip_handle_reads"} <= fine and "hip_cal_align_score" not in fine


@pytest.mark.parametrize("name,binary", [("f1_myers_150", "aligner"), ("f6_myers_ns100", "aligner"), ("f2_myers_1000", "aligner"),
                                         ("f9_myers_140x150", "aligner"), ("f7_bitpal_150", "aligner_bitpal")])
def test_reference_host_pipeline_through_the_coarse_seam(tmp_path, name, binary):
    if not (REF_HIP_COARSE / binary).exists():
        pytest.skip("oracle/_ref/original_hip_coarse not built (needs /root/reference at build time)")
    g = load_golden(name)
    got, report = _run_coarse_host(tmp_path, g, binary)
    assert np.array_equal(got, g["scores"])          # the reference's scores, through its own host code, one launch per query block
    assert "GCUPS" in report


def test_coarse_and_fine_seam_binaries_agree_on_1k_x_1k(tmp_path, oracle):
    """BASELINE configs[0]'s shape (1k x 1k x 150 bp) through both drop-ins: the same result file, and the coarse one gets there
    with ten library calls (ten blocks of REF_BUCKET_COUNT = 100 queries) where the fine one makes thousands."""
    if not (REF_HIP_COARSE / "aligner").exists() or not (REF_HIP / "aligner").exists():
        pytest.skip("oracle/_ref drop-ins not built (need /root/reference at build time)")
    q = oracle.gen_reads(0xB65A0001, 1000, 150)
    s = oracle.gen_reads(0xB65A1001, 1000, 150)
    s[:50] = oracle.mutate(q[np.arange(50) * 19 % 1000], np.arange(50) % 13, 7)
    g = {"queries": q, "subjects": s}
    (tmp_path / "c").mkdir()
    (tmp_path / "f").mkdir()
    coarse, rep_c = _run_coarse_host(tmp_path / "c", g, "aligner", threads=8)
    fine, _ = _run_reference_host(tmp_path / "f", g, "aligner", threads=8)
    assert np.array_equal(coarse, fine) and np.array_equal(coarse, oracle.myers64(q, s))
    assert (tmp_path / "c" / "result.txt").read_bytes() == (tmp_path / "f" / "result.txt").read_bytes()
    import re
    m = re.search(r"seam calls (\d+)", rep_c) or re.search(r"calls[ =:]+(\d+)", rep_c)
    if m:                                   # BGSA_HIP_SEAM_STATS=1: the library's own count of scoring calls
        assert int(m.group(1)) == 10, rep_c[-600:]


def test_reference_host_pipeline_other_scores(tmp_path, oracle):
    sets = [x for x in B.score_sets() if x != (2, -3, -5)]
    if not (REF_HIP / "aligner_bitpal").exists() or not sets:
        pytest.skip("oracle/_ref/original_hip not built or no extra score set")
    m, x, gp = sets[0]
    q = oracle.gen_reads(71, 6, 150)
    s = oracle.gen_reads(72, 200, 150)
    s[:6] = oracle.mutate(q, np.arange(6), 73)
    g = {"queries": q, "subjects": s}
    got, report = _run_reference_host(tmp_path, g, "aligner_bitpal", env_extra={"BGSA_HIP_SCORES": f"{m},{x},{gp}"})
    assert np.array_equal(got, oracle.dp_nw(q, s, m, x, gp))
    assert f"score is {m}, {x}, {gp}" in report


# ---- the banded drop-in: banded/BGSA_CPU's OWN main.c (-k) / file.c / thread.c / cal_cpu.c — its word_num
# formula (cal_cpu.c:253-254), its 64-bit cpu_read_t, its int8 results — compiled unmodified against the
# library (oracle/Makefile: _ref/banded_hip, examples/BGSA_HIP/config_banded_hip.h) -------------------------
BANDED_HIP = ROOT / "oracle" / "_ref" / "banded_hip" / "aligner"


def _run_banded_host(tmp_path, g, k, threads=4):
    (tmp_path / "query.txt").write_bytes(B.rows_to_buffer(g["queries"]).tobytes())
    (tmp_path / "subject.txt").write_bytes(B.rows_to_buffer(g["subjects"]).tobytes())
    cmd = [str(BANDED_HIP), "-q", "query.txt", "-d", "subject.txt", "-f", "result.txt", "-N", str(threads)]
    if k is not None:
        cmd += ["-k", str(k)]
    p = subprocess.run(cmd, cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    conv = ROOT / "oracle" / "_ref" / "banded_cpu" / "convert"       # the reference's own int8 decoder
    subprocess.run([str(conv), "-r", "result.txt", "-o", "scores.txt"], cwd=tmp_path, check=True, capture_output=True)
    flat = np.loadtxt(tmp_path / "scores.txt", dtype=np.int64, ndmin=1)
    return flat.reshape(g["queries"].shape[0], g["subjects"].shape[0]), p.stdout


@pytest.mark.parametrize("name", ["f8_banded_k8_150", "f8_banded_k4_150", "f8_banded_k16_150", "f8_banded_k8_len500",
                                  "f8_banded_k8_len73"])
def test_banded_reference_host_pipeline_on_the_gpu_library(tmp_path, name):
    if not BANDED_HIP.exists():
        pytest.skip("oracle/_ref/banded_hip not built (needs /root/reference at build time)")
    g = load_golden(name)
    got, report = _run_banded_host(tmp_path, g, g["k"])
    assert np.array_equal(got, g["scores"])          # the reference's scores, through its own banded host code
    assert "GCUPS" in report


BANDED_HIP_COARSE = ROOT / "oracle" / "_ref" / "banded_hip_coarse" / "aligner"


@pytest.mark.parametrize("name", ["f8_banded_k8_150", "f8_banded_k4_150", "f8_banded_k16_150", "f8_banded_k8_len500"])
def test_banded_reference_host_pipeline_through_the_coarse_seam(tmp_path, name, monkeypatch):
    """banded/BGSA_CPU's own main.c (-k) / file.c / thread.c and its cal_cpu.c minus the grid function (derive_cal_hip.py): cpu_cal
    lands in hip_cal_align_score with the reference's word_num convention, 64-bit cpu_read_t and int8 results — one launch per block."""
    if not BANDED_HIP_COARSE.exists():
        pytest.skip("oracle/_ref/banded_hip_coarse not built (needs /root/reference at build time)")
    import sys
    monkeypatch.setattr(sys.modules[__name__], "BANDED_HIP", BANDED_HIP_COARSE)     # _run_banded_host starts whatever BANDED_HIP names
    g = load_golden(name)
    got, report = _run_banded_host(tmp_path, g, g["k"])
    assert np.array_equal(got, g["scores"])
    assert "GCUPS" in report
    undefined = {ln.split()[-1] for ln in subprocess.run(["nm", "-D", "--undefined-only", str(BANDED_HIP_COARSE)], capture_output=True,
                                                         text=True, check=True).stdout.splitlines() if ln.strip()}
    assert "hip_cal_align_score" in undefined and "align_hip" not in undefined


def test_banded_reference_host_default_threshold(tmp_path, oracle):
    # no -k: banded/BGSA_CPU/main.c:43 sets threshold = CPU_WORD_SIZE / 2 - 1 = 31 (the 64-bit band kernel);
    # the reference's word_num for k = 31 is smaller than the device layout, so the seam re-pitches it
    if not BANDED_HIP.exists():
        pytest.skip("oracle/_ref/banded_hip not built")
    q = oracle.gen_reads(171, 7, 150)
    s = oracle.gen_reads(172, 200, 150)
    s[:60] = oracle.mutate(q[np.arange(60) % 7], np.arange(60) % 40, 173)
    got, _ = _run_banded_host(tmp_path, {"queries": q, "subjects": s}, None)
    want = oracle.banded64(q, s, 31)
    assert np.array_equal(got, want) and (want != 127).any() and (want == 127).any()


def test_readme_demo_on_the_hip_backend(oracle):
    """The reference README's "use the kernel alignment method" demo (README.md:94-165: AAAA against AAAA, AACA, CAAC,
    AGGG) written against this library (examples/demo/demo_hip.c): built with plain gcc, run on the GPU."""
    demo = ROOT / "examples" / "demo"
    subprocess.run(["make", "-C", str(demo)], check=True, capture_output=True)
    q = np.frombuffer(b"AAAA", dtype=np.uint8).reshape(1, 4)
    s = np.frombuffer(b"AAAAAACACAACAGGG", dtype=np.uint8).reshape(4, 4)
    p = subprocess.run([str(demo / "demo_hip")], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    assert [int(x) for x in p.stdout.split()] == [0, -1, -2, -3] == oracle.myers64(q, s)[0].tolist()
    p = subprocess.run([str(demo / "demo_hip"), "bitpal"], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    assert [int(x) for x in p.stdout.split()] == oracle.bitpal(q, s)[0].tolist() == oracle.dp_nw(q, s, 2, -3, -5)[0].tolist()
