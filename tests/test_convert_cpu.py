"""`convert` (bgsa_amd/host/convert.c, plain C, no GPU): FASTA / FASTQ -> one sequence per line, the
input format of `aligner` (reference original/BGSA_CPU/convert.c:19-165), compared with the reference's
own converter when oracle/_ref is built, and -r on a hand-made result/.info pair."""
import subprocess
from pathlib import Path

import numpy as np
import pytest

import bgsa_amd as B

HOST = Path(B.__file__).resolve().parent / "host"
ROOT = HOST.parent.parent
REF_CONVERT = ROOT / "oracle" / "_ref" / "original_cpu" / "convert"

FASTA = ">r1 some description\nACGTAC\nGGTT\n>r2\nTTTTAC\nGGAA\n>r3\nACGTACGGTA\n"
FASTQ = "@r1\nACGTACGGTT\n+\nIIIIIIIIII\n@r2 x\nTTTTACGGAA\n+r2 x\n!!!!!!!!!!\n"


@pytest.fixture(scope="module", autouse=True)
def _built():
    if not (HOST / "convert").exists():
        B.build_library()
    assert (HOST / "convert").exists()


def _convert(binary, flag, src, tmp_path, name):
    out = tmp_path / name
    subprocess.run([str(binary), flag, str(src), "-o", str(out)], check=True, capture_output=True, cwd=tmp_path)
    return out.read_text()


@pytest.mark.parametrize("flag,text,want", [("-f", FASTA, "ACGTACGGTT\nTTTTACGGAA\nACGTACGGTA\n"),
                                            ("-q", FASTQ, "ACGTACGGTT\nTTTTACGGAA\n")])
def test_sequence_files_become_one_read_per_line(tmp_path, flag, text, want):
    src = tmp_path / "in.txt"
    src.write_text(text)
    assert _convert(HOST / "convert", flag, src, tmp_path, "mine.txt") == want
    if REF_CONVERT.exists():
        assert _convert(REF_CONVERT, flag, src, tmp_path, "ref.txt") == want


def test_result_decoding_drops_padding_and_orders_by_query(tmp_path):
    # two read buckets (64 reads, then 64 of which 60 are padding), 130 queries = two query blocks of 100 + 30
    nq, counts, extra = 130, [64, 64], [0, 60]
    rng = np.random.default_rng(5)
    full = [rng.integers(-150, 1, (nq, c)).astype(np.int16) for c in counts]
    with open(tmp_path / "result.txt", "wb") as f:
        for b in range(2):
            for q0 in range(0, nq, 100):
                f.write(full[b][q0:q0 + 100].tobytes())
    with open(tmp_path / "result.txt.info", "wb") as f:
        f.write(np.array([2, 1], dtype=np.int32).tobytes())
        f.write(np.array([nq], dtype=np.int64).tobytes())
        for b in range(2):
            f.write(np.array([counts[b]], dtype=np.int64).tobytes())
            f.write(np.array([extra[b]], dtype=np.int32).tobytes())
    want = np.concatenate([full[0], full[1][:, :4]], axis=1).reshape(-1)
    for binary in [HOST / "convert"] + ([REF_CONVERT] if REF_CONVERT.exists() else []):
        subprocess.run([str(binary), "-r", "result.txt", "-o", "scores.txt"], check=True, capture_output=True, cwd=tmp_path)
        assert np.array_equal(np.loadtxt(tmp_path / "scores.txt", dtype=np.int64), want)
