"""The committed generated headers are the generator's output: bgsa_amd/csrc/{myers,bitpal,banded}_rows_gen.inc are
committed so that the exact instruction stream that ships can be read without running anything — this regenerates them
into a temporary directory and compares byte for byte, so a change of rows_ir.py / gen_rows_asm.py cannot ship with
stale headers (and nobody edits a generated file by hand)."""
import subprocess
import sys
from pathlib import Path

CSRC = Path(__file__).resolve().parent.parent / "bgsa_amd" / "csrc"


def test_committed_headers_equal_the_generators_output(tmp_path):
    p = subprocess.run([sys.executable, str(CSRC / "gen_rows_asm.py"), "--out", str(tmp_path)], capture_output=True, text=True, timeout=900,
                       env={k: v for k, v in __import__("os").environ.items() if not k.startswith("BGSA_GEN_")})
    assert p.returncode == 0, p.stderr[-2000:]
    for name in ("myers_rows_gen.inc", "bitpal_rows_gen.inc", "banded_rows_gen.inc"):
        fresh, committed = (tmp_path / name).read_bytes(), (CSRC / name).read_bytes()
        assert fresh == committed, f"{name}: the committed file is not what gen_rows_asm.py writes (run `make -C bgsa_amd/csrc`)"
