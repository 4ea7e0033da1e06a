"""Post-compilation check of the generated-asm kernels (scripts/check_asm_kernels.py): cross-compiles the
translation units to gfx950 assembly and verifies what the compiler did around the inline-asm row loops —
operands outside the loops' hard-coded SGPRs, no sign-extended 64-bit scalar (round 1's fault signature,
DESIGN.md §8), the fault-word report present.  Needs hipcc, no GPU."""
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def test_asm_kernels_pass_the_disassembly_check():
    p = subprocess.run([sys.executable, str(ROOT / "scripts" / "check_asm_kernels.py")], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    assert "0 problem(s)" in p.stdout


def test_the_check_catches_the_round_1_fault_pattern():
    sys.path.insert(0, str(ROOT / "scripts"))
    import check_asm_kernels as C
    bad = "\n".join([
        "_ZN4bgsa23myers_global_asm_kernelILi5ELi1EEEvx:",
        "\ts_bfe_i64 s[18:19], s[4:5], 0x200000",       # what 2a37ac0 compiled to: sext(lo) | hi << 32
        "\t;;#ASMSTART", "\ts_mov_b64 s[70:71], s[68:69]", "\tL_anchor_4:", "\tL_done_4:", "\ts_mov_b32 s66, s69", "\t;;#ASMEND",
        "\ts_endpgm", ""])
    n, problems = C.check(bad, "synthetic")
    assert n == 1 and len(problems) == 4      # s_bfe_i64, operand in / out of a hard-coded register, no fault report
