#!/usr/bin/env python3
"""Post-compilation check of the kernels that carry generated inline-asm row loops.

The row loops (bgsa_amd/csrc/gen_rows_asm.py) use hard-coded scalar registers s60..s95, declared as
clobbers, and take the query-stream address in an SGPR pair.  This script compiles the translation
units to gfx950 assembly and checks what the compiler did around the asm blocks:

  1. the operands the compiler hands into a block, and takes out of it, never sit in a register the loop
     hard-codes (s60..s71, s72..s75 of the rows that park a carry chain, the column-block loops' s80/s81, the banded loop's s72..s95) — the clobber list
     is the contract, this checks the compiler kept it (the compiler may and does reuse those registers for
     its own temporaries BETWEEN blocks);
  2. no s_bfe_i64 in an asm kernel — the signature of round 1's fault: the stream address built as
     `int readfirstlane(lo) | (u64(hi) << 32)` sign-extended the low half, so a workspace whose address
     had bit 31 set became 0xffffffffXXXXXXXX;
  3. every asm kernel has the fault-word plumbing: at least one asm block and a global atomic OR.

    python3 scripts/check_asm_kernels.py            # exit status 0 = clean
"""
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "bgsa_amd" / "csrc"


def clobbered(func: str) -> set:
    """The hard-coded SGPRs of the loop a kernel carries (gen_rows_asm.py: CLOBBERS and the per-loop extras)."""
    regs = set(range(60, 72))
    if re.search(r"myers_global_asm_kernelILi3[02]E", func):     # 30 / 32 words: the parked carry chains (gen_rows_asm.py: S_PARK)
        regs |= {72, 73, 74, 75}
    if "blocked_kernel" in func:
        regs |= {80, 81}
    if "banded_asm_kernel" in func:
        regs |= set(range(72, 96))
    if "banded_cut_kernel" in func:
        regs |= set(range(72, 100)) | {55, 56, 57, 58, 59}
    if "banded_chunk_kernel" in func:
        regs = set(range(60, 94))
    return regs


def compile_to_asm(src: Path, out: Path) -> None:
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "--offload-device-only",
                    "-S", str(src), "-o", str(out)], check=True, cwd=CSRC, stderr=subprocess.DEVNULL)


def sgprs(line: str):
    line = line.split(";")[0]
    for m in re.finditer(r"\bs\[(\d+):(\d+)\]", line):
        yield from range(int(m.group(1)), int(m.group(2)) + 1)
    for m in re.finditer(r"(?<![\w\[:])s(\d+)\b", line):
        yield int(m.group(1))


def check(asm_text: str, name: str):
    problems, kernels = [], 0
    func, lines, in_asm = None, [], False
    for raw in asm_text.splitlines():
        m = re.match(r"^(_ZN4bgsa\w+):", raw)
        if m:
            func, lines = m.group(1), []
            continue
        if func is None:
            continue
        lines.append(raw)
        if raw.strip().startswith("s_endpgm") or raw.strip().startswith(".Lfunc_end"):
            body = "\n".join(lines)
            if ";;#ASMSTART" in body:
                kernels += 1
                in_asm = False
                hard = clobbered(func)
                for ln in lines:
                    if ";;#ASMSTART" in ln:
                        in_asm, phase = True, "prologue"
                        continue
                    if ";;#ASMEND" in ln:
                        in_asm = False
                        continue
                    code = ln.split(";")[0].strip()
                    if not code or code.startswith("."):
                        continue
                    if in_asm:
                        # hand-over moves between the compiler's operands and the loop's hard-coded registers
                        # hand-over moves: before L_anchor the sources are the compiler's operands, after L_done the
                        # destinations are; neither may sit in a register the loop hard-codes
                        if code.startswith("L_anchor") or code.startswith("L_chunk"):
                            phase = "loop"
                        elif code.startswith("L_done"):
                            phase = "epilogue"
                        m = re.match(r"s_mov_b(?:32|64)\s+(\S+),\s*(\S+)$", code)
                        if m and phase != "loop":
                            dst, src = (set(sgprs(x)) for x in m.groups())
                            theirs = src if phase == "prologue" else dst
                            if theirs & hard:
                                problems.append(f"{name}: {func}: compiler operand in a hard-coded register ({phase}): {code}")
                        continue
                    if code.startswith("s_bfe_i64"):
                        problems.append(f"{name}: {func}: s_bfe_i64 (sign-extended 64-bit scalar) feeds an asm kernel: {code}")
                # (the chunk kernel has no token dispatch: a damaged token can only select a wrong LDS word, never a jump)
                if "global_atomic_or" not in body and "banded_chunk_kernel" not in func:
                    problems.append(f"{name}: {func}: no global_atomic_or — the stream-fault report is missing")
            func = None
    return kernels, problems


def main() -> int:
    sources = [CSRC / "myers_global.hip", CSRC / "banded.hip"] + sorted((CSRC / "_gen").glob("bitpal_set_2_m3_m5.hip"))
    with tempfile.TemporaryDirectory() as tmp:
        outs = [Path(tmp) / (s.stem + ".s") for s in sources]
        with ThreadPoolExecutor(max_workers=3) as pool:
            list(pool.map(lambda so: compile_to_asm(*so), zip(sources, outs)))
        total, problems = 0, []
        for s, o in zip(sources, outs):
            k, p = check(o.read_text(), s.name)
            total += k
            problems += p
    for p in problems:
        print("FAIL", p)
    print(f"{total} kernels with generated asm row loops checked, {len(problems)} problem(s)")
    return 1 if problems or total == 0 else 0


if __name__ == "__main__":
    raise SystemExit(main())
