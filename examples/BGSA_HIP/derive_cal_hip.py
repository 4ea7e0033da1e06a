#!/usr/bin/env python3
"""derive_cal_hip.py — `original/BGSA_HIP/cal_hip.c` from the reference's `original/BGSA_CPU/cal_cpu.c`, at build time.

    python3 derive_cal_hip.py /root/reference/original/BGSA_CPU/cal_cpu.c /tmp/x/cal_hip.c

INTEGRATION.md §2 says what a maintainer's cal_hip.c is: cal_cpu.c with ONE change — the grid function
`cpu_cal_align_score` (cal_cpu.c:43-85: an OpenMP loop that calls align_cpu per query and chunk) is no longer defined in the
file, because the library exports the function with that signature (`hip_cal_align_score`, include/bgsa_hip.h; cal.h:48 keeps
declaring it).  Everything else — cpu_cal (:88-119), cal_on_cpu with its bucket loop, double buffers, I/O threads and report
(:121-476) — stays as upstream wrote it, and `cpu_cal`'s call now lands in the library: ONE device launch per block of
REF_BUCKET_COUNT queries, which is the seam SURVEY §8(b) designates for a device backend (the precedent is mic_cal,
BGSA_KNC/cal_mic.c:86-154).

This script performs that one change on the reference file WHERE IT LIES and writes the result to a path the caller names
(oracle/Makefile passes a temporary directory and deletes it after compiling): no line of the reference is stored in this
repository.  It removes exactly one function definition — located by name, delimited by brace matching — and fails if the
file does not look as expected; the compile line then renames the remaining references with
`-Dcpu_cal_align_score=hip_cal_align_score` (examples/BGSA_HIP/config_hip.h does the same for the other two seams).
"""
import re
import sys


def strip_function(text: str, name: str) -> str:
    heads = [m for m in re.finditer(r"^[ \t]*void[ \t]*\n?[ \t]*" + re.escape(name) + r"[ \t]*\(", text, flags=re.M)]
    defs = []
    for m in heads:
        close = text.index(")", m.end())
        rest = text[close + 1:].lstrip()
        if rest.startswith("{"):            # a definition, not the prototype
            defs.append((m.start(), text.index("{", close)))
    if len(defs) != 1:
        raise SystemExit(f"derive_cal_hip: expected exactly one definition of {name}, found {len(defs)}")
    start, brace = defs[0]
    depth, i = 0, brace
    while True:
        c = text[i]
        if c == "{":
            depth += 1
        elif c == "}":
            depth -= 1
            if depth == 0:
                break
        i += 1
    body = text[brace:i + 1]
    if "align_cpu" not in body or "omp parallel for" not in body:
        raise SystemExit(f"derive_cal_hip: {name} is not the grid over align_cpu this script was written for")
    note = (f"/* {name}: defined by libbgsa_hip.so (hip_cal_align_score, include/bgsa_hip.h) — one device launch per call;\n"
            f"   the OpenMP grid over align_cpu that stood here is the library's job now (derive_cal_hip.py) */\n")
    return text[:start] + note + text[i + 1:]


def main() -> int:
    src, dst = sys.argv[1], sys.argv[2]
    text = open(src, encoding="utf-8", errors="surrogateescape").read()
    out = strip_function(text, "cpu_cal_align_score")
    if out.count("cpu_cal_align_score(") != text.count("cpu_cal_align_score(") - 1:
        raise SystemExit("derive_cal_hip: the call in cpu_cal must stay")
    open(dst, "w", encoding="utf-8", errors="surrogateescape").write(out)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
