#!/usr/bin/env python3
"""oracle/derive_avx2_myers.py — TEST INFRASTRUCTURE ONLY (build-time, build container only).

The reference's AVX2 *Myers* kernel is generator output that is not committed upstream (only the SSE
instance is: original/BGSA_SSE/align_core.c:19-152) and there is no JVM here to run generator.jar.  The
generator emits the AVX2 file from the same template as the SSE one — MyersGenerator.genMyersCommon
(generator/.../generator/MyersGenerator.java:225-401) asks the arch object for every type name and
intrinsic, and AVX2Arch / AVX2Intrinsics (arch/AVX2Arch.java:23-60, intrinsics/AVX2Intrinsics.java) answer
exactly what SSEArch / SSEIntrinsics answer with `_mm_` -> `_mm256_`, `si128` -> `si256`,
`__m128i` -> `__m256i`, `SSE_` -> `AVX_`, `sse_` -> `avx_`.  This script applies
that token mapping to the SSE instance where it lies under /root/reference and writes the result to the path
it is given — a TEMPORARY file that oracle/Makefile compiles with original/BGSA_AVX2's own host files and
deletes: no reference text, derived or not, enters the repository or travels to the GPU box; only the binary
oracle/_ref/original_avx2_myers/aligner does.

    python3 derive_avx2_myers.py <reference>/original/BGSA_SSE/align_core.c <out.c>
"""
import re
import sys

# (pattern, replacement): the names the arch classes hand to the template, SSE -> AVX2
TOKEN_MAP = [
    (r"\b_mm_", "_mm256_"),          # SSEIntrinsics -> AVX2Intrinsics: every intrinsic keeps its suffix
    (r"si128\b", "si256"),
    (r"\b__m128i\b", "__m256i"),
    (r"\bSSE_", "AVX_"),             # SSE_V_NUM, SSE_WORD_SIZE  (propMap vNumStr / wordStr)
    (r"\bsse_", "avx_"),             # sse_read_t, sse_write_t, sse_data_t  (propMap readType / writeType / wordType)
    (r"\balign_sse\b", "align_avx"),  # propMap declareStr
]


def derive(text: str) -> str:
    for pat, rep in TOKEN_MAP:
        text = re.sub(pat, rep, text)
    # what must hold for the result to be the generator's AVX2 instance and nothing else
    left = re.findall(r"_mm_|si128|__m128i|SSE_|sse_", text)
    if left:
        raise SystemExit(f"derive_avx2_myers: unmapped SSE tokens left: {sorted(set(left))}")
    for needle in ("void align_avx(", "_mm256_add_epi32", "_mm256_mullo_epi32", "AVX_V_NUM", "_mm256_load_si256"):
        if needle not in text:
            raise SystemExit(f"derive_avx2_myers: the source is not the Myers SSE instance this mapping knows ({needle} missing)")
    if "match_score = 0" not in text.replace("  ", " "):
        raise SystemExit("derive_avx2_myers: the source is not a Myers (0, -1, -1) kernel")
    return text


if __name__ == "__main__":
    if len(sys.argv) != 3:
        raise SystemExit(__doc__)
    with open(sys.argv[1]) as f:
        out = derive(f.read())
    with open(sys.argv[2], "w") as f:
        f.write(out)
