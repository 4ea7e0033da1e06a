#!/usr/bin/env python3
"""Generate the gfx950 inline-asm row loops -> myers_rows_gen.inc, bitpal_rows_gen.inc.

The row bodies themselves are described once, as instruction lists, in rows_ir.py (which also
interprets them on the CPU for tests/test_rows_ir.py).  This script wraps five copies of a body
(one per query character class) in the threaded-code row loop and writes the C++ header that the
.hip kernels include.  The output is committed so the exact instruction stream that ships can be
read without running anything.

Why asm at all: the row body must exist five times (selected by a scalar jump, so
`Eq = Peq[c][w]` costs nothing) and must update its state in place.  hipcc turns a five-way switch
into v_mov copies around one shared body (+20 % VALU work) and lowers the inter-word shifts to
v_alignbit_b32, a slow-issue-class instruction on gfx950 (rows_ir.py, scripts/ubench/).

Control flow is threaded code: the query is a packed stream of one-byte codes (0..4 = A C G T N
row bodies, 5 = END, 6 = REFILL), 7 characters + 1 REFILL per 8-byte window, fetched with
s_load_dwordx2 one window ahead (bgsa_common.h "Packed query stream").  (`glc` on these loads was
measured: -3 % Myers, -37 % banded — the short banded rows do not cover an L2 round trip — and an
extra s_dcache_inv per wave cost 3 %; the dispatch's own acquire fence is what keeps the scalar
cache coherent with the packer kernel's stores, as for any kernel argument.)  Each body ends with the
dispatch of the next code (s_and / s_lshr_b64 / s_mul / s_add / s_addc / s_setpc): a row costs one
taken branch and no loop counter.  All five bodies have the same byte length (same opcodes and
operand classes), so the jump target is base + code * stride, both computed by the assembler from
label differences.
"""
from __future__ import annotations

import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
import rows_ir as R  # noqa: E402

MYERS_NW = [1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16, 18, 20, 22, 24, 25]  # Peq masks resident: 8 VALU per word (25 words = 253 VGPRs, the last width with two waves per SIMD)
# resident Peq masks with 9 registers per word (rows_ir.myers_parked_body): 26 and 28 words (801..896 bp) fit 256 VGPRs at all,
# 18 words (545..576 bp) drop from 183 to 165 VGPRs = three waves per SIMD instead of two
MYERS_PARKED_NW = [18, 26, 28]
# 30 and 32 words (897..1024 bp) with the Peq planes resident (round 5): the two carry chains take turns over blocks of
# MYERS_SPLIT words (rows_ir.myers_body(split=K)), so a row holds 2K temporaries instead of 2*nw: 7*nw + 2K + the kernel's own
# registers <= 256.  Until then these widths ran on the code planes (nine instructions per word).
MYERS_SPLIT_NW = [30, 32]
# Block size: K = 4 / 6 / 7 / 8 / 9 / 10 / 11 / 12 measured (config 5, ms per pass; profiles/r05_split_ab.txt, r05_split_k_ab.txt, r05_split_k9.txt):
# static grids 4,855 / 4,287 / 4,174 / 4,174 / 4,132 / 4,114 / (4,250 against 4,107 on another box) / 4,379; the counter instantiation,
# worth 1 - 3.5 %, fits 256 VGPRs up to K = 9 (255): K = 8 4,036 - 4,040, K = 9 3,980 - 3,985 (930 bp: 1,846 -> 1,805).
MYERS_SPLIT = int(os.environ.get("BGSA_GEN_MYERS_SPLIT", "9"))
MYERS_SEMI_SPLIT_NW = [26, 28, 30, 32]   # semi-global beyond 25 words with resident Peq planes (round 5; on the code planes before: MYERS_SEMI_PLANES_NW)
MYERS_PARK = os.environ.get("BGSA_GEN_MYERS_PARK", "sgpr")     # where the pausing chain waits: a scalar pair, or "vgpr" (two more VALU per switch)
# "gap,window[,instructions from carry link to carry link]" of rows_ir.schedule_ilp for the Myers GLOBAL bodies with resident Peq planes
# ("0" = the bodies as written: every instruction right behind the one it reads from).  Default 2,24 since round 5, same-box A/Bs in
# profiles/r05_ilp_ab.txt, r05_balance_ab.txt: 1000 bp (32 words, chains in turns over 8) 4,205 -> 4,030 ms, 768 bp 1,212 -> 1,196, 512 bp
# 488.6 -> 485.8, 150 bp 330.8 -> 329.2; a minimum link distance of 4 or 5 and the `balanced` phases measured no better.  The code-plane
# bodies stay as written: scheduled they measured SLOWER (32 words: 4,164 -> 4,238 / 4,318 ms).
MYERS_ILP = tuple(int(x) for x in os.environ.get("BGSA_GEN_MYERS_ILP", "2,24").split(","))
MYERS_BALANCED = os.environ.get("BGSA_GEN_MYERS_BALANCED", "0") == "1"    # HP of word w + 1 formed in phase B: four instructions per carry link in both phases


def ilp(body: R.Body) -> R.Body:
    return R.schedule_ilp(body, MYERS_ILP[0], MYERS_ILP[1], *MYERS_ILP[2:3]) if MYERS_ILP[0] > 0 else body


def ilp_planes(body: R.Body) -> R.Body:      # BGSA_GEN_MYERS_ILP_PLANES=1: the measurement builds that schedule the code-plane bodies too
    return ilp(body) if os.environ.get("BGSA_GEN_MYERS_ILP_PLANES", "0") == "1" else body


MYERS_PLANES_SPLIT = int(os.environ.get("BGSA_GEN_MYERS_PLANES_SPLIT", "0"))   # A/B: the code-plane rows of 30 / 32 words with the chains in turns
MYERS_PEQ_BLOCK_NW = [12, 14, 16, 18, 20]  # column blocks with resident Peq planes (20 words: 238 VGPRs; 22 would need 256)
MYERS_PAIR_NW = [1, 2]  # two rows per stream token: the 10-20 VALU row cannot hide the scalar dispatch
MYERS_PLANES_NW = [10, 12, 14, 16, 18, 20, 22, 24, 26, 28, 30, 32]  # at most one padding word
MYERS_SEMI_PLANES_NW = [26, 28, 30, 32]  # semi-global beyond the resident Peq planes (24 words), up to 1024 bp
MYERS_BLOCK_NW = [12, 14, 16, 18, 20, 22, 24, 26, 28]  # block widths of the > 1024 bp kernel (32 would need 256 VGPRs: 1 wave/SIMD)
BITPAL_VGPR_BUDGET = 224        # state + masks + temporaries a plain BitPAl kernel may hold
BITPAL_BLOCK_VGPR_BUDGET = 248  # a column-block kernel in total: two waves per SIMD need <= 256

# Scalar scratch registers, hard-coded and declared as clobbers (inline asm cannot name the
# halves of a 64-bit "s" operand, and the jump needs lo/hi arithmetic).
S_WIN, S_WIN_LO = "s[60:61]", "s60"
S_NXT = "s[62:63]"
S_BASE_LO, S_BASE_HI = "s64", "s65"
S_PC, S_PC_LO, S_PC_HI = "s[66:67]", "s66", "s67"
S_C = "s68"
S_PTR, S_PTR_LO, S_PTR_HI = "s[70:71]", "s70", "s71"
S_PARK = ["s[72:73]", "s[74:75]"]   # carry chains parked between their turns (rows_ir: SAVECC / LOADCC), clobbered by the loops that use them
S_LEFT = "s69"  # windows this stream may still fetch; handed back to the caller: >= 0 after a well-formed stream,
                # -1 = the budget ran out before END, -2 = a byte that is no stream code (L_fail slots)
CLOBBERS = ["s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71",
            "vcc", "scc", "memory"]


def dispatch() -> list[str]:
    return [
        f"s_and_b32 {S_C}, {S_WIN_LO}, 7",
        f"s_lshr_b64 {S_WIN}, {S_WIN}, 8",
        f"s_mul_i32 {S_C}, {S_C}, (L_body1_%= - L_body0_%=)",
        f"s_add_u32 {S_PC_LO}, {S_BASE_LO}, {S_C}",
        f"s_addc_u32 {S_PC_HI}, {S_BASE_HI}, 0",
        f"s_setpc_b64 {S_PC}",
    ]


def weave_dispatch(body_lines: list[str], disp_lines: list[str], tail: int = 14) -> list[str]:
    """A row body followed by the dispatch of the next token, with the dispatch's scalar instructions placed BETWEEN the
    body's last vector instructions instead of behind them.  The five scalar instructions that turn the next stream code
    into a jump target are a dependent chain (s_and -> s_mul -> s_add -> s_addc) in front of a taken branch: behind the
    body they are some 40 cycles in which this wave feeds the vector pipe nothing, and a SIMD needs two waves ready at
    any time to issue a vector instruction every other cycle.  Between vector instructions they cost issue slots only.
    Measured (round 3, same box, scripts/r03_weave_ab.sh): it pays where the row is SHORT — the banded loop, whose token
    holds 48 vector instructions — and not where a row is hundreds of them: Myers 150 bp 1046.1 -> 1047.9 ms, 1000 bp
    5100.9 -> 5221.3 ms (two waves per SIMD: the scalar instructions between the links of the VCC chains cost more than the
    bubble they hide), BitPAl unchanged, column blocks at 2,000 bp 42.3 -> 44.4 s.  So only the banded loops use it.
    Nothing that writes SCC may sit between s_add_u32 and s_addc_u32: the bodies hold VALU instructions, s_nop and
    `s_mov_b64 vcc` only — checked here."""
    scalars, jump = disp_lines[:-1], disp_lines[-1]
    assert jump.startswith("s_setpc_b64")
    n = len(body_lines)
    first = max(0, n - tail)
    for ln in body_lines[first:]:
        assert ln.startswith(("v_", "s_nop", "s_mov_b64 vcc", "global_")), f"instruction with unknown SCC behaviour in a woven tail: {ln}"
    slots = n - first
    gap = max(1, slots // (len(scalars) + 1))
    out = list(body_lines[:first])
    k_next = 0
    for i, ln in enumerate(body_lines[first:]):
        out.append(ln)
        if k_next < len(scalars) and (i + 1) % gap == 0 and i + 1 < slots:
            out.append(scalars[k_next])
            k_next += 1
    out += scalars[k_next:]
    out.append(jump)
    return out


def fail_slot() -> list[str]:
    """Slot of a byte value that is no stream code: leave the loop and say so (S_LEFT = -2)."""
    return [f"s_mov_b32 {S_LEFT}, -2", "s_branch L_done_%="]


def done(wait: str = "s_waitcnt lgkmcnt(0)") -> list[str]:
    """Common exit: every path out of the loop lands here; the caller gets S_LEFT (see above)."""
    return ["L_done_%=:", wait, f"s_mov_b32 %[left], {S_LEFT}"]


def gen_function(fn_name: str, template_args: str, body: R.Body, n_state: int, n_eq: int,
                 n_planes: int = 0) -> str:
    """n_planes > 0: the body reads class-independent code planes B[] instead of per-class masks."""
    slot_of, n_slots = body.allocate_temps()

    def reg_for(c: int):
        def reg(name: str) -> str:
            if name.startswith("S"):
                return f"%[s{name[1:]}]"
            if name.startswith("E"):
                return f"%[e{c}_{name[1:]}]"
            if name.startswith("B"):
                return f"%[b{name[1:]}]"
            if name.startswith("$c"):
                return S_PARK[int(name[2:])]
            return f"%[t{slot_of[name]}]"
        return reg

    asm: list[str] = []
    asm += [
        f"s_mov_b64 {S_PTR}, %[qp]",
        f"s_mov_b32 {S_LEFT}, %[nwin]",
        f"s_load_dwordx2 {S_WIN}, {S_PTR}, 0x0",
        f"s_load_dwordx2 {S_NXT}, {S_PTR}, 0x8",
        f"s_getpc_b64 {S_PC}",
        "L_anchor_%=:",
        f"s_add_u32 {S_BASE_LO}, {S_PC_LO}, (L_body0_%= - L_anchor_%=)",
        f"s_addc_u32 {S_BASE_HI}, {S_PC_HI}, 0",
        "s_waitcnt lgkmcnt(0)",  # scalar loads may return out of order: only 0 is safe
    ]
    asm += dispatch()
    for c in range(5):
        asm.append(f"L_body{c}_%=:")
        asm += body.emit_asm(reg_for(c), c)
        asm += dispatch()
    # END (code 5) and REFILL (code 6) live in slots of the same stride as the row bodies.
    asm.append("L_body5_%=:")
    asm.append("s_branch L_done_%=")
    asm.append(".fill ((L_body1_%= - L_body0_%=) - 4) / 4, 4, 0xbf800000")  # s_nop padding, never executed
    asm.append("L_body6_%=:")
    asm += [
        # Exit condition every wave reaches: a well-formed stream never exhausts this budget (its
        # END code comes first); a corrupt one stops here instead of walking memory.
        f"s_sub_u32 {S_LEFT}, {S_LEFT}, 1",
        "s_cbranch_scc1 L_done_%=",
        "s_waitcnt lgkmcnt(0)",
        f"s_mov_b64 {S_WIN}, {S_NXT}",
        f"s_add_u32 {S_PTR_LO}, {S_PTR_LO}, 8",
        f"s_addc_u32 {S_PTR_HI}, {S_PTR_HI}, 0",
        f"s_load_dwordx2 {S_NXT}, {S_PTR}, 0x8",
    ]
    asm += dispatch()
    asm.append("L_refill_end_%=:")
    asm.append(".fill ((L_body1_%= - L_body0_%=) - (L_refill_end_%= - L_body6_%=)) / 4, 4, 0xbf800000")
    asm.append("L_body7_%=:")  # the dispatch masks the code with 7: the one value that is no code lands here
    asm += fail_slot()
    asm += done()

    text = "\n".join(f'        "{line}\\n\\t"' if not line.endswith(":") else f'        "{line}\\n"' for line in asm)
    outs = [f'[s{i}] "+v"(state[{i}])' for i in range(n_state)]
    outs += ['[left] "=s"(left)']
    outs += [f'[t{i}] "=&v"(tmp[{i}])' for i in range(n_slots)]
    if n_planes:
        ins = [f'[b{j}] "v"(B[{j}])' for j in range(n_planes)]
        masks_param = f"const uint32_t (&B)[{n_planes}]"
    else:
        ins = [f'[e{c}_{j}] "v"(P[{c}][{j}])' for c in range(5) for j in range(n_eq)]
        masks_param = f"const uint32_t (&P)[5][{n_eq}]"
    ins.append('[qp] "s"(stream)')
    ins.append('[nwin] "s"(n_windows)')
    parks = ["s72", "s73", "s74", "s75"] if any(op.kind in ("savecc", "loadcc") for op in body.ops) else []
    clob = ", ".join(f'"{c}"' for c in CLOBBERS[:-3] + parks + CLOBBERS[-3:])
    nops = sum(line.startswith("s_nop") for line in body.emit_asm(lambda x: S_PARK[int(x[2:])] if x.startswith("$c") else x, 0))
    return f"""
// {body.valu_count()} VALU per row, {n_slots} temporaries, {nops} hazard nops{f", {body.salu_count()} scalar moves of VCC" if parks else ""}
template <>
__device__ __forceinline__ int {fn_name}<{template_args}>(uint32_t (&state)[{n_state}],
                                                   {masks_param},
                                                   const unsigned long long stream,
                                                   const int n_windows)
{{
    uint32_t tmp[{max(n_slots, 1)}];
    int left;
    asm volatile(
{text}
        : {", ".join(outs)}
        : {", ".join(ins)}
        : {clob});
    return left;
}}
"""


def gen_pair_function(fn_name: str, template_args: str, body: R.Body, n_state: int, n_eq: int) -> str:
    """Row loop for bodies too short to hide the scalar dispatch (Myers on short subjects: 10-20 VALU
    per row): every stream token carries TWO rows.  Slots: 0..24 two rows of classes a, b (code 5a + b),
    25..29 one row (the odd last row), 30 END, 31 REFILL (bgsa_common.h: pair_stream_window)."""
    slot_of, n_slots = body.allocate_temps()

    def reg_for(c: int):
        def reg(name: str) -> str:
            if name.startswith("S"):
                return f"%[s{name[1:]}]"
            if name.startswith("E"):
                return f"%[e{c}_{name[1:]}]"
            return f"%[t{slot_of[name]}]"
        return reg

    def disp() -> list[str]:
        return [
            f"s_and_b32 {S_C}, {S_WIN_LO}, 0x1f",
            f"s_lshr_b64 {S_WIN}, {S_WIN}, 8",
            f"s_mul_i32 {S_C}, {S_C}, (L_body1_%= - L_body0_%=)",
            f"s_add_u32 {S_PC_LO}, {S_BASE_LO}, {S_C}",
            f"s_addc_u32 {S_PC_HI}, {S_BASE_HI}, 0",
            f"s_setpc_b64 {S_PC}",
        ]

    def pad(slot: int) -> str:
        return f".fill ((L_body1_%= - L_body0_%=) - (L_end{slot}_%= - L_body{slot}_%=)) / 4, 4, 0xbf800000"

    asm = [
        f"s_mov_b64 {S_PTR}, %[qp]",
        f"s_mov_b32 {S_LEFT}, %[nwin]",
        f"s_load_dwordx2 {S_WIN}, {S_PTR}, 0x0",
        f"s_load_dwordx2 {S_NXT}, {S_PTR}, 0x8",
        f"s_getpc_b64 {S_PC}",
        "L_anchor_%=:",
        f"s_add_u32 {S_BASE_LO}, {S_PC_LO}, (L_body0_%= - L_anchor_%=)",
        f"s_addc_u32 {S_BASE_HI}, {S_PC_HI}, 0",
        "s_waitcnt lgkmcnt(0)",
    ]
    asm += disp()
    for a in range(5):
        for b2 in range(5):
            asm.append(f"L_body{5 * a + b2}_%=:")
            asm += body.emit_asm(reg_for(a), a)
            asm += body.emit_asm(reg_for(b2), b2)
            asm += disp()
    for c in range(5):
        asm.append(f"L_body{25 + c}_%=:")
        asm += body.emit_asm(reg_for(c), c)
        asm += disp()
        asm.append(f"L_end{25 + c}_%=:")
        asm.append(pad(25 + c))
    asm.append("L_body30_%=:")  # END
    asm.append("s_branch L_done_%=")
    asm.append("L_end30_%=:")
    asm.append(pad(30))
    asm.append("L_body31_%=:")  # REFILL (same exit condition as gen_function)
    asm += [
        f"s_sub_u32 {S_LEFT}, {S_LEFT}, 1",
        "s_cbranch_scc1 L_done_%=",
        "s_waitcnt lgkmcnt(0)",
        f"s_mov_b64 {S_WIN}, {S_NXT}",
        f"s_add_u32 {S_PTR_LO}, {S_PTR_LO}, 8",
        f"s_addc_u32 {S_PTR_HI}, {S_PTR_HI}, 0",
        f"s_load_dwordx2 {S_NXT}, {S_PTR}, 0x8",
    ]
    asm += disp()
    asm += done()   # the dispatch masks with 0x1f and all 32 slots exist: no byte value can leave the table

    text = "\n".join(f'        "{line}\\n\\t"' if not line.endswith(":") else f'        "{line}\\n"' for line in asm)
    outs = [f'[s{i}] "+v"(state[{i}])' for i in range(n_state)]
    outs += ['[left] "=s"(left)']
    outs += [f'[t{i}] "=&v"(tmp[{i}])' for i in range(n_slots)]
    ins = [f'[e{c}_{j}] "v"(P[{c}][{j}])' for c in range(5) for j in range(n_eq)]
    ins.append('[qp] "s"(stream)')
    ins.append('[nwin] "s"(n_windows)')
    clob = ", ".join(f'"{c}"' for c in CLOBBERS)
    return f"""
// two rows per token: {2 * body.valu_count()} VALU per token, {n_slots} temporaries
template <>
__device__ __forceinline__ int {fn_name}<{template_args}>(uint32_t (&state)[{n_state}],
                                                   const uint32_t (&P)[5][{n_eq}],
                                                   const unsigned long long stream,
                                                   const int n_windows)
{{
    uint32_t tmp[{max(n_slots, 1)}];
    int left;
    asm volatile(
{text}
        : {", ".join(outs)}
        : {", ".join(ins)}
        : {clob});
    return left;
}}
"""


def gen_banded_function(wide: bool, phase: bool = False) -> str:
    """Row loop of the banded kernel: 32-bit band word (k <= 15) or a 64-bit pair (k <= 31).  Same
    threaded-code skeleton plus stream code 63 = EVENT (followed by an argument byte): test/latch the
    reject mask, reset the error count at row k, advance the match-string words every 32 rows
    (rows_ir.banded_tokens).  phase: the band held in place (rows_ir.banded_phase_body, k <= 11) — the rows read
    the phase's window of their class (W[c]) and a re-anchor event (bit 16) every banded_phase_rows(k) rows shifts
    the state back down, folds the collected error bits and cuts the next windows."""
    assert not (wide and phase)
    body = R.banded_phase_body() if phase else (R.banded_body64() if wide else R.banded_body())
    n_state = 6 if phase else (5 if wide else 3)   # VP, VN (lo/hi when wide), errors; phase: + band mask, its low bit, error bits
    n_m = 4 if wide else 3              # resident match-string words per class (last = prefetch target)
    acc = 2 if phase else n_state - 1   # the error count
    slot_of, n_slots = body.allocate_temps()
    S_SH, S_ARG, S_THR = "s73", "s74", "s78"
    S_PH, S_MASK0 = "s72", "s75"       # phase only: rows per phase, the band mask at offset 0
    S_DEAD, S_TMP = "s[76:77]", "s[90:91]"
    S_BASE = [f"s[{80 + 2 * c}:{81 + 2 * c}]" for c in range(5)]
    # survivor compaction (banded.hip "survivor queue"): rows done = 32 * S_CHUNK + S_SH; a test at or after row
    # S_PUSHROW that finds 1..S_PUSHMAX lanes alive ends the wave with S_EARLY = 1, the alive lanes go to the queue
    S_CHUNK, S_PUSHROW, S_PUSHMAX, S_EARLY, S_CNT = "s92", "s93", "s94", "s95", "s79"
    clobbers = CLOBBERS[:-3] + ["s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79"] + [f"s{i}" for i in range(80, 96)] + \
        ["vcc", "scc", "memory"]

    def reg_for(c: int):
        def reg(name: str) -> str:
            if name.startswith("S"):
                return f"%[s{name[1:]}]"
            if name.startswith("E"):
                return f"%[w{c}]" if phase else f"%[m{name[1:]}_{c}]"
            if name in ("$mask", "$mask_lo"):
                # the band mask lives in a VGPR: a three-source VOP3 with an SGPR source issues in the half-rate class
                # (scripts/ubench: k3_sgpr 4.4 cycles against 2.7-2.95 with three VGPRs)
                return "%[vmask]"
            if name == "$mask_hi":
                return "%[vmask_hi]"
            if name == "$sh":
                return S_SH
            if name == "$one":
                return "1"
            return f"%[t{slot_of[name]}]"
        return reg

    def disp() -> list[str]:
        # token codes 0..31 and 63 (bgsa_common.h: banded_stream_layout); slot stride = a two-row body
        return [
            f"s_and_b32 {S_C}, {S_WIN_LO}, 0x3f",
            f"s_lshr_b64 {S_WIN}, {S_WIN}, 8",
            f"s_mul_i32 {S_C}, {S_C}, (L_body1_%= - L_body0_%=)",
            f"s_add_u32 {S_PC_LO}, {S_BASE_LO}, {S_C}",
            f"s_addc_u32 {S_PC_HI}, {S_BASE_HI}, 0",
            f"s_setpc_b64 {S_PC}",
        ]

    def pad(slot: int) -> str:
        """s_nop filler (never executed) up to the common slot stride."""
        return f".fill ((L_body1_%= - L_body0_%=) - (L_end{slot}_%= - L_body{slot}_%=)) / 4, 4, 0xbf800000"

    asm = [
        f"s_mov_b64 {S_PTR}, %[qp]",
        f"s_mov_b32 {S_LEFT}, %[nwin]",
        f"s_load_dwordx2 {S_WIN}, {S_PTR}, 0x0",
        f"s_load_dwordx2 {S_NXT}, {S_PTR}, 0x8",
        f"s_mov_b32 {S_THR}, %[thr]",
    ] + ([f"s_mov_b32 {S_PH}, %[phase]", f"s_mov_b32 {S_MASK0}, %[mask0]"] if phase else []) + [
        f"s_mov_b32 {S_SH}, 0",
        f"s_mov_b64 {S_DEAD}, 0",
        f"s_mov_b32 {S_CHUNK}, 0",
        f"s_mov_b32 {S_EARLY}, 0",
        f"s_mov_b32 {S_PUSHROW}, %[pushrow]",
        f"s_mov_b32 {S_PUSHMAX}, %[pushmax]",
    ]
    asm += [f"s_mov_b64 {S_BASE[c]}, %[base{c}]" for c in range(5)]
    asm += [
        f"s_getpc_b64 {S_PC}",
        "L_anchor_%=:",
        f"s_add_u32 {S_BASE_LO}, {S_PC_LO}, (L_body0_%= - L_anchor_%=)",
        f"s_addc_u32 {S_BASE_HI}, {S_PC_HI}, 0",
        "s_waitcnt lgkmcnt(0)",
    ]
    asm += disp()
    for a in range(5):          # slots 0..24: two rows per token (the dispatch is the loop's scalar bottleneck)
        for b2 in range(5):
            asm.append(f"L_body{5 * a + b2}_%=:")
            asm += body.emit_asm(reg_for(a), a)
            asm.append(f"s_add_u32 {S_SH}, {S_SH}, 1")
            asm += body.emit_asm(reg_for(b2), b2)
            asm.append(f"s_add_u32 {S_SH}, {S_SH}, 1")
            asm += disp()
    for c in range(5):          # slots 25..29: one row
        asm.append(f"L_body{25 + c}_%=:")
        asm += body.emit_asm(reg_for(c), c)
        asm.append(f"s_add_u32 {S_SH}, {S_SH}, 1")
        asm += disp()
        asm.append(f"L_end{25 + c}_%=:")
        asm.append(pad(25 + c))
    asm.append("L_body30_%=:")  # END
    asm.append("s_branch L_done_%=")
    asm.append("L_end30_%=:")
    asm.append(pad(30))
    asm.append("L_body31_%=:")  # REFILL
    asm += [
        f"s_sub_u32 {S_LEFT}, {S_LEFT}, 1",
        "s_cbranch_scc1 L_done_%=",
        "s_waitcnt lgkmcnt(0)",
        f"s_mov_b64 {S_WIN}, {S_NXT}",
        f"s_add_u32 {S_PTR_LO}, {S_PTR_LO}, 8",
        f"s_addc_u32 {S_PTR_HI}, {S_PTR_HI}, 0",
        f"s_load_dwordx2 {S_NXT}, {S_PTR}, 0x8",
    ]
    asm += disp()
    asm.append("L_end31_%=:")
    asm.append(pad(31))
    for slot in range(32, 63):   # the dispatch masks with 0x3f: every value that is no token lands in a fail slot
        asm.append(f"L_body{slot}_%=:")
        asm += fail_slot()
        asm.append(f"L_end{slot}_%=:")
        asm.append(pad(slot))
    asm.append("L_body63_%=:")  # EVENT <arg>: the last slot, so it may be longer than the slot stride
    asm += [
        f"s_and_b32 {S_ARG}, {S_WIN_LO}, 0xff",
        f"s_lshr_b64 {S_WIN}, {S_WIN}, 8",
        # by far the most frequent event is the plain test (argument 4): its own short path, no bit tests
        f"s_cmp_eq_u32 {S_ARG}, 4",
        "s_cbranch_scc0 L_ev_general_%=",
    ] + ([f"v_bcnt_u32_b32 %[t0], %[s5], %[s{acc}]", f"v_cmp_lt_u32 vcc, {S_THR}, %[t0]"] if phase else
         [f"v_cmp_lt_u32 vcc, {S_THR}, %[s{acc}]"]) + [
        f"s_andn2_b64 {S_TMP}, exec, vcc",
        "s_cbranch_scc0 L_ev_alldead_%=",
        f"s_lshl_b32 {S_CNT}, {S_CHUNK}, 5",
        f"s_add_u32 {S_CNT}, {S_CNT}, {S_SH}",
        f"s_cmp_ge_u32 {S_CNT}, {S_PUSHROW}",
        "s_cbranch_scc0 L_ev_out_%=",
        f"s_bcnt1_i32_b64 {S_CNT}, {S_TMP}",
        f"s_cmp_le_u32 {S_CNT}, {S_PUSHMAX}",
        "s_cbranch_scc0 L_ev_out_%=",
        f"s_mov_b64 {S_DEAD}, vcc",
        f"s_mov_b32 {S_EARLY}, 1",
        "s_branch L_done_%=",
        "L_ev_alldead_%=:",
        f"s_mov_b64 {S_DEAD}, exec",
        "s_branch L_done_%=",
        "L_ev_general_%=:",
        # bit 2: test err > limit on every lane; bit 3: latch the reject mask (last checkpoint)
        f"s_bitcmp1_b32 {S_ARG}, 2",
        "s_cbranch_scc0 L_ev_reset_%=",
    ] + ([f"v_bcnt_u32_b32 %[t0], %[s5], %[s{acc}]", f"v_cmp_lt_u32 vcc, {S_THR}, %[t0]"] if phase else
         [f"v_cmp_lt_u32 vcc, {S_THR}, %[s{acc}]"]) + [
        f"s_bitcmp1_b32 {S_ARG}, 3",
        "s_cbranch_scc0 L_ev_all_%=",
        f"s_mov_b64 {S_DEAD}, vcc",
        "L_ev_all_%=:",
        f"s_andn2_b64 {S_TMP}, exec, vcc",  # lanes still within the limit
        "s_cbranch_scc1 L_ev_some_%=",
        f"s_mov_b64 {S_DEAD}, exec",       # every lane is past the limit: the wave is done
        "s_branch L_done_%=",
        "L_ev_some_%=:",
        # few survivors late enough: hand them to the pair queue instead of running 64 lanes for them
        f"s_lshl_b32 {S_CNT}, {S_CHUNK}, 5",
        f"s_add_u32 {S_CNT}, {S_CNT}, {S_SH}",
        f"s_cmp_ge_u32 {S_CNT}, {S_PUSHROW}",
        "s_cbranch_scc0 L_ev_reset_%=",
        f"s_bcnt1_i32_b64 {S_CNT}, {S_TMP}",
        f"s_cmp_le_u32 {S_CNT}, {S_PUSHMAX}",
        "s_cbranch_scc0 L_ev_reset_%=",
        f"s_mov_b64 {S_DEAD}, vcc",
        f"s_mov_b32 {S_EARLY}, 1",
        "s_branch L_done_%=",
        "L_ev_reset_%=:",
        f"s_bitcmp1_b32 {S_ARG}, 0",      # bit 0: scoring starts (row k)
        "s_cbranch_scc0 L_ev_adv_%=",
        f"v_mov_b32 %[s{acc}], 0",
    ] + (["v_mov_b32 %[s5], 0"] if phase else []) + [
        "L_ev_adv_%=:",
        f"s_bitcmp1_b32 {S_ARG}, 1",      # bit 1: next 32 rows -> shift the match-string words down
        "s_cbranch_scc0 L_ev_anchor_%=",
        "s_waitcnt vmcnt(0)",
    ]
    for c in range(5):
        asm += [f"v_mov_b32 %[m{w}_{c}], %[m{w + 1}_{c}]" for w in range(n_m - 1)]
    for c in range(5):
        asm.append(f"global_load_dword %[m{n_m - 1}_{c}], %[voff], {S_BASE[c]}")
    asm += [
        "v_add_u32 %[voff], 0x100, %[voff]",
        f"s_mov_b32 {S_SH}, 0",
        f"s_add_u32 {S_CHUNK}, {S_CHUNK}, 1",
        "L_ev_anchor_%=:",
    ]
    if phase:
        asm += [
            f"s_bitcmp1_b32 {S_ARG}, 4",  # bit 4: the phase is over -> the band back to bit 0, its error bits counted, new windows
            "s_cbranch_scc0 L_ev_out_%=",
            f"v_lshrrev_b32 %[s0], {S_PH}, %[s0]",
            f"v_lshrrev_b32 %[s1], {S_PH}, %[s1]",
            f"v_bcnt_u32_b32 %[s{acc}], %[s5], %[s{acc}]",
            "v_mov_b32 %[s5], 0",
            f"v_mov_b32 %[s3], {S_MASK0}",
            "v_mov_b32 %[s4], 1",
        ]
        asm += [f"v_alignbit_b32 %[w{c}], %[m1_{c}], %[m0_{c}], {S_SH}" for c in range(5)]
    asm += [
        "L_ev_out_%=:",
    ]
    asm += disp()
    asm += done("s_waitcnt vmcnt(0) lgkmcnt(0)")
    asm.append(f"s_mov_b64 %[dead], {S_DEAD}")
    asm.append(f"s_mov_b32 %[early], {S_EARLY}")

    text = "\n".join(f'        "{line}\\n\\t"' if not line.endswith(":") else f'        "{line}\\n"' for line in asm)
    outs = [f'[s{i}] "+v"(state[{i}])' for i in range(n_state)]
    outs += [f'[m{w}_{c}] "+v"(M[{c}][{w}])' for c in range(5) for w in range(n_m)]
    if phase:
        outs += [f'[w{c}] "+v"(W[{c}])' for c in range(5)]
    outs += ['[voff] "+v"(voff)', '[dead] "=s"(dead)', '[left] "=s"(left)', '[early] "=s"(early)']
    outs += [f'[t{i}] "=&v"(tmp[{i}])' for i in range(n_slots)]
    ins = ['[qp] "s"(stream)', '[nwin] "s"(n_windows)'] + \
          (['[phase] "s"(phase_rows)', '[mask0] "s"(band_mask)'] if phase else ['[vmask] "v"(band_mask)']) + \
          (['[vmask_hi] "v"(band_mask_hi)'] if wide else []) + \
          ['[thr] "s"(limit)', '[pushrow] "s"(push_row)', '[pushmax] "s"(push_max)']
    ins += [f'[base{c}] "s"(base[{c}])' for c in range(5)]
    clob = ", ".join(f'"{x}"' for x in clobbers)
    fn_name = "banded_rows_phase_asm32" if phase else f"banded_rows_asm{64 if wide else 32}"
    w_param = "uint32_t (&W)[5], " if phase else ""
    last_param = "phase_rows" if phase else "band_mask_hi"
    phase_doc = ("// The band held in place: state = {VP, VN, errors before this phase, band mask at the row's offset, its lowest bit,\n"
                 "// the phase's error bits}; W[c] = the phase's match window of class c; phase_rows = banded_phase_rows(k).  After the\n"
                 "// loop the state sits at the offset of the rows of the LAST phase: the caller shifts VP / VN down by that and adds\n"
                 "// popcount(error bits) to the errors.\n") if phase else ""
    return f"""
// {body.valu_count()} VALU per row ({sum(op.kind in ('lshr1', 'alignbit') for op in body.ops)} of them slow-class), {n_slots} temporaries
// state = {{VP, VN, errors since row k}} (VP lo/hi, VN lo/hi, errors when wide); M[c][..] = consecutive
// 32-bit words of class c's offset match string (the last one is the prefetch target); voff = byte
// offset of the next word to fetch relative to base[c]; returns the reject mask (lanes whose error
// count passed `limit` at the last checkpoint, or all lanes if the wave stopped early).  left = the
// stream's remaining window budget, negative after a malformed stream (gen_rows_asm.py: S_LEFT).
// early = 1: a test at or after row push_row found 1..push_max lanes within the limit and the wave stopped
// there; the returned mask then holds the lanes past the limit, the others go to the survivor queue.
{phase_doc}__device__ __forceinline__ unsigned long long {fn_name}(uint32_t (&state)[{n_state}], {w_param}uint32_t (&M)[5][{n_m}], uint32_t &voff,
                                                              const unsigned long long (&base)[5],
                                                              const unsigned long long stream, const int n_windows,
                                                              const uint32_t band_mask, const uint32_t {last_param},
                                                              const uint32_t limit, const uint32_t push_row,
                                                              const uint32_t push_max, int &left, int &early)
{{
    uint32_t tmp[{max(n_slots, 1)}];
    unsigned long long dead;
    asm volatile(
{text}
        : {", ".join(outs)}
        : {", ".join(ins)}
        : {clob});
    return dead;
}}
"""


def gen_banded_cut_function(groups: int, form: str = "cut", sh64: bool = False) -> str:
    """form = "cut": row loop of the one-word-window banded kernel (k <= 12; rows_ir.banded_cut_body), for one or two subject
    groups per wave.  form = "funnel32" / "funnel64" (round 4): the same loop — two groups per wave sharing every dispatch,
    shift counter and event, the next token's dispatch woven under the last row, solid-survivor pushes — around the
    funnel-shift rows of thresholds 13 .. 15 (rows_ir.banded_funnel_body: one v_alignbit per group in the row, three
    match-string words per class and group, the last one the prefetch target) and 16 .. 31 (the 64-bit pair: four words);
    no cut events.  Same threaded-code skeleton, stream and events as gen_banded_function, plus EVENT bit 5 = cut.
    Per group and class TWO registers: m0 = the window the rows shift (A: at an advance the 32-bit match-string word the
    next 32 rows start in), m1 = the word behind it (B).  advance (bit 1, every 32 rows): A <- B, fetch B (awaited by
    the next cut, 16 or 8 rows later); cut (bit 5, every `cutrows` rows in between): A <- {B >> rows cut so far, A} >>
    cutrows, i.e. the window moves up by cutrows bits in place.  The row's shift count restarts at either.  (Until
    round 3's last change the loop kept four registers: the cut window apart from A, and the word after B prefetched a
    whole advance ahead: 91 VGPRs = five waves per SIMD; with two registers per class and group the kernel holds 73 = six.)
    sh64 (funnel64): rows_ir.banded_body64_sh64 — D0 of the pair row sits in a FIXED aligned register pair (clobbered by name; the
    compiler keeps clear of it) and D0 >> 1 is ONE v_lshrrev_b64 instead of a funnel shift and a plain one: 21 VALU per row.
    With two groups the tests, the push decision and the early exit look at both: the wave stops when all 128 lanes
    are past the limit."""
    G = groups
    assert form in ("cut", "funnel32", "funnel64")
    funnel, wide = form != "cut", form == "funnel64"
    assert wide or not sh64
    body = R.banded_cut_body(G) if not funnel else R.schedule(R.banded_funnel_body(G, wide, sh64=sh64), 8)
    PBASE = 2                                            # first fixed VGPR of the 'P' registers (rows_ir.Body)
    n_fixed = 2 * G if sh64 else 0
    P = lambda n: f"v{PBASE + n}"
    per_state = 5 if wide else 3
    n_state = per_state * G
    acc = [per_state * g + per_state - 1 for g in range(G)]
    n_m = 2 if not funnel else (4 if wide else 3)      # match-string words per class and group (funnel: the last = prefetch target)
    n_eq = 3 if wide else 2                            # ... that the row reads
    slot_of, n_slots = body.allocate_temps()
    S_CUT, S_SH, S_ARG, S_CUTROWS, S_THR, S_CNT = "s72", "s73", "s74", "s75", "s78", "s79"
    S_DEAD = ["s[76:77]", "s[96:97]"]
    S_ALIVE = ["s[90:91]", "s[98:99]"]
    S_ANY, S_VCC1, S_CNT2 = "s[58:59]", "s[98:99]", "s57"   # (s100 / s101 are reserved by the compiler)
    S_BASE = [f"s[{80 + 2 * c}:{81 + 2 * c}]" for c in range(5)]
    S_CHUNK, S_PUSHROW, S_PUSHMAX, S_EARLY = "s92", "s93", "s94", "s95"
    S_PUSHSOLID, S_THRSOLID = "s56", "s55"
    clobbers = [P(i) for i in range(n_fixed)] + ["s55", "s56", "s57", "s58", "s59"] + CLOBBERS[:-3] + [f"s{i}" for i in range(72, 100)] + \
               ["vcc", "scc", "memory"]

    def reg_for(c: int):
        def reg(name: str) -> str:
            if name.startswith("P"):
                return P(int(name[1:]))
            if name.startswith("S"):
                return f"%[s{name[1:]}]"
            if name.startswith("E") and funnel:
                j = int(name[1:])
                return f"%[m{j % n_eq}_{c}_{j // n_eq}]"
            if name.startswith("E"):
                return f"%[m0_{c}_{name[1:]}]"       # the window: register A
            if name in ("$mask", "$mask_lo"):
                return "%[vmask]"
            if name == "$mask_hi":
                return "%[vmask_hi]"
            if name == "$sh":
                return S_SH
            if name == "$one":
                return "1"
            return f"%[t{slot_of[name]}]"
        return reg

    def disp() -> list[str]:
        return [
            f"s_and_b32 {S_C}, {S_WIN_LO}, 0x3f",
            f"s_lshr_b64 {S_WIN}, {S_WIN}, 8",
            f"s_mul_i32 {S_C}, {S_C}, (L_body1_%= - L_body0_%=)",
            f"s_add_u32 {S_PC_LO}, {S_BASE_LO}, {S_C}",
            f"s_addc_u32 {S_PC_HI}, {S_BASE_HI}, 0",
            f"s_setpc_b64 {S_PC}",
        ]

    def pad(slot: int) -> str:
        return f".fill ((L_body1_%= - L_body0_%=) - (L_end{slot}_%= - L_body{slot}_%=)) / 4, 4, 0xbf800000"

    def test(tag: str) -> list[str]:
        """err > limit on every lane of every group -> vcc (group 0), S_VCC1 (group 1); alive masks; scc = any alive."""
        out = [f"v_cmp_lt_u32 vcc, {S_THR}, %[s{acc[0]}]"]
        if G == 2:
            out += [f"v_cmp_lt_u32_e64 {S_VCC1}, {S_THR}, %[s{acc[1]}]", "s_nop 1"]
        return out

    def alive() -> list[str]:
        out = [f"s_andn2_b64 {S_ALIVE[0]}, exec, vcc"]
        if G == 2:
            # S_ALIVE[1] aliases S_VCC1: take the reject mask out first
            out += [f"s_mov_b64 {S_ANY}, {S_VCC1}",
                    f"s_andn2_b64 {S_ALIVE[1]}, exec, {S_ANY}",
                    f"s_or_b64 {S_ANY}, {S_ALIVE[0]}, {S_ALIVE[1]}"]
        return out

    def latch_from_alive() -> list[str]:
        """reject masks = the lanes that are not alive."""
        return [f"s_andn2_b64 {S_DEAD[g]}, exec, {S_ALIVE[g]}" for g in range(G)]

    def push_or(label_no: str, tag: str) -> list[str]:
        """Few lanes within the limit, late enough: the wave stops here and hands them to the regroup list (banded.hip).
        Late enough = from row S_PUSHROW on (random pairs are dead by then: whatever is alive mostly stays alive), or
        already from row S_PUSHSOLID on if one of the alive lanes is a SOLID survivor — its error count at most S_THRSOLID.
        Between those rows the lanes still alive among random pairs are stragglers about to cross the limit: pushing them
        costs each a share of a dense pass for nothing (10k x 1M random pairs, pushing from row k + 40 without this test:
        88 -> 92 ms), while a wave that does hold a real survivor gains the rows it no longer runs for it (1 % dense
        survivors: 145 -> 133 ms)."""
        out = [
            f"s_lshl_b32 {S_CNT}, {S_CHUNK}, 5",
            f"s_add_u32 {S_CNT}, {S_CNT}, {S_CUT}",
            f"s_add_u32 {S_CNT}, {S_CNT}, {S_SH}",
            f"s_cmp_ge_u32 {S_CNT}, {S_PUSHSOLID}",
            f"s_cbranch_scc0 {label_no}",
            f"s_bcnt1_i32_b64 {S_CNT2}, {S_ALIVE[0]}",
        ]
        if G == 2:
            out += [f"s_bcnt1_i32_b64 {S_C}, {S_ALIVE[1]}", f"s_add_u32 {S_CNT2}, {S_CNT2}, {S_C}"]
        out += [
            f"s_cmp_le_u32 {S_CNT2}, {S_PUSHMAX}",
            f"s_cbranch_scc0 {label_no}",
            f"s_cmp_ge_u32 {S_CNT}, {S_PUSHROW}",
            f"s_cbranch_scc1 L_push_{tag}_%=",
            f"v_cmp_ge_u32 vcc, {S_THRSOLID}, %[s{acc[0]}]",
        ]
        if G == 2:
            out += [f"v_cmp_ge_u32_e64 {S_ANY}, {S_THRSOLID}, %[s{acc[1]}]", "s_nop 1", f"s_or_b64 vcc, vcc, {S_ANY}"]
        out += [
            "s_cmp_lg_u64 vcc, 0",
            f"s_cbranch_scc0 {label_no}",
            f"L_push_{tag}_%=:",
        ]
        out += latch_from_alive()
        out += [f"s_mov_b32 {S_EARLY}, 1", "s_branch L_done_%="]
        return out

    asm = [
        f"s_mov_b64 {S_PTR}, %[qp]",
        f"s_mov_b32 {S_LEFT}, %[nwin]",
        f"s_load_dwordx2 {S_WIN}, {S_PTR}, 0x0",
        f"s_load_dwordx2 {S_NXT}, {S_PTR}, 0x8",
        f"s_mov_b32 {S_THR}, %[thr]",
        f"s_mov_b32 {S_CUTROWS}, %[cutrows]",
        f"s_mov_b32 {S_SH}, 0",
        f"s_mov_b32 {S_CUT}, 0",
        f"s_mov_b32 {S_CHUNK}, 0",
        f"s_mov_b32 {S_EARLY}, 0",
        f"s_mov_b32 {S_PUSHROW}, %[pushrow]",
        f"s_mov_b32 {S_PUSHMAX}, %[pushmax]",
        f"s_mov_b32 {S_PUSHSOLID}, %[pushsolid]",
        f"s_mov_b32 {S_THRSOLID}, %[solidthr]",
    ]
    asm += [f"s_mov_b64 {S_DEAD[g]}, 0" for g in range(G)]
    asm += [f"s_mov_b64 {S_BASE[c]}, %[base{c}]" for c in range(5)]
    asm += [
        f"s_getpc_b64 {S_PC}",
        "L_anchor_%=:",
        f"s_add_u32 {S_BASE_LO}, {S_PC_LO}, (L_body0_%= - L_anchor_%=)",
        f"s_addc_u32 {S_BASE_HI}, {S_PC_HI}, 0",
        "s_waitcnt lgkmcnt(0)",
    ]
    asm += disp()
    def last_row_and_dispatch(c: int) -> list[str]:
        """The token's last row with the next token's dispatch computed UNDER it: the five scalar instructions that turn
        the next code into a jump target are a dependent chain (s_and -> s_mul -> s_add -> s_addc) followed by a taken
        branch; behind the row they are ~40 cycles in which this wave issues nothing for the vector pipe, and with four
        or five waves per SIMD that shows (two waves must be ready at any time to issue a vector instruction every other
        cycle).  Woven between the row's vector instructions they cost issue slots only; the row's shift counter moves up
        right behind the instructions that read it, so nothing but VALU sits between s_add and s_addc (SCC)."""
        rows = body.emit_asm(reg_for(c), c)
        if os.environ.get("BGSA_GEN_BANDED_WEAVE", "1") == "0":   # A/B builds: the dispatch behind the row
            return rows + [f"s_add_u32 {S_SH}, {S_SH}, 1"] + disp()
        n_head = G * (2 if wide else 1)                      # the window shifts read S_SH (the scheduler keeps them in front)
        assert all(S_SH in ln for ln in rows[:n_head]) and not any(S_SH in ln for ln in rows[n_head:])
        head, rest = rows[:n_head], rows[n_head:]
        out = head + [f"s_add_u32 {S_SH}, {S_SH}, 1"]
        d = disp()
        scalars, jump = d[:-1], d[-1]
        gap = max(1, (len(rest) - 2) // (len(scalars) + 1))
        k_next = 0
        for i, ln in enumerate(rest):
            out.append(ln)
            if k_next < len(scalars) and i >= 1 and (i - 1) % gap == 0:
                out.append(scalars[k_next])
                k_next += 1
        out += scalars[k_next:]
        out.append(jump)
        return out

    for a in range(5):          # slots 0..24: two rows per token
        for b2 in range(5):
            asm.append(f"L_body{5 * a + b2}_%=:")
            asm += body.emit_asm(reg_for(a), a)
            asm.append(f"s_add_u32 {S_SH}, {S_SH}, 1")
            asm += last_row_and_dispatch(b2)
    for c in range(5):          # slots 25..29: one row
        asm.append(f"L_body{25 + c}_%=:")
        asm += last_row_and_dispatch(c)
        asm.append(f"L_end{25 + c}_%=:")
        asm.append(pad(25 + c))
    asm.append("L_body30_%=:")  # END
    asm.append("s_branch L_done_%=")
    asm.append("L_end30_%=:")
    asm.append(pad(30))
    asm.append("L_body31_%=:")  # REFILL
    asm += [
        f"s_sub_u32 {S_LEFT}, {S_LEFT}, 1",
        "s_cbranch_scc1 L_done_%=",
        "s_waitcnt lgkmcnt(0)",
        f"s_mov_b64 {S_WIN}, {S_NXT}",
        f"s_add_u32 {S_PTR_LO}, {S_PTR_LO}, 8",
        f"s_addc_u32 {S_PTR_HI}, {S_PTR_HI}, 0",
        f"s_load_dwordx2 {S_NXT}, {S_PTR}, 0x8",
    ]
    asm += disp()
    asm.append("L_end31_%=:")
    asm.append(pad(31))
    for slot in range(32, 63):   # every value of the 6-bit dispatch mask that is no token: a fail slot
        asm.append(f"L_body{slot}_%=:")
        asm += fail_slot()
        asm.append(f"L_end{slot}_%=:")
        asm.append(pad(slot))
    asm.append("L_body63_%=:")  # EVENT <arg>: the last slot, so it may be longer than the slot stride
    asm += [
        f"s_and_b32 {S_ARG}, {S_WIN_LO}, 0xff",
        f"s_lshr_b64 {S_WIN}, {S_WIN}, 8",
        f"s_cmp_eq_u32 {S_ARG}, 4",           # the plain test, by far the most frequent event: its own short path
        "s_cbranch_scc0 L_ev_general_%=",
    ]
    asm += test("p") + alive()
    asm += ["s_cbranch_scc0 L_ev_alldead_%="]
    asm += push_or("L_ev_out_%=", "p")
    asm += ["L_ev_alldead_%=:"] + [f"s_mov_b64 {S_DEAD[g]}, exec" for g in range(G)] + ["s_branch L_done_%="]
    asm += [
        "L_ev_general_%=:",
        f"s_bitcmp1_b32 {S_ARG}, 2",          # bit 2: test; bit 3: latch the reject masks (the reference's last checkpoint)
        "s_cbranch_scc0 L_ev_reset_%=",
    ]
    asm += test("g") + alive()
    asm += ["s_cbranch_scc0 L_ev_alldead_%=",
            f"s_bitcmp1_b32 {S_ARG}, 3",
            "s_cbranch_scc0 L_ev_nolatch_%="]
    asm += latch_from_alive()
    asm += ["L_ev_nolatch_%=:"]
    asm += push_or("L_ev_reset_%=", "g")
    asm += [
        "L_ev_reset_%=:",
        f"s_bitcmp1_b32 {S_ARG}, 0",          # bit 0: scoring starts (row k)
        "s_cbranch_scc0 L_ev_adv_%=",
    ]
    asm += [f"v_mov_b32 %[s{a}], 0" for a in acc]
    asm += [
        "L_ev_adv_%=:",
        f"s_bitcmp1_b32 {S_ARG}, 1",          # bit 1: next 32 rows
        "s_cbranch_scc0 L_ev_cut_%=",
        "s_waitcnt vmcnt(0)",
    ]
    for g in range(G):       # (vmcnt(0): a word fetched by the previous advance that no cut has waited for yet)
        for w in range(n_m - 1):
            asm += [f"v_mov_b32 %[m{w}_{c}_{g}], %[m{w + 1}_{c}_{g}]" for c in range(5)]
    for g in range(G):
        for c in range(5):
            asm.append(f"global_load_dword %[m{n_m - 1}_{c}_{g}], %[voff{g}], {S_BASE[c]}")
    asm += [f"v_add_u32 %[voff{g}], 0x100, %[voff{g}]" for g in range(G)]
    asm += [
        f"s_mov_b32 {S_SH}, 0",
        f"s_mov_b32 {S_CUT}, 0",
        f"s_add_u32 {S_CHUNK}, {S_CHUNK}, 1",
        "L_ev_cut_%=:",
        f"s_bitcmp1_b32 {S_ARG}, 5",          # bit 5: the window of every class moves up by cutrows bits
        "s_cbranch_scc0 L_ev_out_%=",
        f"s_mov_b32 {S_SH}, 0",
        "s_waitcnt vmcnt(0)",                 # B, fetched by the last advance
        f"s_cmp_eq_u32 {S_CUT}, 0",
        "s_cbranch_scc0 L_ev_cut_again_%=",
    ]
    for g in range(G):       # the first cut behind an advance: B's low bits follow A's
        asm += [f"v_alignbit_b32 %[m0_{c}_{g}], %[m1_{c}_{g}], %[m0_{c}_{g}], {S_CUTROWS}" for c in range(5)]
    asm += ["s_branch L_ev_cut_done_%=", "L_ev_cut_again_%=:"]
    for g in range(G):       # a later one: the bits of B that earlier cuts took are gone first
        for c in range(5):
            asm += [f"v_lshrrev_b32 %[t0], {S_CUT}, %[m1_{c}_{g}]",
                    f"v_alignbit_b32 %[m0_{c}_{g}], %[t0], %[m0_{c}_{g}], {S_CUTROWS}"]
    asm += ["L_ev_cut_done_%=:", f"s_add_u32 {S_CUT}, {S_CUT}, {S_CUTROWS}", "L_ev_out_%=:"]
    asm += disp()
    asm += done("s_waitcnt vmcnt(0) lgkmcnt(0)")
    asm += [f"s_mov_b64 %[dead{g}], {S_DEAD[g]}" for g in range(G)]
    asm.append(f"s_mov_b32 %[early], {S_EARLY}")

    text = "\n".join(f'        "{line}\\n\\t"' if not line.endswith(":") else f'        "{line}\\n"' for line in asm)
    outs = [f'[s{i}] "+v"(state[{i}])' for i in range(n_state)]
    outs += [f'[m{w}_{c}_{g}] "+v"(M[{g}][{c}][{w}])' for g in range(G) for c in range(5) for w in range(n_m)]
    outs += [f'[voff{g}] "+v"(voff[{g}])' for g in range(G)]
    outs += [f'[dead{g}] "=s"(dead[{g}])' for g in range(G)]
    outs += ['[left] "=s"(left)', '[early] "=s"(early)']
    outs += [f'[t{i}] "=&v"(tmp[{i}])' for i in range(n_slots)]
    ins = ['[qp] "s"(stream)', '[nwin] "s"(n_windows)', '[vmask] "v"(band_mask)'] + (['[vmask_hi] "v"(band_mask_hi)'] if wide else []) + \
          ['[thr] "s"(limit)', '[cutrows] "s"(cut_rows)',
           '[pushrow] "s"(push_row)', '[pushmax] "s"(push_max)', '[pushsolid] "s"(push_row_solid)', '[solidthr] "s"(solid_limit)']
    ins += [f'[base{c}] "s"(base[{c}])' for c in range(5)]
    clob = ", ".join(f'"{x}"' for x in clobbers)
    if funnel:
        bits = 64 if wide else 32
        sfx = "s" if sh64 else ""
        note = f"  D0 >> 1 is ONE v_lshrrev_b64 on the fixed pair {P(0)}:{P(1)} (clobbered by name)." if sh64 else ""
        return f"""
// Funnel-shift banded rows ({bits}-bit band: thresholds {'16 .. 31' if wide else '13 .. 15'}), {G} subject group{'s' if G > 1 else ''} per wave: {body.valu_count()} VALU per row
// ({sum(op.kind in ('alignbit',) for op in body.ops)} funnel shifts among them), {n_slots} temporaries.  state[{per_state}g ..] = {{VP, VN, errors since row k}} of group g
// (VP lo / hi, VN lo / hi, errors when the band is a pair); M[g][c] = {n_m} consecutive 32-bit words of class c's offset match string, the last
// one the prefetch target; voff[g] = byte offset of the next word to fetch relative to base[c] (group 1: the group stride included);
// dead / left / early and the push rules as banded_cut_rows_asm_g{G} (cut_rows is not used: the stream carries no cut events).{note}
__device__ __forceinline__ void banded_funnel{bits}{sfx}_rows_asm_g{G}(uint32_t (&state)[{n_state}], uint32_t (&M)[{G}][5][{n_m}], uint32_t (&voff)[{G}],
                                                       const unsigned long long (&base)[5],
                                                       const unsigned long long stream, const int n_windows,
                                                       const uint32_t band_mask, {'const uint32_t band_mask_hi, ' if wide else ''}const uint32_t cut_rows,
                                                       const uint32_t limit, const uint32_t push_row, const uint32_t push_row_solid, const uint32_t solid_limit,
                                                       const uint32_t push_max, unsigned long long (&dead)[{G}],
                                                       int &left, int &early)
{{
    uint32_t tmp[{max(n_slots, 1)}];
    asm volatile(
{text}
        : {", ".join(outs)}
        : {", ".join(ins)}
        : {clob});
}}
"""
    return f"""
// One-word-window banded rows, {G} subject group{'s' if G > 1 else ''} per wave: {body.valu_count()} VALU per row, all fast class
// ({sum(op.kind in ('alignbit',) for op in body.ops)} funnel shifts in the row; the cut event holds them), {n_slots} temporaries.
// state[3g..3g+2] = {{VP, VN, errors since row k}} of group g; M[g][c] = {{window (in: word 0), the word behind it (in: word 1)}}
// of class c's offset match string; voff[g] = byte offset of the next word to fetch (in: word 2) relative
// to base[c] (group 1: the group stride included); dead[g] = reject mask of group g (lanes whose error count passed `limit` at the last
// checkpoint, or all lanes if the wave stopped with every lane of every group past it); left / early as banded_rows_asm32
// (early: the lanes NOT in dead[] go to the regroup list; from row push_row on whenever 1..push_max lanes are within the
// limit, from row push_row_solid on if one of them has at most solid_limit errors since row k).
__device__ __forceinline__ void banded_cut_rows_asm_g{G}(uint32_t (&state)[{n_state}], uint32_t (&M)[{G}][5][2], uint32_t (&voff)[{G}],
                                                       const unsigned long long (&base)[5],
                                                       const unsigned long long stream, const int n_windows,
                                                       const uint32_t band_mask, const uint32_t cut_rows,
                                                       const uint32_t limit, const uint32_t push_row, const uint32_t push_row_solid, const uint32_t solid_limit,
                                                       const uint32_t push_max, unsigned long long (&dead)[{G}],
                                                       int &left, int &early)
{{
    uint32_t tmp[{max(n_slots, 1)}];
    asm volatile(
{text}
        : {", ".join(outs)}
        : {", ".join(ins)}
        : {clob});
}}
"""


def gen_banded_chunk_function() -> str:
    """Banded row loop for the 32-bit band (k <= 15) WITHOUT a per-row jump: straight-line code for 32 rows, the
    query character selecting the match words through an LDS ADDRESS instead of a branch.

    The threaded loop (gen_banded_function) pays 6 scalar instructions of dispatch per token plus a shift counter per
    row, events and refills — 7.85 SALU per wave-row against 13 VALU, and a CU has ONE scalar unit for its four SIMDs:
    4 x 7.85 scalar issue slots per ~31-cycle row saturate it (adding four dummy SALU per row cost 43 %).
    Here the wave keeps its match words in LDS ([class][slot lo/hi/next][lane]); a row's token is the byte offset of its
    class (one SGPR per row, 32 loaded at a time with two s_load_dwordx16), the row reads its two words with
    ds_read_b32 at `lane base + token` two rows ahead, and the funnel shift amount is the row's position in the chunk —
    an immediate.  Per row: 13 VALU, 2 LDS reads, and no scalar instruction at all in the chunks that hold no special
    row; a chunk that holds row k (error count starts), the last checkpoint (reject mask latched) or the query's end
    runs a second copy of the 32 rows with a two-instruction check behind every row.  Tests every 8 rows are inline,
    enabled by a per-chunk mask; every 32 rows the words move down one slot and the next one, fetched from global
    memory during the chunk, takes the free slot."""
    body = R.banded_body()
    slot_of, n_slots = body.allocate_temps()
    TOK0 = 60                       # s[60:91]: the 32 row tokens of the current chunk
    S_RET, S_RET_LO, S_RET_HI = "s[92:93]", "s92", "s93"
    clobbers = [f"s{i}" for i in range(60, 94)] + ["vcc", "scc", "memory"]
    NM = 3

    def reg_row(j: int):
        pipe = j % 3
        def reg(name: str) -> str:
            if name.startswith("S"):
                return f"%[s{name[1:]}]"
            if name == "E0":
                return f"%[lo{pipe}]"
            if name == "E1":
                return f"%[hi{pipe}]"
            if name in ("$mask", "$mask_lo"):
                return "%[mask]"
            if name == "$sh":
                return str(j)
            if name == "$one":
                return "1"
            return f"%[t{slot_of[name]}]"
        return reg

    def fetch(j: int) -> list[str]:
        pipe = j % 3
        return [f"v_add_u32 %[vaddr], s{TOK0 + j}, %[lanebase]",
                f"ds_read_b32 %[lo{pipe}], %[vaddr]",
                f"ds_read_b32 %[hi{pipe}], %[vaddr] offset:256"]

    def rows(tag: str, checked: bool) -> list[str]:
        out = fetch(0) + fetch(1)
        for j in range(32):
            if j + 2 < 32:
                out += fetch(j + 2)
            out.append(f"s_waitcnt lgkmcnt({2 * min(2, 31 - j)})")
            out += body.emit_asm(reg_row(j), 0)
            if j % 8 == 7:   # a test after every 8th row, if this chunk's mask enables it
                idx = j // 8
                out += [
                    f"s_bitcmp1_b32 %[testen], {idx}",
                    f"s_cbranch_scc0 L_{tag}_nt{idx}_%=",
                    "v_cmp_lt_u32 vcc, %[thr], %[s2]",
                    "s_andn2_b64 %[alive], exec, vcc",
                    "s_cbranch_scc0 L_alldead_%=",
                    # few survivors late enough: hand them to the regroup list (banded.hip)
                    f"s_add_u32 %[cnt], %[row0], {j + 1}",
                    "s_cmp_ge_u32 %[cnt], %[pushrow]",
                    f"s_cbranch_scc0 L_{tag}_nt{idx}_%=",
                    "s_bcnt1_i32_b64 %[cnt], %[alive]",
                    "s_cmp_le_u32 %[cnt], %[pushmax]",
                    f"s_cbranch_scc0 L_{tag}_nt{idx}_%=",
                    "s_mov_b64 %[dead], vcc",
                    "s_mov_b32 %[early], 1",
                    "s_branch L_done_%=",
                    f"L_{tag}_nt{idx}_%=:",
                ]
            if checked:
                out += [f"s_bitcmp1_b32 %[ev], {j}", f"s_cbranch_scc1 L_sp{j}_%=", f"L_ret{j}_%=:"]
        return out

    asm = [
        "s_mov_b32 %[early], 0",
        "s_mov_b64 %[dead], 0",
        "s_mov_b32 %[row0], 0",
        "L_chunk_%=:",
        # this chunk's tokens; the next match word of every class, due when the chunk ends
        f"s_load_dwordx16 s[{TOK0}:{TOK0 + 15}], %[tokbase], %[tokoff]",
        "s_add_u32 %[cnt], %[tokoff], 64",
        f"s_load_dwordx16 s[{TOK0 + 16}:{TOK0 + 31}], %[tokbase], %[cnt]",
        "s_add_u32 %[tokoff], %[tokoff], 128",
    ]
    asm += [f"global_load_dword %[next{c}], %[voff{c}], %[gbase]" for c in range(5)]
    asm += [f"v_add_u32 %[voff{c}], 0x100, %[voff{c}]" for c in range(5)]
    # special rows of this chunk: bit j of `ev` = something happens once row j is done (rows done = row0 + j + 1)
    asm += ["s_mov_b32 %[ev], 0"]
    for what in ("k", "last", "len"):
        asm += [
            f"s_sub_u32 %[cnt], %[{what}], %[row0]",
            "s_sub_u32 %[cnt], %[cnt], 1",
            "s_cmp_lt_u32 %[cnt], 32",                 # unsigned: also false when the row lies before this chunk
            f"s_cbranch_scc0 L_no_{what}_%=",
            "s_lshl_b32 %[cnt], 1, %[cnt]",
            "s_or_b32 %[ev], %[ev], %[cnt]",
            f"L_no_{what}_%=:",
        ]
    # tests after rows 8, 16, 24, 32 of the chunk: enabled while k < rows done <= last
    asm += ["s_mov_b32 %[testen], 0"]
    for idx in range(4):
        asm += [
            f"s_add_u32 %[cnt], %[row0], {8 * (idx + 1)}",
            "s_cmp_gt_u32 %[cnt], %[k]",
            f"s_cbranch_scc0 L_te{idx}_%=",
            "s_cmp_le_u32 %[cnt], %[last]",
            f"s_cbranch_scc0 L_te{idx}_%=",
            f"s_bitset1_b32 %[testen], {idx}",
            f"L_te{idx}_%=:",
        ]
    asm += [
        "s_waitcnt lgkmcnt(0)",
        "s_cmp_eq_u32 %[ev], 0",
        "s_cbranch_scc0 L_checked_%=",
    ]
    asm += rows("f", False)
    asm.append("s_branch L_advance_%=")
    asm.append("L_checked_%=:")
    asm += rows("c", True)
    asm.append("L_advance_%=:")
    asm.append("s_add_u32 %[row0], %[row0], 32")
    # the words move down one slot; the one fetched during the chunk takes the free slot
    for c in range(5):
        base = c * NM * 256
        asm += [f"ds_read_b32 %[lo0], %[lanebase] offset:{base + 256}",
                f"ds_read_b32 %[hi0], %[lanebase] offset:{base + 512}",
                "s_waitcnt lgkmcnt(0)",
                f"ds_write_b32 %[lanebase], %[lo0] offset:{base}",
                f"ds_write_b32 %[lanebase], %[hi0] offset:{base + 256}"]
    asm.append("s_waitcnt vmcnt(0)")
    for c in range(5):
        asm.append(f"ds_write_b32 %[lanebase], %[next{c}] offset:{c * NM * 256 + 512}")
    asm += ["s_waitcnt lgkmcnt(0)", "s_branch L_chunk_%="]
    # ---- out of line: what a special row does, then back behind that row
    for j in range(32):
        asm += [f"L_sp{j}_%=:", f"s_mov_b32 %[jreg], {j}", "s_branch L_special_%="]
    asm += [
        "L_special_%=:",
        "s_add_u32 %[cnt], %[row0], %[jreg]",
        "s_add_u32 %[cnt], %[cnt], 1",                  # rows done
        "s_cmp_eq_u32 %[cnt], %[last]",                 # the reference's last checkpoint: latch the reject mask
        "s_cbranch_scc0 L_sp_nolatch_%=",
        "v_cmp_lt_u32 vcc, %[thr], %[s2]",
        "s_mov_b64 %[dead], vcc",
        "L_sp_nolatch_%=:",
        "s_cmp_eq_u32 %[cnt], %[k]",                    # scoring starts at row k
        "s_cbranch_scc0 L_sp_noreset_%=",
        "v_mov_b32 %[s2], 0",
        "L_sp_noreset_%=:",
        "s_cmp_eq_u32 %[cnt], %[len]",
        "s_cbranch_scc1 L_done_%=",
        f"s_getpc_b64 {S_RET}",
        "L_tabanchor_%=:",
        "s_lshl_b32 %[cnt], %[jreg], 2",
        f"s_add_u32 {S_RET_LO}, {S_RET_LO}, %[cnt]",
        f"s_addc_u32 {S_RET_HI}, {S_RET_HI}, 0",
        f"s_add_u32 {S_RET_LO}, {S_RET_LO}, (L_rettab_%= - L_tabanchor_%=)",
        f"s_addc_u32 {S_RET_HI}, {S_RET_HI}, 0",
        f"s_setpc_b64 {S_RET}",
        "L_rettab_%=:",
    ]
    asm += [f"s_branch L_ret{j}_%=" for j in range(32)]
    asm += [
        "L_alldead_%=:",
        "s_mov_b64 %[dead], exec",
        "L_done_%=:",
        "s_waitcnt vmcnt(0) lgkmcnt(0)",
    ]

    text = "\n".join(f'        "{line}\\n\\t"' if not line.endswith(":") else f'        "{line}\\n"' for line in asm)
    outs = [f'[s{i}] "+v"(state[{i}])' for i in range(3)]
    outs += [f'[voff{c}] "+v"(voff[{c}])' for c in range(5)]
    outs += ['[tokoff] "+s"(tokoff)', '[dead] "=&s"(dead)', '[early] "=&s"(early)', '[row0] "=&s"(row0)', '[ev] "=&s"(ev)',
             '[testen] "=&s"(testen)', '[cnt] "=&s"(cnt)', '[jreg] "=&s"(jreg)', '[alive] "=&s"(alive)']
    outs += [f'[lo{i}] "=&v"(lo[{i}])' for i in range(3)] + [f'[hi{i}] "=&v"(hi[{i}])' for i in range(3)]
    outs += [f'[next{c}] "=&v"(nxt[{c}])' for c in range(5)]
    outs += ['[vaddr] "=&v"(vaddr)']
    outs += [f'[t{i}] "=&v"(tmp[{i}])' for i in range(n_slots)]
    ins = ['[lanebase] "v"(lanebase)', '[tokbase] "s"(tokens)', '[gbase] "s"(gbase)', '[mask] "v"(band_mask)', '[thr] "s"(limit)',
           '[k] "s"(k)', '[last] "s"(last)', '[len] "s"(len)', '[pushrow] "s"(push_row)', '[pushmax] "s"(push_max)']
    clob = ", ".join(f'"{x}"' for x in clobbers)
    return f"""
// Straight-line banded rows, 32-bit band (gen_rows_asm.py: gen_banded_chunk_function): {body.valu_count()} + 1 VALU and two LDS reads per
// row, no scalar instruction per row outside the chunks that hold row k, the last checkpoint or the query's end.
// state = {{VP, VN, errors since row k}}; lanebase = LDS byte address of this lane's dword in the wave's match-word block
// ([class][slot][lane], slots = words of the current chunk: lo, hi, next); tokens + tokoff = this query's row tokens
// (one dword per row = class * 768, 32 per chunk, zero-padded to whole chunks); voff[c] = byte offset of class c's next
// word to fetch relative to gbase (the group's Mext block).  Returns the reject mask; early as banded_rows_asm32.
__device__ __forceinline__ unsigned long long banded_chunk_rows_asm32(uint32_t (&state)[3], uint32_t (&voff)[5], const uint32_t lanebase,
                                                                   const unsigned long long tokens, uint32_t tokoff,
                                                                   const unsigned long long gbase, const uint32_t band_mask,
                                                                   const uint32_t limit, const uint32_t k, const uint32_t last,
                                                                   const uint32_t len, const uint32_t push_row, const uint32_t push_max,
                                                                   int &early_out)
{{
    uint32_t tmp[{max(n_slots, 1)}], lo[3], hi[3], nxt[5], vaddr, row0, ev, testen, cnt, jreg;
    int early;
    unsigned long long dead, alive;
    asm volatile(
{text}
        : {", ".join(outs)}
        : {", ".join(ins)}
        : {clob});
    early_out = early;
    return dead;
}}
"""


def gen_blocked_function(fn_name: str, nw: int, body: R.Body, n_base: int, n_chains: int, n_planes: int, n_eq: int) -> str:
    """Row loop of a column-block kernel: one block of a long subject.  Stream code 7 (no
    argument) every 32 rows = CARRY: store the carry-out words of the finished 32 rows to the
    wave's carry buffer, fetch the carry-in words of the next 32 rows.  State = n_base block
    registers, then n_chains carry-in words, then n_chains carry-out words."""
    slot_of, n_slots = body.allocate_temps()
    S_CB = "s[80:81]"
    clobbers = CLOBBERS[:-3] + ["s80", "s81", "vcc", "scc", "memory"]
    cin = [n_base + i for i in range(n_chains)]
    cout = [n_base + n_chains + i for i in range(n_chains)]

    def reg_for(c: int):
        def reg(name: str) -> str:
            if name.startswith("S"):
                return f"%[s{name[1:]}]"
            if name.startswith("B"):
                return f"%[b{name[1:]}]"
            if name.startswith("E"):
                return f"%[e{c}_{name[1:]}]"
            return f"%[t{slot_of[name]}]"
        return reg

    asm = [
        f"s_mov_b64 {S_PTR}, %[qp]",
        f"s_mov_b32 {S_LEFT}, %[nwin]",
        f"s_load_dwordx2 {S_WIN}, {S_PTR}, 0x0",
        f"s_load_dwordx2 {S_NXT}, {S_PTR}, 0x8",
        f"s_mov_b64 {S_CB}, %[cbase]",
        f"s_getpc_b64 {S_PC}",
        "L_anchor_%=:",
        f"s_add_u32 {S_BASE_LO}, {S_PC_LO}, (L_body0_%= - L_anchor_%=)",
        f"s_addc_u32 {S_BASE_HI}, {S_PC_HI}, 0",
        "s_waitcnt lgkmcnt(0)",
    ]
    asm += dispatch()
    for c in range(5):
        asm.append(f"L_body{c}_%=:")
        asm += body.emit_asm(reg_for(c), c)
        asm += dispatch()
    asm.append("L_body5_%=:")
    asm.append("s_branch L_done_%=")
    asm.append(".fill ((L_body1_%= - L_body0_%=) - 4) / 4, 4, 0xbf800000")
    asm.append("L_body6_%=:")
    asm += [
        f"s_sub_u32 {S_LEFT}, {S_LEFT}, 1",
        "s_cbranch_scc1 L_done_%=",
        "s_waitcnt lgkmcnt(0)",
        f"s_mov_b64 {S_WIN}, {S_NXT}",
        f"s_add_u32 {S_PTR_LO}, {S_PTR_LO}, 8",
        f"s_addc_u32 {S_PTR_HI}, {S_PTR_HI}, 0",
        f"s_load_dwordx2 {S_NXT}, {S_PTR}, 0x8",
    ]
    asm += dispatch()
    asm.append("L_refill_end_%=:")
    asm.append(".fill ((L_body1_%= - L_body0_%=) - (L_refill_end_%= - L_body6_%=)) / 4, 4, 0xbf800000")
    asm.append(f"L_body7_%=:")  # CARRY: chunk j out, chunk j+1 in ([chunk][chain][64 lanes] dwords)
    # (the instruction offset is 13-bit signed: beyond 15 chains voff itself moves on in between)
    moved = 0
    for i in range(n_chains):
        if 256 * i - moved > 3840:
            asm.append(f"v_add_u32 %[voff], 0x{256 * i - moved:x}, %[voff]")
            moved = 256 * i
        asm.append(f"global_store_dword %[voff], %[s{cout[i]}], {S_CB} offset:{256 * i - moved}")
    asm.append(f"v_add_u32 %[voff], 0x{256 * n_chains - moved:x}, %[voff]")
    moved = 0
    for i in range(n_chains):
        if 256 * i - moved > 3840:
            asm.append(f"v_add_u32 %[voff], 0x{256 * i - moved:x}, %[voff]")
            moved = 256 * i
        asm.append(f"global_load_dword %[s{cin[i]}], %[voff], {S_CB} offset:{256 * i - moved} sc1")
    if moved:
        asm.append(f"v_add_u32 %[voff], 0x{(-moved) & 0xFFFFFFFF:x}, %[voff]")
    asm.append("s_waitcnt vmcnt(0)")
    asm += dispatch()
    asm += done("s_waitcnt vmcnt(0) lgkmcnt(0)")   # codes 0..7 all exist here (7 = CARRY)

    n_state = n_base + 2 * n_chains
    text = "\n".join(f'        "{line}\\n\\t"' if not line.endswith(":") else f'        "{line}\\n"' for line in asm)
    outs = [f'[s{i}] "+v"(state[{i}])' for i in range(n_state)]
    outs += ['[voff] "+v"(voff)', '[left] "=s"(left)']
    outs += [f'[t{i}] "=&v"(tmp[{i}])' for i in range(n_slots)]
    if n_planes:
        ins = [f'[b{j}] "v"(B[{j}])' for j in range(n_planes)]
        masks_param = f"const uint32_t (&B)[{n_planes}]"
    else:
        ins = [f'[e{c}_{j}] "v"(P[{c}][{j}])' for c in range(5) for j in range(n_eq)]
        masks_param = f"const uint32_t (&P)[5][{n_eq}]"
    ins += ['[qp] "s"(stream)', '[nwin] "s"(n_windows)', '[cbase] "s"(carry_base)']
    clob = ", ".join(f'"{x}"' for x in clobbers)
    nops = sum(line.startswith("s_nop") for line in body.emit_asm(lambda x: x, 0))
    return f"""
// {body.valu_count()} VALU per row, {n_slots} temporaries, {n_chains} carry chains, {nops} hazard nops
template <>
__device__ __forceinline__ int {fn_name}<{nw}>(uint32_t (&state)[{n_state}], {masks_param},
                                                       uint32_t &voff, const unsigned long long carry_base,
                                                       const unsigned long long stream, const int n_windows)
{{
    uint32_t tmp[{max(n_slots, 1)}];
    int left;
    asm volatile(
{text}
        : {", ".join(outs)}
        : {", ".join(ins)}
        : {clob});
    return left;
}}
"""


def gen_packed_blocked_function(fn_name: str, nw: int, body: R.Body, n_base: int, n_words: int, n_eq: int) -> str:
    """Row loop of a packed-carry column-block kernel (rows_ir.make_blocked_packed): the carries of one row are bits of
    n_words words per direction, exchanged with the neighbouring blocks EVERY row.  Plain stream (codes 0..4, END,
    REFILL; no CARRY token).  State = n_base block registers, n_words carry-in words, n_words carry-out words; operand
    `nxt` = the carry-in words of the NEXT row, fetched while this row is computed.  Per row: wait for the words
    fetched one row ago, move them in, fetch the next row's, run the body, store the carry-out words over this row's
    carry-in words (the block to the right reads them there), advance the row offset.  The buffer holds one row more
    than the query has, so the last row's fetch stays inside it."""
    slot_of, n_slots = body.allocate_temps()
    S_CB = "s[80:81]"
    clobbers = CLOBBERS[:-3] + ["s80", "s81", "vcc", "scc", "memory"]
    cin = [n_base + j for j in range(n_words)]
    cout = [n_base + n_words + j for j in range(n_words)]
    row_bytes = 256 * n_words

    def reg_for(c: int):
        def reg(name: str) -> str:
            if name.startswith("S"):
                return f"%[s{name[1:]}]"
            if name.startswith("E"):
                return f"%[e{c}_{name[1:]}]"
            return f"%[t{slot_of[name]}]"
        return reg

    def row(c: int) -> list[str]:
        pre = ["s_waitcnt vmcnt(0)"]
        pre += [f"v_mov_b32 %[s{cin[j]}], %[n{j}]" for j in range(n_words)]
        pre += [f"global_load_dword %[n{j}], %[voff], {S_CB} offset:{row_bytes + 256 * j} sc1" for j in range(n_words)]
        post = [f"global_store_dword %[voff], %[s{cout[j]}], {S_CB} offset:{256 * j}" for j in range(n_words)]
        post.append(f"v_add_u32 %[voff], 0x{row_bytes:x}, %[voff]")
        return pre + body.emit_asm(reg_for(c), c) + post + dispatch()

    asm = [
        f"s_mov_b64 {S_PTR}, %[qp]",
        f"s_mov_b32 {S_LEFT}, %[nwin]",
        f"s_load_dwordx2 {S_WIN}, {S_PTR}, 0x0",
        f"s_load_dwordx2 {S_NXT}, {S_PTR}, 0x8",
        f"s_mov_b64 {S_CB}, %[cbase]",
        f"s_getpc_b64 {S_PC}",
        "L_anchor_%=:",
        f"s_add_u32 {S_BASE_LO}, {S_PC_LO}, (L_body0_%= - L_anchor_%=)",
        f"s_addc_u32 {S_BASE_HI}, {S_PC_HI}, 0",
        "s_waitcnt lgkmcnt(0)",
    ]
    asm += dispatch()
    for c in range(5):
        asm.append(f"L_body{c}_%=:")
        asm += row(c)
    asm.append("L_body5_%=:")
    asm.append("s_branch L_done_%=")
    asm.append(".fill ((L_body1_%= - L_body0_%=) - 4) / 4, 4, 0xbf800000")
    asm.append("L_body6_%=:")
    asm += [
        f"s_sub_u32 {S_LEFT}, {S_LEFT}, 1",
        "s_cbranch_scc1 L_done_%=",
        "s_waitcnt lgkmcnt(0)",
        f"s_mov_b64 {S_WIN}, {S_NXT}",
        f"s_add_u32 {S_PTR_LO}, {S_PTR_LO}, 8",
        f"s_addc_u32 {S_PTR_HI}, {S_PTR_HI}, 0",
        f"s_load_dwordx2 {S_NXT}, {S_PTR}, 0x8",
    ]
    asm += dispatch()
    asm.append("L_refill_end_%=:")
    asm.append(".fill ((L_body1_%= - L_body0_%=) - (L_refill_end_%= - L_body6_%=)) / 4, 4, 0xbf800000")
    asm.append("L_body7_%=:")
    asm += fail_slot()
    asm += done("s_waitcnt vmcnt(0) lgkmcnt(0)")

    n_state = n_base + 2 * n_words
    text = "\n".join(f'        "{line}\\n\\t"' if not line.endswith(":") else f'        "{line}\\n"' for line in asm)
    outs = [f'[s{i}] "+v"(state[{i}])' for i in range(n_state)]
    outs += [f'[n{j}] "+v"(next_in[{j}])' for j in range(n_words)]
    outs += ['[voff] "+v"(voff)', '[left] "=s"(left)']
    outs += [f'[t{i}] "=&v"(tmp[{i}])' for i in range(n_slots)]
    ins = [f'[e{c}_{j}] "v"(P[{c}][{j}])' for c in range(5) for j in range(n_eq)]
    ins += ['[qp] "s"(stream)', '[nwin] "s"(n_windows)', '[cbase] "s"(carry_base)']
    clob = ", ".join(f'"{x}"' for x in clobbers)
    return f"""
// {body.valu_count()} VALU per row, {n_slots} temporaries, {n_words} packed carry word(s) per direction
template <>
__device__ __forceinline__ int {fn_name}<{nw}>(uint32_t (&state)[{n_state}], const uint32_t (&P)[5][{n_eq}],
                                                       uint32_t (&next_in)[{n_words}], uint32_t &voff,
                                                       const unsigned long long carry_base,
                                                       const unsigned long long stream, const int n_windows)
{{
    uint32_t tmp[{max(n_slots, 1)}];
    int left;
    asm volatile(
{text}
        : {", ".join(outs)}
        : {", ".join(ins)}
        : {clob});
    return left;
}}
"""


class BitpalDomainError(ValueError):
    """A score set whose kernels do not fit the register file: says which quantity is over which budget."""


def bitpal_widths(sc: R.BitpalScores) -> tuple[list[int], list[int], bool]:
    """Kernel widths for one score set: (plain widths, column-block widths, packed).

    Plain kernels (state in registers for the whole subject) for 1..P words, P the widest that fits the VGPR budget (at
    most 12; measured for 2/-3/-5: 9-11 words at two waves per SIMD run at 27.8-28.5 TCUPS against 24.5 for the same
    subjects as column blocks).  Column-block kernels for longer subjects: the four widths up to W <= 8 words in the
    per-chain carry form (a block also carries 2 x chains carry words — round 2's only form), or, for score sets
    with so many chains that not even a one-word block fits that way (match - mismatch >= ~20: chains = 1 + (M - I - 1) +
    bits(M - 2G)), in the PACKED carry form (rows_ir.make_blocked_packed: ceil(chains / 32) words per direction
    whatever the set).  What bounds the domain then is the row body itself: it keeps M - I one-hot class masks and as
    many incoming-value masks of one word alive at once, so a one-word body needs about 2 (M - I) + bits(M - 2G) + 16
    registers — M - I up to ~100 fits.  Raises BitpalDomainError beyond that."""
    def plain_regs(nw):
        return sc.planes * nw + 5 * nw + R.bitpal_body(nw, sc).allocate_temps()[1]

    def block_regs(nw):
        body, _ = R.bitpal_block_body(nw, sc)
        asm_operands = sc.planes * nw + 5 * nw + body.allocate_temps()[1] + 2 * sc.chains + 2
        # what hipcc adds around the asm block (addresses, the carry words' stores after the loop):
        # measured 24 / 39 / 44 / 75 registers for 3 / 10 / 13 / 22 chains
        return asm_operands + 20 + (5 * sc.chains + 1) // 2

    def packed_regs(nw):
        body, _, n_words = R.bitpal_packed_block_body(nw, sc)
        return sc.planes * nw + 5 * nw + body.allocate_temps()[1] + 3 * n_words + 2 + 24

    def fitting(regs, budget, upto):
        """Widths 1..upto whose register need is within the budget (the need grows with the width: stop at the first miss)."""
        out = []
        for nw in range(1, upto + 1):
            if regs(nw) > budget:
                break
            out.append(nw)
        return out

    fits_plain = fitting(plain_regs, BITPAL_VGPR_BUDGET, 12)
    if not fits_plain:
        raise BitpalDomainError(
            f"BitPAl {sc.match}/{sc.mismatch}/{sc.gap}: the row body of ONE word needs {plain_regs(1)} VGPRs "
            f"(budget {BITPAL_VGPR_BUDGET}): match - mismatch = {sc.K} value classes (two masks each) and "
            f"bits(match - 2 gap) = {sc.nb} planes are over what a wave can hold")
    plain = max(fits_plain)
    fits_block = fitting(block_regs, BITPAL_BLOCK_VGPR_BUDGET, 8)
    fits_packed = fitting(packed_regs, BITPAL_VGPR_BUDGET, 8)
    # the per-chain form (carry words exchanged every 32 rows) wherever it reaches four-word blocks; below that the packed
    # form if it offers WIDER blocks (fewer blocks per subject): with one chain per class (round 4) 10/-9/-15 has 25 chains
    # and one- and two-word blocks fit per chain, three-word ones packed
    if fits_block and (not fits_packed or max(fits_block) >= min(4, max(fits_packed))):
        wide = max(fits_block)
        return list(range(1, plain + 1)), list(range(max(1, wide - 3), wide + 1)), False
    if not fits_packed:
        raise BitpalDomainError(
            f"BitPAl {sc.match}/{sc.mismatch}/{sc.gap}: no column-block kernel fits: one word needs {packed_regs(1)} VGPRs "
            f"in the packed carry form (budget {BITPAL_VGPR_BUDGET}; match - mismatch = {sc.K}, {sc.chains} carry chains)")
    wide = max(fits_packed)
    return list(range(1, plain + 1)), list(range(max(1, wide - 3), wide + 1)), True


def bitpal_inc_text(sc: R.BitpalScores) -> str:
    """The generated header of one BitPAl score set: constants + row loops of every width."""
    B, NC = sc.planes, sc.chains
    plain, blocks, packed = bitpal_widths(sc)
    NCW = (NC + 31) // 32
    per_word = R.bitpal_body(1, sc).valu_count()
    case = lambda ws: " ".join(f"X({w})" for w in ws)
    parts = ["// GENERATED by gen_rows_asm.py from rows_ir.py — do not edit.\n",
             f"// BitPAl packed, match {sc.match} / mismatch {sc.mismatch} / gap {sc.gap}: {B} planes per word (u = dH - gap, unsigned,\n"
             f"// values 0..{sc.C}), {sc.K} value classes above the mismatch class {sc.D}, {NC} carry chains,\n"
             f"// {per_word} VALU per (row, word).\n"
             f"constexpr int kBitpalMatch = {sc.match}, kBitpalMismatch = {sc.mismatch}, kBitpalGap = {sc.gap};\n"
             f"constexpr int kBitpalPlanes = {B};\n"
             f"constexpr int kBitpalChains = {NC};\n"
             f"constexpr int kBitpalValuPerWord = {per_word};\n"
             f"constexpr int kBitpalMaxPlain = {plain[-1]};   // widest kernel that keeps the whole subject in registers\n"
             f"constexpr int kBitpalBlockMin = {blocks[0]}, kBitpalBlockMax = {blocks[-1]};   // column-block widths\n"
             f"constexpr bool kBitpalPackedBlocks = {'true' if packed else 'false'};   // carries of a row packed into kBitpalCarryWords words (many chains)\n"
             f"constexpr int kBitpalCarryWords = {NCW};\n"
             f"constexpr int kBitpalWeights[{B}] = {{{', '.join(str(x) for x in sc.weights())}}};   // score weight of a set bit per plane\n"
             f"#define BGSA_BITPAL_PLAIN_WIDTHS(X) {case(plain)}\n"
             f"#define BGSA_BITPAL_BLOCK_WIDTHS(X) {case(blocks)}\n"
             "// All rows of one query against one group.  state[w*kBitpalPlanes+i] = plane i of word w\n"
             "// (weight 2^i); P[c][w] = match mask of character class c.\n"
             "template <int NW>\n"
             "__device__ __forceinline__ int bitpal_rows_asm(uint32_t (&state)[kBitpalPlanes * NW],\n"
             "                                                const uint32_t (&P)[5][NW],\n"
             "                                                const unsigned long long stream, const int n_windows);\n"]
    for nw in plain:
        parts.append(gen_function("bitpal_rows_asm", f"{nw}", R.bitpal_body(nw, sc), B * nw, nw))
    parts.append("\n// One column block of a subject too long for the plain kernels (rows_ir.py: make_blocked(bitpal_body)).\n"
                 "// state = planes x NW, then the carry-in words, then the carry-out words; voff / carry_base as\n"
                 "// in myers_block_rows_asm ([32-row chunk][chain][64 lanes] dwords).\n"
                 "template <int NW>\n"
                 "__device__ __forceinline__ int bitpal_block_rows_asm(uint32_t (&state)[kBitpalPlanes * NW + 2 * kBitpalChains],\n"
                 "                                                      const uint32_t (&P)[5][NW], uint32_t &voff,\n"
                 "                                                      const unsigned long long carry_base,\n"
                 "                                                      const unsigned long long stream, const int n_windows);\n")
    parts.append("\n// The same with the carries of a row packed into kBitpalCarryWords words per direction, exchanged every row\n"
                 "// (rows_ir.py: make_blocked_packed): for score sets with too many chains for a register pair each.\n"
                 "// state = planes x NW, carry-in words, carry-out words; next_in = the next row's carry-in words;\n"
                 "// carry buffer = [row][word][64 lanes] dwords at carry_base, voff = this lane's byte offset of the current row.\n"
                 "template <int NW>\n"
                 "__device__ __forceinline__ int bitpal_packed_block_rows_asm(uint32_t (&state)[kBitpalPlanes * NW + 2 * kBitpalCarryWords],\n"
                 "                                                             const uint32_t (&P)[5][NW], uint32_t (&next_in)[kBitpalCarryWords],\n"
                 "                                                             uint32_t &voff, const unsigned long long carry_base,\n"
                 "                                                             const unsigned long long stream, const int n_windows);\n")
    for nw in blocks:
        if packed:
            body, init, n_words = R.bitpal_packed_block_body(nw, sc)
            assert len(init) == NC and not any(init) and n_words == NCW
            parts.append(gen_packed_blocked_function("bitpal_packed_block_rows_asm", nw, body, B * nw, n_words, nw))
        else:
            blocked, init = R.bitpal_block_body(nw, sc)
            assert len(init) == NC and not any(init)  # every BitPAl chain starts with carry-in 0
            parts.append(gen_blocked_function("bitpal_block_rows_asm", nw, blocked, B * nw, NC, 0, nw))
    return "".join(parts)


def main() -> int:
    # `--out DIR`: write the three headers there instead of beside this script (tests/test_generated_inc_cpu.py compares them
    # with the committed ones)
    here = Path(sys.argv[sys.argv.index("--out") + 1]) if "--out" in sys.argv else Path(__file__).resolve().parent
    head = "// GENERATED by gen_rows_asm.py from rows_ir.py — do not edit.\n"
    # ---- Myers --------------------------------------------------------------------------------
    parts = [head,
             "// All rows of one query against the G groups of one wave.  state[(g*NW+w)*2] = VP,\n"
             "// +1 = VN; P[c][g*NW+w] = match mask of character class c; `stream` = device address\n"
             "// (8-byte aligned, wave-uniform) of the packed query stream; n_windows = windows the stream\n"
             "// holds minus one = REFILLs a well-formed stream performs (the loop never does more).\n"
             "template <int NW, int G>\n"
             "__device__ __forceinline__ int myers_rows_asm(uint32_t (&state)[2 * G * NW],\n"
             "                                               const uint32_t (&P)[5][G * NW],\n"
             "                                               const unsigned long long stream, const int n_windows);\n"]
    for nw in MYERS_NW:  # G = 1 only: two groups per wave measured slower (fewer waves per SIMD)
        if nw not in MYERS_PARKED_NW:
            parts.append(gen_function("myers_rows_asm", f"{nw}, 1", ilp(R.myers_body(nw, 1, balanced=MYERS_BALANCED)), 2 * nw, nw))
    for nw in MYERS_PARKED_NW:  # 9 registers per word: HN parked in the VP register
        parts.append(gen_function("myers_rows_asm", f"{nw}, 1", ilp(R.myers_body(nw, 1, balanced=MYERS_BALANCED)), 2 * nw, nw))
    for nw in MYERS_SPLIT_NW:   # the chains in turns over blocks of MYERS_SPLIT words: 7 registers per word + 2 * MYERS_SPLIT
        parts.append(gen_function("myers_rows_asm", f"{nw}, 1", ilp(R.myers_body(nw, 1, split=MYERS_SPLIT, park=MYERS_PARK, balanced=MYERS_BALANCED)), 2 * nw, nw))
    parts.append("\n// Short subjects: the row is so short that the scalar dispatch bounds the loop, so a stream token\n"
                 "// carries two rows (bgsa_common.h: pair_stream_window).\n"
                 "// G = 2: two subject groups per wave — twice the vector work behind every dispatch, and these bodies are so\n"
                 "// small that the registers of two groups still leave eight waves per SIMD.\n"
                 "template <int NW, int G>\n"
                 "__device__ __forceinline__ int myers_pair_rows_asm(uint32_t (&state)[2 * G * NW],\n"
                 "                                                    const uint32_t (&P)[5][G * NW],\n"
                 "                                                    const unsigned long long stream, const int n_windows);\n")
    for nw in MYERS_PAIR_NW:
        for g in (1, 2):
            parts.append(gen_pair_function("myers_pair_rows_asm", f"{nw}, {g}", R.myers_body(nw, g), 2 * g * nw, g * nw))
    parts.append("\n// Semi-global (generator -m 0 -s): rows_ir.py: myers_semi_body — the subject right-aligned in its NW words,\n"
                 "// state = {VP, VN} x NW, then D[i][n] (running) and its minimum; 8 VALU per word + 3 per row.\n"
                 "template <int NW>\n"
                 "__device__ __forceinline__ int myers_semi_rows_asm(uint32_t (&state)[2 * NW + 2],\n"
                 "                                                   const uint32_t (&P)[5][NW],\n"
                 "                                                   const unsigned long long stream, const int n_windows);\n")
    for nw in MYERS_NW:
        parts.append(gen_function("myers_semi_rows_asm", f"{nw}", R.myers_semi_body(nw), 2 * nw + 2, nw))
    for nw in MYERS_SEMI_SPLIT_NW:   # 801..1024 bp: the chains in turns (as the global kernels of 30 / 32 words), scheduled
        parts.append(gen_function("myers_semi_rows_asm", f"{nw}", ilp(R.myers_semi_body(nw, split=MYERS_SPLIT)), 2 * nw + 2, nw))
    parts.append("\n// Long subjects (NW 26..32; the widths below 26 serve the column-block kernel and A/B runs): 3-bit character-code planes B[w*3+i] instead of five Peq planes.\n"
                 "template <int NW>\n"
                 "__device__ __forceinline__ int myers_planes_rows_asm(uint32_t (&state)[2 * NW],\n"
                 "                                                      const uint32_t (&B)[3 * NW],\n"
                 "                                                      const unsigned long long stream, const int n_windows);\n")
    for nw in MYERS_PLANES_NW:
        parts.append(gen_function("myers_planes_rows_asm", f"{nw}", ilp_planes(R.myers_planes_body(nw, MYERS_PLANES_SPLIT if nw >= 30 else 0, balanced=MYERS_BALANCED)),
                                  2 * nw, 0, n_planes=3 * nw))
    parts.append("\n// Semi-global on the code planes (subjects of 769..1024 bp): rows_ir.py: myers_semi_planes_body — the unused low\n"
                 "// columns carry code 7, which matches every class; 9 VALU per word + 3 per row.\n"
                 "template <int NW>\n"
                 "__device__ __forceinline__ int myers_semi_planes_rows_asm(uint32_t (&state)[2 * NW + 2],\n"
                 "                                                           const uint32_t (&B)[3 * NW],\n"
                 "                                                           const unsigned long long stream, const int n_windows);\n")
    for nw in MYERS_SEMI_PLANES_NW:
        parts.append(gen_function("myers_semi_planes_rows_asm", f"{nw}", R.myers_semi_planes_body(nw), 2 * nw + 2, 0,
                                  n_planes=3 * nw))
    parts.append("\n// One column block of a subject longer than 1024 bp (rows_ir.py:myers_block_body).  state =\n"
                 "// {VP, VN} x NW, carry-in words (add, HP, HN), carry-out words; voff = this lane's byte offset\n"
                 "// of the current 32-row chunk in the wave's carry buffer ([chunk][3][64] dwords at carry_base).\n"
                 "template <int NW>\n"
                 "__device__ __forceinline__ int myers_block_rows_asm(uint32_t (&state)[2 * NW + 6], const uint32_t (&B)[3 * NW],\n"
                 "                                                     uint32_t &voff, const unsigned long long carry_base,\n"
                 "                                                     const unsigned long long stream, const int n_windows);\n")
    for nw in MYERS_BLOCK_NW:
        parts.append(gen_blocked_function("myers_block_rows_asm", nw, R.myers_block_body(nw), 2 * nw, 3, 3 * nw, 0))
    parts.append("\n// The same with the five Peq planes of the block resident instead of the code planes: 8 VALU per word,\n"
                 "// narrower blocks (rows_ir.py: myers_peq_block_body).\n"
                 "template <int NW>\n"
                 "__device__ __forceinline__ int myers_peq_block_rows_asm(uint32_t (&state)[2 * NW + 6], const uint32_t (&P)[5][NW],\n"
                 "                                                         uint32_t &voff, const unsigned long long carry_base,\n"
                 "                                                         const unsigned long long stream, const int n_windows);\n")
    for nw in MYERS_PEQ_BLOCK_NW:
        parts.append(gen_blocked_function("myers_peq_block_rows_asm", nw, R.myers_peq_block_body(nw), 2 * nw, 3, 0, nw))
    parts.append("\n// Which of a block row's three carry-word pairs holds the HN bit that leaves the block's last column (pair 1 holds the HP bit):\n"
                 "// the eight-instruction row reads HN off the ADDITION's carries (pair 0), the ten-instruction row shifted it in a chain of its own (pair 2).\n"
                 f"constexpr int kMyersBlockHnPair = {0 if R.MYERS_EIGHT else 2};\n")
    (here / "myers_rows_gen.inc").write_text("".join(parts))
    # ---- BitPAl, default scores (other score sets: gen_bitpal_sets.py) ------------------------
    (here / "bitpal_rows_gen.inc").write_text(bitpal_inc_text(R.BITPAL_DEFAULT))
    # ---- banded -------------------------------------------------------------------------------
    (here / "banded_rows_gen.inc").write_text(head + gen_banded_function(False) + gen_banded_function(False, phase=True) + gen_banded_function(True) +
                                              gen_banded_chunk_function() + gen_banded_cut_function(1) + gen_banded_cut_function(2) +
                                              gen_banded_cut_function(2, "funnel32") + gen_banded_cut_function(2, "funnel64") +
                                              gen_banded_cut_function(1, "funnel64", sh64=True))
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
