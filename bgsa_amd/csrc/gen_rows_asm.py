#!/usr/bin/env python3
"""Generate the gfx950 row-loop inline-asm bodies for the Myers kernels -> myers_rows_gen.inc.

BGSA's reference emits its kernels from a generator too (generator/source/.../MyersGenerator.java);
this is the gfx950 counterpart for the Myers recurrence.  The output is committed so the exact
instruction stream that is shipped can be read without running anything.

Why asm at all: the row body must exist five times (one per query character class, selected by a
scalar jump so `Eq = Peq[c][w]` costs nothing) and must update VP/VN in place.  hipcc turns the
five-way switch into v_mov copies around one shared body (+20 % VALU work) and lowers the
inter-word shifts to v_alignbit_b32, which issues at half rate on gfx950 (scripts/ubench/
valu_rate.hip: v_alignbit/v_lshlrev/v_lshl_or/v_bcnt = 4 cycles per wave64 instruction;
v_and/v_or/v_bitop3/v_add/v_addc = 2).  Here every instruction of the body is in the fast class:

    per (row, word), 10 VALU, all in the fast issue class:
      v_and        d   = P & E
      v_addc_co    d   = d + P + vcc              (phase A: the addition's carry chain)
      v_bitop3     d   = (d ^ P) | M
      v_or         d   = d | E                    (D0)
      v_bitop3     hp  = ~(d | P) | M
      v_and        hn  = d & P
      v_addc_co    hp  = hp + hp + vcc            (phase C: the 1-bit left shift across words IS
      v_addc_co    hn  = hn + hn + vcc             an add with carry; HP word 0 takes carry-in 1)
      v_and        M   = d & hp
      v_bitop3     P   = ~(d | hp) | hn

Control flow is threaded code: the query is a packed stream of one-byte codes (0..4 = A C G T N
row bodies, 5 = END, 6 = REFILL), 7 characters + 1 REFILL per 8-byte window, fetched with
s_load_dwordx2 one window ahead.  Each body ends with the dispatch of the next code
(s_and/s_lshr_b64/s_mul/s_add/s_addc/s_setpc), so a row costs one taken branch and no loop
counter.  gfx950 hazard "VALU writes VCC -> VALU reads it as carry-in" (2 wait states) is met by
construction: consecutive links of a carry chain always have two other instructions between them.
"""
from __future__ import annotations

import sys
from pathlib import Path

NW_LIST = [1, 2, 3, 4, 5, 6, 7, 8]
CODE_END, CODE_REFILL = 5, 6
NBODIES = 7

# Scalar scratch registers, hard-coded and declared as clobbers (inline asm cannot name the
# halves of a 64-bit "s" operand, and the jump needs lo/hi arithmetic).
S_WIN = "s[60:61]"; S_WIN_LO = "s60"
S_NXT = "s[62:63]"
S_BASE_LO, S_BASE_HI = "s64", "s65"
S_PC = "s[66:67]"; S_PC_LO, S_PC_HI = "s66", "s67"
S_C = "s68"
S_PTR = "s[70:71]"; S_PTR_LO, S_PTR_HI = "s70", "s71"
CLOBBERS = ["s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s70", "s71",
            "vcc", "scc", "memory"]


def dispatch(tag: str) -> list[str]:
    return [
        f"s_and_b32 {S_C}, {S_WIN_LO}, 7",
        f"s_lshr_b64 {S_WIN}, {S_WIN}, 8",
        f"s_mul_i32 {S_C}, {S_C}, (L_body1_%= - L_body0_%=)",
        f"s_add_u32 {S_PC_LO}, {S_BASE_LO}, {S_C}",
        f"s_addc_u32 {S_PC_HI}, {S_BASE_HI}, 0",
        f"s_setpc_b64 {S_PC}",
    ]


def body(nw: int, g_count: int, c: int) -> list[str]:
    """One DP row for character class c: in-place update of every word of every group.

    All three inter-word carry chains (the addition, the HP shift, the HN shift) go through VCC
    with the 4-byte VOP2 forms of v_add_co/v_addc_co, one chain at a time, so the body is three
    phases per group.  The VOP3 forms with an SGPR-pair carry, v_alignbit_b32 and v_lshlrev_b32
    all belong to the slow issue class on gfx950 and, worse, drag neighbouring fast instructions
    down to 4 cycles (scripts/ubench: 215 / 209 cycles per 50-instruction row against 156 here).
    Consecutive links of a chain are always >= 2 instructions apart (gfx950 hazard: VALU writes
    VCC -> VALU reads it as carry-in needs 2 wait states).
    """
    o: list[str] = []
    for g in range(g_count):
        P = lambda w: f"%[p{g}_{w}]"
        M = lambda w: f"%[m{g}_{w}]"
        E = lambda w: f"%[e{c}_{g}_{w}]"
        D = lambda w: f"%[d{w}]"
        HP = lambda w: f"%[hp{w}]"
        HN = lambda w: f"%[hn{w}]"
        # phase A: sum = (P & E) + P with carry, then D0 = ((sum ^ P) | M) | E
        for w in range(nw):
            o.append(f"v_and_b32 {D(w)}, {P(w)}, {E(w)}")
            o.append(f"v_add_co_u32 {D(w)}, vcc, {D(w)}, {P(w)}" if w == 0
                     else f"v_addc_co_u32 {D(w)}, vcc, {D(w)}, {P(w)}, vcc")
            o.append(f"v_bitop3_b32 {D(w)}, {D(w)}, {P(w)}, {M(w)} bitop3:0xbe")
            o.append(f"v_or_b32 {D(w)}, {D(w)}, {E(w)}")
        # phase B/C: HP = ~(D0 | P) | M, HN = D0 & P; HP <<= 1 across words, carry-in 1 (row edge)
        o.append("s_mov_b64 vcc, -1")
        for w in range(nw):
            o.append(f"v_bitop3_b32 {HP(w)}, {D(w)}, {P(w)}, {M(w)} bitop3:0xab")
            o.append(f"v_and_b32 {HN(w)}, {D(w)}, {P(w)}")
            o.append(f"v_addc_co_u32 {HP(w)}, vcc, {HP(w)}, {HP(w)}, vcc")
        # phase D: HN <<= 1 across words, then VN = D0 & HP, VP = ~(D0 | HP) | HN
        for w in range(nw):
            o.append(f"v_add_co_u32 {HN(w)}, vcc, {HN(w)}, {HN(w)}" if w == 0
                     else f"v_addc_co_u32 {HN(w)}, vcc, {HN(w)}, {HN(w)}, vcc")
            o.append(f"v_and_b32 {M(w)}, {D(w)}, {HP(w)}")
            o.append(f"v_bitop3_b32 {P(w)}, {D(w)}, {HP(w)}, {HN(w)} bitop3:0xab")
    return o


def gen_function(nw: int, g_count: int) -> str:
    asm: list[str] = []
    # ---- prologue -------------------------------------------------------------------------
    asm += [
        f"s_mov_b64 {S_PTR}, %[qp]",
        f"s_load_dwordx2 {S_WIN}, {S_PTR}, 0x0",
        f"s_load_dwordx2 {S_NXT}, {S_PTR}, 0x8",
        f"s_getpc_b64 {S_PC}",
        "L_anchor_%=:",
        f"s_add_u32 {S_BASE_LO}, {S_PC_LO}, (L_body0_%= - L_anchor_%=)",
        f"s_addc_u32 {S_BASE_HI}, {S_PC_HI}, 0",
        "s_waitcnt lgkmcnt(0)",  # scalar loads may return out of order: only 0 is safe
    ]
    asm += dispatch("pro")
    # ---- the five row bodies, identical length --------------------------------------------
    for c in range(5):
        asm.append(f"L_body{c}_%=:")
        asm += body(nw, g_count, c)
        asm += dispatch(f"b{c}")
    # ---- END (code 5) and REFILL (code 6) live in slots of the same stride ------------------
    asm.append("L_body5_%=:")
    asm.append("s_branch L_done_%=")
    asm.append(".fill ((L_body1_%= - L_body0_%=) - 4) / 4, 4, 0xbf800000")  # s_nop padding, never executed
    asm.append("L_body6_%=:")
    asm += [
        "s_waitcnt lgkmcnt(0)",
        f"s_mov_b64 {S_WIN}, {S_NXT}",
        f"s_add_u32 {S_PTR_LO}, {S_PTR_LO}, 8",
        f"s_addc_u32 {S_PTR_HI}, {S_PTR_HI}, 0",
        f"s_load_dwordx2 {S_NXT}, {S_PTR}, 0x8",
    ]
    asm += dispatch("refill")
    asm.append("L_done_%=:")
    asm.append("s_waitcnt lgkmcnt(0)")

    text = "\n".join(f'        "{line}\\n\\t"' if not line.endswith(":") else f'        "{line}\\n"' for line in asm)

    outs, ins = [], []
    for g in range(g_count):
        for w in range(nw):
            outs.append(f'[p{g}_{w}] "+v"(vp[{g * nw + w}])')
            outs.append(f'[m{g}_{w}] "+v"(vn[{g * nw + w}])')
    for w in range(nw):
        outs += [f'[d{w}] "=&v"(d[{w}])', f'[hp{w}] "=&v"(hp[{w}])', f'[hn{w}] "=&v"(hn[{w}])']
    for c in range(5):
        for g in range(g_count):
            for w in range(nw):
                ins.append(f'[e{c}_{g}_{w}] "v"(P[{c}][{g * nw + w}])')
    ins.append('[qp] "s"(stream)')
    clob = ", ".join(f'"{c}"' for c in CLOBBERS)
    n = nw * g_count
    return f"""
template <>
__device__ __forceinline__ void myers_rows_asm<{nw}, {g_count}>(uint32_t (&vp)[{n}], uint32_t (&vn)[{n}],
                                                       const uint32_t (&P)[5][{n}],
                                                       const unsigned long long stream)
{{
    uint32_t d[{nw}], hp[{nw}], hn[{nw}];
    asm volatile(
{text}
        : {", ".join(outs)}
        : {", ".join(ins)}
        : {clob});
}}
"""


def main() -> int:
    out = Path(sys.argv[1]) if len(sys.argv) > 1 else Path(__file__).with_name("myers_rows_gen.inc")
    parts = [
        "// GENERATED by gen_rows_asm.py — do not edit.  See that file for the design notes.\n",
        "// All rows of one query against G groups of one wave: in-place Myers update of vp/vn.\n",
        "// `stream` = device address (8-byte aligned, wave-uniform) of the packed query stream.\n",
        "template <int NW, int G>\n"
        "__device__ __forceinline__ void myers_rows_asm(uint32_t (&vp)[G * NW], uint32_t (&vn)[G * NW],\n"
        "                                               const uint32_t (&P)[5][G * NW],\n"
        "                                               const unsigned long long stream);\n",
    ]
    for nw in NW_LIST:
        for g in (1, 2):
            if nw * g <= 10:
                parts.append(gen_function(nw, g))
    out.write_text("".join(parts))
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
