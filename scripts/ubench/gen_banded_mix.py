#!/usr/bin/env python3
"""Emit banded_mix.hip: what a banded row costs on the vector pipe, by form and by how two subject groups share a wave.

Round 2 left the banded loop at 0.67 of the VALU issue peak with two explanations that each removed one cause and
neither won: the scalar side of the threaded loop, and the two half-rate instructions of the sliding
row.  This prices the vector side alone — no stream, no dispatch, no events; every register hard-coded inside ONE asm
block so that the loop is exactly the instructions listed — for:

  slide      the shipped 12-instruction row (v_alignbit window, D0 >> 1, error bit, error add)
  slide64    the same row with `v_lshrrev_b64 {d0:coll}, 1` doing D0 >> 1 AND collecting D0's bit 0 (10 instructions)
  phase      the band held in place (rows_ir.banded_phase_body without its low-bit register): 12 fast-class instructions
  *_seq2     two groups per wave, one group's row after the other's
  *_zip2     two groups per wave, instruction by instruction (the half-rate instructions of both groups adjacent)
and a few pure streams (fast only, half-rate only, one or two half-rate among ten fast, adjacent or apart).

Output: cycles per GROUP-row per SIMD at the nominal 2.4 GHz, for 8 / 4 / 2 / 1 waves per SIMD."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "bgsa_amd" / "csrc"))
from rows_ir import tt  # noqa: E402

T_X = tt(lambda w, m, vn: (w & m) | vn)
T_D0 = tt(lambda t, vp, x: (t ^ vp) | x)
T_HP = tt(lambda d, vp, vn: ~(d | vp) | vn)
T_VP = tt(lambda hp, x2, hn: ~(hp | x2) | hn)
T_E = tt(lambda d, one, _o: ~d & one)
T_PX = tt(lambda w, vn, m: (w | vn) & m)
T_S5 = tt(lambda a, d, m: (a & ~m) | (~d & m))

VMASK = "v10"
S_SH = "s20"


def regs(g):
    b = 20 + 20 * g
    names = "vp vn acc pad coll d0 e0 e1 w x t hp hn x2 e m3 s5".split()
    return {n: f"v{b + i}" for i, n in enumerate(names)} | {"pair": f"v[{b + 4}:{b + 5}]"}


def slide(g):
    r = regs(g)
    return [
        f"v_alignbit_b32 {r['w']}, {r['e1']}, {r['e0']}, {S_SH}",
        f"v_bitop3_b32 {r['x']}, {r['w']}, {VMASK}, {r['vn']} bitop3:0x{T_X:02x}",
        f"v_and_b32 {r['t']}, {r['x']}, {r['vp']}",
        f"v_add_u32 {r['t']}, {r['t']}, {r['vp']}",
        f"v_bitop3_b32 {r['d0']}, {r['t']}, {r['vp']}, {r['x']} bitop3:0x{T_D0:02x}",
        f"v_bitop3_b32 {r['hp']}, {r['d0']}, {r['vp']}, {r['vn']} bitop3:0x{T_HP:02x}",
        f"v_and_b32 {r['hn']}, {r['d0']}, {r['vp']}",
        f"v_lshrrev_b32 {r['x2']}, 1, {r['d0']}",
        f"v_and_b32 {r['vn']}, {r['x2']}, {r['hp']}",
        f"v_bitop3_b32 {r['vp']}, {r['hp']}, {r['x2']}, {r['hn']} bitop3:0x{T_VP:02x}",
        f"v_bitop3_b32 {r['e']}, {r['d0']}, 1, 1 bitop3:0x{T_E:02x}",
        f"v_add_u32 {r['acc']}, {r['acc']}, {r['e']}",
    ]


def slide64(g):
    r = regs(g)
    return [
        f"v_alignbit_b32 {r['w']}, {r['e1']}, {r['e0']}, {S_SH}",
        f"v_bitop3_b32 {r['x']}, {r['w']}, {VMASK}, {r['vn']} bitop3:0x{T_X:02x}",
        f"v_and_b32 {r['t']}, {r['x']}, {r['vp']}",
        f"v_add_u32 {r['t']}, {r['t']}, {r['vp']}",
        f"v_bitop3_b32 {r['d0']}, {r['t']}, {r['vp']}, {r['x']} bitop3:0x{T_D0:02x}",
        f"v_bitop3_b32 {r['hp']}, {r['d0']}, {r['vp']}, {r['vn']} bitop3:0x{T_HP:02x}",
        f"v_and_b32 {r['hn']}, {r['d0']}, {r['vp']}",
        f"v_lshrrev_b64 {r['pair']}, 1, {r['pair']}",
        f"v_and_b32 {r['vn']}, {r['d0']}, {r['hp']}",
        f"v_bitop3_b32 {r['vp']}, {r['hp']}, {r['d0']}, {r['hn']} bitop3:0x{T_VP:02x}",
    ]


def phase(g):
    r = regs(g)
    return [
        f"v_bitop3_b32 {r['x']}, {r['e0']}, {r['vn']}, {r['m3']} bitop3:0x{T_PX:02x}",
        f"v_and_b32 {r['t']}, {r['x']}, {r['vp']}",
        f"v_add_u32 {r['t']}, {r['t']}, {r['vp']}",
        f"v_bitop3_b32 {r['d0']}, {r['t']}, {r['vp']}, {r['x']} bitop3:0x{T_D0:02x}",
        f"v_bitop3_b32 {r['hp']}, {r['d0']}, {r['vp']}, {r['vn']} bitop3:0x{T_HP:02x}",
        f"v_and_b32 {r['hn']}, {r['d0']}, {r['vp']}",
        f"v_add_u32 {r['hp']}, {r['hp']}, {r['hp']}",
        f"v_add_u32 {r['hn']}, {r['hn']}, {r['hn']}",
        f"v_and_b32 {r['vn']}, {r['d0']}, {r['hp']}",
        f"v_bitop3_b32 {r['vp']}, {r['d0']}, {r['hp']}, {r['hn']} bitop3:0x{T_VP:02x}",
        f"v_bitop3_b32 {r['s5']}, {r['s5']}, {r['d0']}, {r['m3']} bitop3:0x{T_S5:02x}",
        f"v_add_u32 {r['m3']}, {r['m3']}, {r['m3']}",
    ]


def ablate(what):
    """The shipped sliding row with ONE instruction exchanged (which of them makes the row slow?)."""
    def fn(g):
        r = regs(g)
        b = slide(g)
        if "a0" in what: b[0] = f"v_mov_b32 {r['w']}, {r['e0']}"                          # no funnel shift at all
        if "shr" in what: b[0] = f"v_lshrrev_b32 {r['w']}, {S_SH}, {r['e0']}"             # one-word window, shift by an SGPR
        if "shi" in what: b[0] = f"v_lshrrev_b32 {r['w']}, 5, {r['e0']}"                  # ... by an immediate
        if "shv" in what: b[0] = f"v_lshrrev_b32 {r['w']}, v12, {r['e0']}"                # ... by a VGPR
        if "ai" in what: b[0] = f"v_alignbit_b32 {r['w']}, {r['e1']}, {r['e0']}, 5"       # funnel shift by an immediate
        if "av" in what: b[0] = f"v_alignbit_b32 {r['w']}, {r['e1']}, {r['e0']}, v12"     # ... by a VGPR
        if "l0" in what: b[7] = f"v_and_b32 {r['x2']}, {r['d0']}, {VMASK}"                # no D0 >> 1
        if "e0" in what: b[10] = f"v_not_b32 {r['e']}, {r['d0']}"                         # no two-constant v_bitop3
        if "e1" in what: b[10] = f"v_bitop3_b32 {r['e']}, {r['d0']}, v13, v13 bitop3:0x{T_E:02x}"   # the constant 1 from a VGPR
        return b
    return fn


def seq2(fn):
    return fn(0) + fn(1)


def zip2(fn):
    out = []
    for a, b in zip(fn(0), fn(1)):
        out += [a, b]
    return out


def pure(kind):
    """Independent instructions on sixteen registers v20..v35: no dependency between neighbours."""
    fast = lambda i: f"v_and_b32 v{20 + i % 16}, v{20 + i % 16}, {VMASK}"
    fast3 = lambda i: f"v_bitop3_b32 v{20 + i % 16}, v{20 + i % 16}, {VMASK}, v11 bitop3:0x{T_X:02x}"
    slow = lambda i: f"v_lshrrev_b32 v{20 + i % 16}, 1, v{20 + i % 16}"
    s64 = lambda i: f"v_lshrrev_b64 v[{20 + 2 * (i % 8)}:{21 + 2 * (i % 8)}], 1, v[{20 + 2 * (i % 8)}:{21 + 2 * (i % 8)}]"
    s64s = lambda i: f"v_lshrrev_b64 v[{20 + 2 * (i % 8)}:{21 + 2 * (i % 8)}], {S_SH}, v[{20 + 2 * (i % 8)}:{21 + 2 * (i % 8)}]"
    align = lambda i: f"v_alignbit_b32 v{20 + i % 16}, v{20 + (i + 1) % 16}, v{20 + i % 16}, {S_SH}"
    one = {
        "lshr_imm_inplace": lambda i: f"v_lshrrev_b32 v{20 + i % 16}, 1, v{20 + i % 16}",
        "lshr_imm": lambda i: f"v_lshrrev_b32 v{20 + i % 16}, 1, v{20 + (i + 5) % 16}",
        "lshl_imm": lambda i: f"v_lshlrev_b32 v{20 + i % 16}, 1, v{20 + i % 16}",
        "lshr_sgpr": lambda i: f"v_lshrrev_b32 v{20 + i % 16}, {S_SH}, v{20 + i % 16}",
        "lshr_vgpr": lambda i: f"v_lshrrev_b32 v{20 + i % 16}, v12, v{20 + i % 16}",
        "alignbit_imm": lambda i: f"v_alignbit_b32 v{20 + i % 16}, v{20 + (i + 1) % 16}, v{20 + i % 16}, 5",
        "alignbit_vgpr": lambda i: f"v_alignbit_b32 v{20 + i % 16}, v{20 + (i + 1) % 16}, v{20 + i % 16}, v12",
        "bitop3_c11": lambda i: f"v_bitop3_b32 v{20 + i % 16}, v{20 + i % 16}, 1, 1 bitop3:0x{T_E:02x}",
        "bitop3_vvs": lambda i: f"v_bitop3_b32 v{20 + i % 16}, v{20 + i % 16}, v11, {S_SH} bitop3:0x{T_X:02x}",
        "bfe": lambda i: f"v_bfe_u32 v{20 + i % 16}, v{20 + i % 16}, 5, 17",
        "bfe_sgpr": lambda i: f"v_bfe_u32 v{20 + i % 16}, v{20 + i % 16}, {S_SH}, 17",
        "perm": lambda i: f"v_perm_b32 v{20 + i % 16}, v{20 + (i + 1) % 16}, v{20 + i % 16}, v11",
        "alignbyte": lambda i: f"v_alignbyte_b32 v{20 + i % 16}, v{20 + (i + 1) % 16}, v{20 + i % 16}, 1",
        "bcnt": lambda i: f"v_bcnt_u32_b32 v{20 + i % 16}, v{20 + i % 16}, v11",
        "and_sgpr": lambda i: f"v_and_b32 v{20 + i % 16}, {S_SH}, v{20 + i % 16}",
        "lshl_add": lambda i: f"v_lshl_add_u32 v{20 + i % 16}, v{20 + i % 16}, 1, v11",
        "mul_u24": lambda i: f"v_mul_u32_u24 v{20 + i % 16}, v{20 + i % 16}, v11",
        "mov": lambda i: f"v_mov_b32 v{20 + i % 16}, v{20 + (i + 3) % 16}",
        "cndmask": lambda i: f"v_cndmask_b32 v{20 + i % 16}, v{20 + i % 16}, v11, vcc",
        "addc_vcc": lambda i: f"v_addc_co_u32 v{20 + i % 16}, vcc, v{20 + i % 16}, v11, vcc",
        "sub": lambda i: f"v_sub_u32 v{20 + i % 16}, v{20 + i % 16}, v11",
        "min_u32": lambda i: f"v_min_u32 v{20 + i % 16}, v{20 + i % 16}, v11",
        "pk_lshr16": lambda i: f"v_pk_lshrrev_b16 v{20 + i % 16}, 1, v{20 + i % 16}",
        "lshr16": lambda i: f"v_lshrrev_b16 v{20 + i % 16}, 1, v{20 + i % 16}",
    }
    if kind in one:
        return [one[kind](i) for i in range(12)]
    if kind.startswith("f10x2_"):    # two of the instruction among ten fast ones, apart
        f = one[kind[6:]]
        return [f(0)] + [(fast if i % 2 else fast3)(i) for i in range(1, 6)] + [f(6)] + [(fast if i % 2 else fast3)(i) for i in range(7, 12)]
    if kind == "fast12":
        return [(fast if i % 2 else fast3)(i) for i in range(12)]
    if kind == "slow12":
        return [slow(i) for i in range(12)]
    if kind == "lshr64_12":
        return [s64(i) for i in range(12)]
    if kind == "lshr64s_12":
        return [s64s(i) for i in range(12)]
    if kind == "align12":
        return [align(i) for i in range(12)]
    if kind == "f10_s2_apart":      # S F5 S F5
        return [slow(0)] + [(fast if i % 2 else fast3)(i) for i in range(1, 6)] + [slow(6)] + [(fast if i % 2 else fast3)(i) for i in range(7, 12)]
    if kind == "f10_s2_adjacent":   # S S F10
        return [slow(0), slow(1)] + [(fast if i % 2 else fast3)(i) for i in range(2, 12)]
    if kind == "f20_s4_adjacent":   # S S S S F20 (what zip2 of two sliding rows offers, twice)
        return [slow(i) for i in range(4)] + [(fast if i % 2 else fast3)(i) for i in range(4, 24)]
    if kind == "f11_s1":
        return [slow(0)] + [(fast if i % 2 else fast3)(i) for i in range(1, 12)]
    if kind == "f10_l64_2_apart":
        return [s64(0)] + [(fast if i % 2 else fast3)(i + 4) for i in range(1, 6)] + [s64(1)] + [(fast if i % 2 else fast3)(i + 4) for i in range(7, 12)]
    raise ValueError(kind)


UNROLL = 8
KERNELS = [
    # name, instruction list of ONE unrolled step, group-rows per step
    ("slide", slide(0), 1), ("slide_seq2", seq2(slide), 2), ("slide_zip2", zip2(slide), 2),
    ("slide64", slide64(0), 1), ("slide64_seq2", seq2(slide64), 2), ("slide64_zip2", zip2(slide64), 2),
    ("phase", phase(0), 1), ("phase_seq2", seq2(phase), 2), ("phase_zip2", zip2(phase), 2),
    ("slide_a0", ablate("a0")(0), 1), ("slide_l0", ablate("l0")(0), 1), ("slide_e0", ablate("e0")(0), 1), ("slide_e1", ablate("e1")(0), 1),
    ("slide_a0l0", ablate("a0 l0")(0), 1), ("slide_a0e0", ablate("a0 e0")(0), 1), ("slide_a0l0e0", ablate("a0 l0 e0")(0), 1),
    ("slide_ai", ablate("ai")(0), 1), ("slide_av", ablate("av")(0), 1),
    ("slide_shr", ablate("shr")(0), 1), ("slide_shi", ablate("shi")(0), 1), ("slide_shv", ablate("shv")(0), 1),
    ("slide_shr_e1", ablate("shr e1")(0), 1), ("slide_shr_seq2", seq2(ablate("shr")), 2), ("slide_shr_zip2", zip2(ablate("shr")), 2),
    ("slide_shr_e1_zip2", zip2(ablate("shr e1")), 2),
    ("fast12", pure("fast12"), 1), ("slow12", pure("slow12"), 1), ("lshr64_12", pure("lshr64_12"), 1),
    ("lshr64s_12", pure("lshr64s_12"), 1), ("align12", pure("align12"), 1),
    ("f11_s1", pure("f11_s1"), 1), ("f10_s2_apart", pure("f10_s2_apart"), 1), ("f10_s2_adjacent", pure("f10_s2_adjacent"), 1),
    ("f20_s4_adjacent", pure("f20_s4_adjacent"), 2), ("f10_l64_2_apart", pure("f10_l64_2_apart"), 1),
    ("lshr_imm_inplace", pure("lshr_imm_inplace"), 1), ("lshr_imm", pure("lshr_imm"), 1), ("lshl_imm", pure("lshl_imm"), 1), ("lshr_sgpr", pure("lshr_sgpr"), 1), ("lshr_vgpr", pure("lshr_vgpr"), 1), ("alignbit_imm", pure("alignbit_imm"), 1), ("alignbit_vgpr", pure("alignbit_vgpr"), 1), ("bitop3_c11", pure("bitop3_c11"), 1), ("bitop3_vvs", pure("bitop3_vvs"), 1), ("bfe", pure("bfe"), 1), ("bfe_sgpr", pure("bfe_sgpr"), 1), ("perm", pure("perm"), 1), ("alignbyte", pure("alignbyte"), 1), ("bcnt", pure("bcnt"), 1), ("and_sgpr", pure("and_sgpr"), 1), ("lshl_add", pure("lshl_add"), 1), ("mul_u24", pure("mul_u24"), 1), ("mov", pure("mov"), 1), ("cndmask", pure("cndmask"), 1), ("addc_vcc", pure("addc_vcc"), 1), ("sub", pure("sub"), 1), ("min_u32", pure("min_u32"), 1), ("pk_lshr16", pure("pk_lshr16"), 1), ("lshr16", pure("lshr16"), 1), ("f10x2_lshr_imm", pure("f10x2_lshr_imm"), 1), ("f10x2_lshr_sgpr", pure("f10x2_lshr_sgpr"), 1), ("f10x2_lshl_imm", pure("f10x2_lshl_imm"), 1), ("f10x2_alignbit_imm", pure("f10x2_alignbit_imm"), 1), ("f10x2_bitop3_c11", pure("f10x2_bitop3_c11"), 1), ("f10x2_bfe", pure("f10x2_bfe"), 1), ("f10x2_perm", pure("f10x2_perm"), 1), ("f10x2_bcnt", pure("f10x2_bcnt"), 1), ("f10x2_lshl_add", pure("f10x2_lshl_add"), 1),
]


def kernel(name, step):
    lines = ["v_mov_b32 v10, 0x1ffff", "v_mov_b32 v11, %[seed]", "v_mov_b32 v12, 5", "v_mov_b32 v13, 1", f"s_mov_b32 {S_SH}, 5"]
    lines += [f"v_xad_u32 v{i}, %[seed], {i}, %[seed]" for i in range(20, 60)]
    lines += ["s_mov_b32 s21, %[iters]", "L_loop_%=:"]
    lines += step * UNROLL
    lines += ["s_sub_u32 s21, s21, 1", "s_cmp_lg_u32 s21, 0", "s_cbranch_scc1 L_loop_%="]
    lines += ["v_mov_b32 %[res], v20"] + [f"v_xor_b32 %[res], %[res], v{i}" for i in range(21, 60)]
    text = "\n".join(f'        "{l}\\n"' if l.endswith(":") else f'        "{l}\\n\\t"' for l in lines)
    clob = ", ".join(f'"v{i}"' for i in [10, 11, 12, 13] + list(range(20, 60))) + ', "s20", "s21", "vcc", "scc"'
    return f"""
__global__ __launch_bounds__(256) void k_{name}(uint32_t *out, int iters)
{{
    uint32_t seed = threadIdx.x * 2654435761u + blockIdx.x, res;
    asm volatile(
{text}
        : [res] "=&v"(res)
        : [seed] "v"(seed), [iters] "s"(iters)
        : {clob});
    out[blockIdx.x * 256 + threadIdx.x] = res;
}}
"""


src = """// GENERATED by gen_banded_mix.py — see its docstring.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
"""
for n, step, _ in KERNELS:
    src += kernel(n, step)
src += f"""
struct Entry {{ const char *name; void (*fn)(uint32_t *, int); int rows; int insts; }};
int main(int argc, char **argv)
{{
    int iters = argc > 1 ? atoi(argv[1]) : 4000;
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    uint32_t *out; CHECK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
    Entry tests[] = {{{", ".join(f'{{"{n}", k_{n}, {r}, {len(step)}}}' for n, step, r in KERNELS)}}};
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("device %s, %d CUs; cycles at the nominal 2.4 GHz per GROUP-row per SIMD (and per instruction)\\n", prop.gcnArchName, cus);
    for (int wps : {{8, 4, 2}}) {{
        for (auto &t : tests) {{
            int blocks = cus * wps;
            hipLaunchKernelGGL(t.fn, dim3(blocks), dim3(256), 0, 0, out, 50);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(t.fn, dim3(blocks), dim3(256), 0, 0, out, iters);
            CHECK(hipEventRecord(e1));
            CHECK(hipDeviceSynchronize());
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            double steps_per_simd = (double)iters * {UNROLL} * wps;
            double cyc = ms * 1e6 * 2.4 / steps_per_simd;
            printf("%d waves/SIMD  %-16s %8.3f ms  %6.2f cycles/group-row  %5.2f cycles/instruction (%d instructions per %d group-rows)\\n",
                   wps, t.name, ms, cyc / t.rows, cyc / t.insts, t.insts, t.rows);
        }}
    }}
    return 0;
}}
"""
Path(__file__).with_name("banded_mix.hip").write_text(src)
