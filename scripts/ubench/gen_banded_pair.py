#!/usr/bin/env python3
"""Emit banded_pair.hip: what the 64-bit banded row (thresholds 16 .. 31, the reference's default 31 among them) costs on
the vector pipe, and what its alternatives would cost — before any of them is built into the kernel.

Round 3's finding (gen_banded_mix.py) was that ONE half-rate-class instruction makes a 12-instruction row issue at the slow
rate, so the 32-bit band's funnel shift left the row (one-word windows, k <= 12).  A 64-bit band has no such window: each
32-bit half of it needs 32 valid bits at an offset that moves by one bit per row, and a right shift of ONE register cannot
supply them.  This prices, with every register hard-coded inside one asm block (no stream, no dispatch):

  pair            the shipped 22-instruction row: two v_alignbit for the window, one for (D0 >> 1).lo
  pair_zip2       two subject groups per wave, interleaved instruction by instruction
  pair_x2fast     (D0 >> 1).lo without a funnel shift: lshr, and, sub, bitop3-insert (25 instructions, two v_alignbit left)
  pair_win1       the second row of a two-row token: window.hi = previous window.hi >> 1 (one v_alignbit fewer)
  pair_allfast    HYPOTHETICAL: the 22 instructions with every funnel shift replaced by a fast-class instruction of the same
                  operand count — not a correct row, the ceiling an all-fast 22-instruction row would have
  pair_2blk       HYPOTHETICAL count of the only all-fast form found: the band as two column blocks of < 32 bits with
                  one-word windows (k <= 24): 28 fast-class instructions
  pair_inplace    HYPOTHETICAL count of the band held in place on the pair (k <= 27): 24 fast-class instructions, plus a
                  re-anchor event per phase that this loop does not contain

Output: cycles per group-row per SIMD at the nominal 2.4 GHz, for 8 / 6 / 4 waves per SIMD."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "bgsa_amd" / "csrc"))
from rows_ir import tt  # noqa: E402

T_X = tt(lambda w, m, vn: (w & m) | vn)
T_D0 = tt(lambda t, vp, x: (t ^ vp) | x)
T_HP = tt(lambda d, vp, vn: ~(d | vp) | vn)
T_VP = tt(lambda hp, x2, hn: ~(hp | x2) | hn)
T_E = tt(lambda d, one, _o: ~d & one)
T_INS = tt(lambda a, b, c: (a & ~c) | (b & c))

ML, MH, TOP = "v10", "v11", "v14"
S_SH = "s20"
NAMES = "vpl vph vnl vnh acc e0 e1 e2 wl wh xl xh tl th dl dh hpl hph hnl hnh x2l x2h e n q1 q2 q3 q4".split()


def regs(g):
    b = 20 + 30 * g
    return {n: f"v{b + i}" for i, n in enumerate(NAMES)}


def pair(g, x2="align", win="align", add64=False, coll=False):
    r = regs(g)
    p2 = lambda lo: f"v[{r[lo][1:]}:{int(r[lo][1:]) + 1}]"   # the even-aligned pair that starts at register `lo`
    out = []
    if win == "align":
        out += [f"v_alignbit_b32 {r['wl']}, {r['e1']}, {r['e0']}, {S_SH}", f"v_alignbit_b32 {r['wh']}, {r['e2']}, {r['e1']}, {S_SH}"]
    elif win == "align_v":  # the shift amount in a VGPR (v12), kept by one more fast-class instruction per row
        out += [f"v_alignbit_b32 {r['wl']}, {r['e1']}, {r['e0']}, v12", f"v_alignbit_b32 {r['wh']}, {r['e2']}, {r['e1']}, v12",
                f"v_xor_b32 {r['q1']}, 1, {r['q1']}"]
    elif win == "align_i":  # HYPOTHETICAL: immediate shift amounts (would need one body per row position)
        out += [f"v_alignbit_b32 {r['wl']}, {r['e1']}, {r['e0']}, 5", f"v_alignbit_b32 {r['wh']}, {r['e2']}, {r['e1']}, 5"]
    elif win == "one":      # second row of a token: the high half is the previous one shifted (bit 63 is outside every band)
        out += [f"v_alignbit_b32 {r['wl']}, {r['wh']}, {r['wl']}, 1", f"v_lshrrev_b32 {r['wh']}, 1, {r['wh']}"]
    else:                   # hypothetical: fast-class stand-ins
        out += [f"v_lshrrev_b32 {r['wl']}, {S_SH}, {r['e0']}", f"v_lshrrev_b32 {r['wh']}, {S_SH}, {r['e1']}"]
    out += [
        f"v_bitop3_b32 {r['xl']}, {r['wl']}, {ML}, {r['vnl']} bitop3:0x{T_X:02x}",
        f"v_bitop3_b32 {r['xh']}, {r['wh']}, {MH}, {r['vnh']} bitop3:0x{T_X:02x}",
        f"v_and_b32 {r['tl']}, {r['xl']}, {r['vpl']}",
        f"v_and_b32 {r['th']}, {r['xh']}, {r['vph']}",
    ]
    if add64:               # gfx940+: one 64-bit (S0 << S1) + S2 in place of the carry pair (and no VCC hazard)
        out += [f"v_lshl_add_u64 {p2('tl')}, {p2('tl')}, 0, {p2('vpl')}"]
    else:
        out += [f"v_add_co_u32 {r['tl']}, vcc, {r['tl']}, {r['vpl']}",
                "s_nop 1",                                                       # as the emitter pads the shipped row (VCC hazard)
                f"v_addc_co_u32 {r['th']}, vcc, {r['th']}, {r['vph']}, vcc"]
    out += [
        f"v_bitop3_b32 {r['dl']}, {r['tl']}, {r['vpl']}, {r['xl']} bitop3:0x{T_D0:02x}",
        f"v_bitop3_b32 {r['dh']}, {r['th']}, {r['vph']}, {r['xh']} bitop3:0x{T_D0:02x}",
        f"v_bitop3_b32 {r['hpl']}, {r['dl']}, {r['vpl']}, {r['vnl']} bitop3:0x{T_HP:02x}",
        f"v_bitop3_b32 {r['hph']}, {r['dh']}, {r['vph']}, {r['vnh']} bitop3:0x{T_HP:02x}",
        f"v_and_b32 {r['hnl']}, {r['dl']}, {r['vpl']}",
        f"v_and_b32 {r['hnh']}, {r['dh']}, {r['vph']}",
    ]
    if x2 == "align":
        out += [f"v_alignbit_b32 {r['x2l']}, {r['dh']}, {r['dl']}, 1"]
    elif x2 == "fast":
        out += [f"v_lshrrev_b32 {r['x2l']}, 1, {r['dl']}", f"v_and_b32 {r['n']}, 1, {r['dh']}", f"v_sub_u32 {r['n']}, 0, {r['n']}",
                f"v_bitop3_b32 {r['x2l']}, {r['x2l']}, {r['n']}, {TOP} bitop3:0x{T_INS:02x}"]
    elif x2 == "b64":      # one 64-bit shift for both halves
        out += [f"v_lshrrev_b64 {p2('x2l')}, 1, {p2('dl')}"]
    else:
        out += [f"v_lshrrev_b32 {r['x2l']}, 1, {r['dl']}"]
    if x2 != "b64":
        out += [f"v_lshrrev_b32 {r['x2h']}, 1, {r['dh']}"]
    out += [
        f"v_and_b32 {r['vnl']}, {r['x2l']}, {r['hpl']}",
        f"v_and_b32 {r['vnh']}, {r['x2h']}, {r['hph']}",
        f"v_bitop3_b32 {r['vpl']}, {r['hpl']}, {r['x2l']}, {r['hnl']} bitop3:0x{T_VP:02x}",
        f"v_bitop3_b32 {r['vph']}, {r['hph']}, {r['x2h']}, {r['hnh']} bitop3:0x{T_VP:02x}",
    ]
    if coll:                # the lowest diagonal's D0 bits collected by a funnel shift (popcount once per 32 rows, in an event)
        out += [f"v_alignbit_b32 {r['acc']}, {r['dl']}, {r['acc']}, 1"]
    else:
        out += [f"v_bitop3_b32 {r['e']}, {r['dl']}, 1, 1 bitop3:0x{T_E:02x}",
                f"v_add_u32 {r['acc']}, {r['acc']}, {r['e']}"]
    return out


def extra_fast(g, n):
    """n more fast-class instructions on the group's spare registers (for the hypothetical instruction counts)."""
    r = regs(g)
    spare = [r["q1"], r["q2"], r["q3"], r["q4"]]
    return [f"v_xor_b32 {spare[i % 4]}, {spare[i % 4]}, {r['dl'] if i % 2 else r['dh']}" for i in range(n)]


def zip2(fn):
    a, b = fn(0), fn(1)
    out = []
    for x, y in zip(a, b):
        out += [x, y]
    return out


UNROLL = 8
KERNELS = [
    ("pair", pair(0), 1),
    ("pair_zip2", zip2(lambda g: pair(g)), 2),
    ("pair_vsh", pair(0, win="align_v"), 1),                          # window funnel shifts by a VGPR instead of an SGPR
    ("pair_vsh_zip2", zip2(lambda g: pair(g, win="align_v")), 2),
    ("pair_ish", pair(0, win="align_i"), 1),
    ("pair_x2fast", pair(0, x2="fast"), 1),
    ("pair_x2fast_zip2", zip2(lambda g: pair(g, x2="fast")), 2),
    ("pair_win1", pair(0, win="one"), 1),
    ("pair_tok2", pair(0) + pair(0, win="one"), 2),                     # a two-row token: full window, then the shifted one
    ("pair_tok2_zip2", zip2(lambda g: pair(g) + pair(g, win="one")), 4),
    ("pair_coll", pair(0, coll=True), 1),                               # 21: error bits collected by a funnel shift
    ("pair_sh64", pair(0, x2="b64"), 1),                                # 21: D0 >> 1 as one v_lshrrev_b64
    ("pair_add64", pair(0, add64=True), 1),                             # 21: the carry pair as one v_lshl_add_u64
    ("pair_19", pair(0, x2="b64", add64=True, coll=True), 1),           # 19: all three
    ("pair_19_zip2", zip2(lambda g: pair(g, x2="b64", add64=True, coll=True)), 2),
    ("pair_sh64_top", (lambda b: b[:2] + [b[15]] + b[16:20] + b[2:15] + b[20:])(pair(0, x2="b64")), 1),   # the ROTATED row: the windows'
    # funnel shifts and the previous row's D0 >> 1 adjacent at the top (one slow-class group per row), then the previous row's VN / VP, then this row
    ("pair_sh64_top_zip2", zip2(lambda g: (lambda b: b[:2] + [b[15]] + b[16:20] + b[2:15] + b[20:])(pair(g, x2="b64"))), 2),
    ("pair_sh64_zip2", zip2(lambda g: pair(g, x2="b64")), 2),
    ("pair_sh64_narrow", (lambda b, r: [b[0], f"v_lshrrev_b32 {r['wh']}, {S_SH}, {r['e1']}"] + b[2:])(pair(0, x2="b64"), regs(0)), 1),   # the
    # high window by a plain shift: what a row may do while the band's high word does not reach into the third match word (row mod 32 <= 63 - 2k)
    ("pair_x2only", pair(0, win="fake"), 1),                            # ONE funnel shift left in 22 instructions
    ("pair_allfast", pair(0, x2="fake", win="fake"), 1),
    ("pair_allfast_zip2", zip2(lambda g: pair(g, x2="fake", win="fake")), 2),
    ("pair_2blk", pair(0, x2="fast", win="fake") + extra_fast(0, 3), 1),            # 28 fast-class instructions
    ("pair_2blk_zip2", zip2(lambda g: pair(g, x2="fast", win="fake") + extra_fast(g, 3)), 2),
    ("pair_inplace", pair(0, x2="fake", win="fake")[2:] + extra_fast(0, 4), 1),     # 24: no window shift, masks and collector move
    ("pair_inplace_zip2", zip2(lambda g: pair(g, x2="fake", win="fake")[2:] + extra_fast(g, 4)), 2),
]


def kernel(name, step):
    lines = ["v_mov_b32 v10, 0xffffffff", "v_mov_b32 v11, 0x7fffffff", "v_mov_b32 v12, 5", "v_mov_b32 v13, 1", "v_mov_b32 v14, 0x80000000",
             f"s_mov_b32 {S_SH}, 5"]
    lines += [f"v_add_u32 v{i}, {i * 2654435 + 17}, %[seed]" for i in range(20, 80)]
    lines += ["s_mov_b32 s21, %[iters]", "L_loop_%=:"]
    lines += step * UNROLL
    lines += ["s_sub_u32 s21, s21, 1", "s_cmp_lg_u32 s21, 0", "s_cbranch_scc1 L_loop_%="]
    lines += ["v_mov_b32 %[res], v20"] + [f"v_xor_b32 %[res], %[res], v{i}" for i in range(21, 80)]
    text = "\n".join(f'        "{l}\\n"' if l.endswith(":") else f'        "{l}\\n\\t"' for l in lines)
    clob = ", ".join(f'"v{i}"' for i in [10, 11, 12, 13, 14] + list(range(20, 80))) + ', "s20", "s21", "vcc", "scc"'
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


src = """// GENERATED by gen_banded_pair.py — see its docstring.
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
    int iters = argc > 1 ? atoi(argv[1]) : 3000;
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    uint32_t *out; CHECK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
    Entry tests[] = {{{", ".join(f'{{"{n}", k_{n}, {r}, {sum(not x.startswith("s_nop") for x in step)}}}' for n, step, r in KERNELS)}}};
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("device %s, %d CUs; cycles at the nominal 2.4 GHz per GROUP-row per SIMD (and per instruction)\\n", prop.gcnArchName, cus);
    for (int wps : {{8, 6, 4}}) {{
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
            printf("%d waves/SIMD  %-20s %8.3f ms  %6.2f cycles/group-row  %5.2f cycles/instruction (%d instructions per %d group-rows)\\n",
                   wps, t.name, ms, cyc / t.rows, cyc / t.insts, t.insts, t.rows);
        }}
    }}
    return 0;
}}
"""
Path(__file__).with_name("banded_pair.hip").write_text(src)
