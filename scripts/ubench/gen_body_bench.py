#!/usr/bin/env python3
"""Emit body_rate.hip: microbenchmarks of candidate Myers row bodies (cycles per row per SIMD)."""
NW = 5
def word(w, variant, T="%[t]", HP="%[hp]", HN="%[hn]", sfx=""):
    P, M, E = f"%[p{w}{sfx}]", f"%[m{w}{sfx}]", f"%[e{w}{sfx}]"
    CP, CN, ONES = ("s[72:73]", "s[74:75]", "s[76:77]") if not sfx else ("s[78:79]", "s[80:81]", "s[76:77]")
    VC = "vcc" if not sfx else "s[82:83]"
    o = [f"v_and_b32 {T}, {P}, {E}"]
    if variant == "nocarry":
        o.append(f"v_add_u32 {T}, {T}, {P}")
    elif VC == "vcc":
        o.append(f"v_add_co_u32 {T}, vcc, {T}, {P}" if w == 0 else f"v_addc_co_u32 {T}, vcc, {T}, {P}, vcc")
    else:
        o.append(f"v_add_co_u32 {T}, {VC}, {T}, {P}" if w == 0 else f"v_addc_co_u32 {T}, {VC}, {T}, {P}, {VC}")
    o += [f"v_bitop3_b32 {T}, {T}, {P}, {M} bitop3:0xbe", f"v_or_b32 {T}, {T}, {E}",
          f"v_bitop3_b32 {HP}, {T}, {P}, {M} bitop3:0xab", f"v_and_b32 {HN}, {T}, {P}"]
    if variant == "nocarry":
        o += [f"v_add_u32 {HP}, {HP}, {HP}", f"v_add_u32 {HN}, {HN}, {HN}"]
    elif variant == "alignbit":
        o += [f"v_alignbit_b32 {HP}, {HP}, {HN}, 31", f"v_alignbit_b32 {HN}, {HN}, {HP}, 31"]
    elif variant == "vccshift":   # all three chains through vcc-like VOP2? (only one vcc) -> use e64 sgpr
        o += [f"v_addc_co_u32 {HP}, {CP}, {HP}, {HP}, {CP if w else ONES}",
              (f"v_addc_co_u32 {HN}, {CN}, {HN}, {HN}, {CN}" if w else f"v_add_co_u32 {HN}, {CN}, {HN}, {HN}")]
    else:
        o += [f"v_addc_co_u32 {HP}, {CP}, {HP}, {HP}, {CP if w else ONES}",
              (f"v_addc_co_u32 {HN}, {CN}, {HN}, {HN}, {CN}" if w else f"v_add_co_u32 {HN}, {CN}, {HN}, {HN}")]
    o += [f"v_bitop3_b32 {P}, {T}, {HP}, {HN} bitop3:0xab", f"v_and_b32 {M}, {T}, {HP}"]
    return o

def body_vccphase():
    o = []
    P = lambda w: f"%[p{w}]"; M = lambda w: f"%[m{w}]"; E = lambda w: f"%[e{w}]"
    D = lambda w: f"%[d{w}]"; HP = lambda w: f"%[hp{w}]"; HN = lambda w: f"%[hn{w}]"
    # phase A: sum chain through vcc
    for w in range(NW):
        o.append(f"v_and_b32 {D(w)}, {P(w)}, {E(w)}")
        o.append(f"v_add_co_u32 {D(w)}, vcc, {D(w)}, {P(w)}" if w == 0 else f"v_addc_co_u32 {D(w)}, vcc, {D(w)}, {P(w)}, vcc")
        o.append(f"v_bitop3_b32 {D(w)}, {D(w)}, {P(w)}, {M(w)} bitop3:0xbe")
        o.append(f"v_or_b32 {D(w)}, {D(w)}, {E(w)}")
    # phase B+C: hp/hn, HP shift chain
    o.append("s_mov_b64 vcc, -1")
    for w in range(NW):
        o.append(f"v_bitop3_b32 {HP(w)}, {D(w)}, {P(w)}, {M(w)} bitop3:0xab")
        o.append(f"v_and_b32 {HN(w)}, {D(w)}, {P(w)}")
        if w == 0:
            o.append("s_nop 0")
        o.append(f"v_addc_co_u32 {HP(w)}, vcc, {HP(w)}, {HP(w)}, vcc")
    # phase D: HN shift chain + final
    for w in range(NW):
        o.append(f"v_add_co_u32 {HN(w)}, vcc, {HN(w)}, {HN(w)}" if w == 0 else f"v_addc_co_u32 {HN(w)}, vcc, {HN(w)}, {HN(w)}, vcc")
        o.append(f"v_and_b32 {M(w)}, {D(w)}, {HP(w)}")
        o.append(f"v_bitop3_b32 {P(w)}, {D(w)}, {HP(w)}, {HN(w)} bitop3:0xab")
    return o

def body(variant):
    if variant == "vccphase":
        return body_vccphase()
    if variant == "interleave2":
        a = [word(w, "carry") for w in range(NW)]
        b = [word(w, "carry", "%[t2]", "%[hp2]", "%[hn2]", "b") for w in range(NW)]
        out = []
        for wa, wb in zip(a, b):
            for x, y in zip(wa, wb):
                out += [x, y]
        return out
    out = []
    for w in range(NW):
        out += word(w, variant)
    return out

def kernel(name, variant, two=False):
    lines = body(variant)
    text = "\n".join(f'            "{l}\\n\\t"' for l in lines)
    outs = []
    for w in range(NW):
        outs += [f'[p{w}] "+v"(p[{w}])', f'[m{w}] "+v"(m[{w}])']
    if two:
        for w in range(NW):
            outs += [f'[p{w}b] "+v"(p2[{w}])', f'[m{w}b] "+v"(m2[{w}])']
        outs += ['[t2] "=&v"(t2)', '[hp2] "=&v"(hp2)', '[hn2] "=&v"(hn2)']
    outs += ['[t] "=&v"(t)', '[hp] "=&v"(hp)', '[hn] "=&v"(hn)']
    if variant == "vccphase":
        for w in range(NW):
            outs += [f'[d{w}] "=&v"(dd[{w}])', f'[hp{w}] "=&v"(hh[{w}])', f'[hn{w}] "=&v"(nn[{w}])']
    ins = [f'[e{w}] "v"(e[{w}])' for w in range(NW)]
    if two:
        ins += [f'[e{w}b] "v"(e[{w}])' for w in range(NW)]
    return f"""
__global__ __launch_bounds__(256) void {name}(uint32_t *out, int iters)
{{
    uint32_t p[{NW}], m[{NW}], e[{NW}], p2[{NW}], m2[{NW}], dd[{NW}], hh[{NW}], nn[{NW}], t, hp, hn, t2, hp2, hn2;
    for (int w = 0; w < {NW}; w++) {{ p[w] = ~0u; m[w] = 0; e[w] = threadIdx.x * 2654435761u + w * 40503u + blockIdx.x; p2[w] = ~0u; m2[w] = 0; }}
    asm volatile("s_mov_b64 s[76:77], -1" ::: "s76", "s77");
    for (int i = 0; i < iters; i++) {{
        asm volatile(
{text}
            : {", ".join(outs)}
            : {", ".join(ins)}
            : "vcc", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83");
    }}
    uint32_t acc = 0;
    for (int w = 0; w < {NW}; w++) acc ^= p[w] ^ m[w] ^ p2[w] ^ m2[w];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}}
"""

src = """// GENERATED by gen_body_bench.py
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
"""
variants = [("b_carry", "carry", False, 1), ("b_nocarry", "nocarry", False, 1), ("b_alignbit", "alignbit", False, 1), ("b_interleave2", "interleave2", True, 2), ("b_vccphase", "vccphase", False, 1)]
for n, v, two, _ in variants:
    src += kernel(n, v, two)
src += """
struct Entry { const char *name; void (*fn)(uint32_t *, int); int rows; };
int main(int argc, char **argv)
{
    int iters = argc > 1 ? atoi(argv[1]) : 20000;
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    uint32_t *out; CHECK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
    Entry tests[] = {""" + ", ".join(f'{{"{n}", {n}, {r}}}' for n, _, _, r in variants) + """};
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int wps : {8, 4, 2, 1}) {
        for (auto &t : tests) {
            int blocks = cus * wps;
            hipLaunchKernelGGL(t.fn, dim3(blocks), dim3(256), 0, 0, out, 100);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(t.fn, dim3(blocks), dim3(256), 0, 0, out, iters);
            CHECK(hipEventRecord(e1));
            CHECK(hipDeviceSynchronize());
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            double rows_per_simd = (double)iters * t.rows * wps;
            printf("%d waves/SIMD  %-14s %8.3f ms  %7.1f cycles/row/SIMD @2.4GHz (50 VALU/row)\\n", wps, t.name, ms, ms * 1e6 * 2.4 / rows_per_simd);
        }
    }
    return 0;
}
"""
open("body_rate.hip", "w").write(src)
