// bank_conflict.hip — does the VGPR bank (index mod 4) of a VALU instruction's source operands change
// its issue cost on gfx950?  Every kernel runs the same instruction count with hard-coded registers
// v16..v47 (declared as clobbers), eight independent destinations so dependent-issue latency is not
// what is measured, and differs only in which registers the sources name.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)

#define CLOBBERS "v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31", \
                 "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47"

#define KERNEL(NAME, BODY)                                                                   \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, int iters)                    \
    {                                                                                        \
        uint32_t r;                                                                          \
        asm volatile("v_mov_b32 v16, %1\n\tv_mov_b32 v17, %1\n\tv_mov_b32 v18, %1\n\tv_mov_b32 v19, %1\n\t" \
                     "v_mov_b32 v20, %1\n\tv_mov_b32 v21, %1\n\tv_mov_b32 v22, %1\n\tv_mov_b32 v23, %1\n\t" \
                     "v_mov_b32 v24, %1\n\tv_mov_b32 v25, %1\n\tv_mov_b32 v26, %1\n\tv_mov_b32 v27, %1\n\t" \
                     "v_mov_b32 v28, %1\n\tv_mov_b32 v29, %1\n\tv_mov_b32 v30, %1\n\tv_mov_b32 v31, %1\n\t" \
                     "v_mov_b32 v32, %1\n\tv_mov_b32 v33, %1\n\tv_mov_b32 v34, %1\n\tv_mov_b32 v35, %1\n\t" \
                     "v_mov_b32 v36, %1\n\tv_mov_b32 v37, %1\n\tv_mov_b32 v38, %1\n\tv_mov_b32 v39, %1\n\t" \
                     "s_mov_b32 s20, %2\n\ts_mov_b32 s21, 0x55\n\t"                                                 \
                     "L_loop_%=:\n\t" REP16(BODY)                                            \
                     "s_sub_u32 s20, s20, 1\n\ts_cmp_lg_u32 s20, 0\n\ts_cbranch_scc1 L_loop_%=\n\t" \
                     "v_xor_b32 %0, v40, v41\n\tv_xor_b32 %0, %0, v42\n\tv_xor_b32 %0, %0, v43\n\t" \
                     "v_xor_b32 %0, %0, v44\n\tv_xor_b32 %0, %0, v45\n\tv_xor_b32 %0, %0, v46\n\tv_xor_b32 %0, %0, v47\n\t" \
                     : "=v"(r) : "v"(threadIdx.x), "s"(iters) : CLOBBERS, "s20", "s21", "scc", "vcc");  \
        out[blockIdx.x * 256 + threadIdx.x] = r;                                             \
    }

// destinations v40..v47; sources chosen per test.  bank(vN) = N mod 4 (hypothesis).
#define B3(d, a, b, c) "v_bitop3_b32 v" #d ", v" #a ", v" #b ", v" #c " bitop3:0x96\n\t"
#define A2(d, a, b) "v_and_b32 v" #d ", v" #a ", v" #b "\n\t"
#define A2E(d, a, b) "v_and_b32_e64 v" #d ", v" #a ", v" #b "\n\t"
#define B2K(d, a, b) "v_bitop3_b32 v" #d ", v" #a ", v" #b ", -1 bitop3:0x96\n\t"
#define B2S(d, a, b) "v_bitop3_b32 v" #d ", v" #a ", v" #b ", s21 bitop3:0x96\n\t"
#define B3RMW(d, a, b) "v_bitop3_b32 v" #d ", v" #d ", v" #a ", v" #b " bitop3:0x96\n\t"
#define AC(d, a, b) "v_addc_co_u32 v" #d ", vcc, v" #a ", v" #b ", vcc\n\t"

// three sources, three different banks (16:0 17:1 18:2 | 20:0 21:1 22:2 ...)
KERNEL(k3_distinct, B3(40,16,17,18) B3(41,20,21,22) B3(42,24,25,26) B3(43,28,29,30) B3(44,17,18,19) B3(45,21,22,23) B3(46,25,26,27) B3(47,29,30,31))
// three sources, two share a bank
KERNEL(k3_two_same, B3(40,16,20,17) B3(41,21,25,18) B3(42,24,28,19) B3(43,17,21,22) B3(44,18,22,23) B3(45,19,23,16) B3(46,25,29,26) B3(47,26,30,27))
// three sources, all in one bank
KERNEL(k3_all_same, B3(40,16,20,24) B3(41,17,21,25) B3(42,18,22,26) B3(43,19,23,27) B3(44,20,24,28) B3(45,21,25,29) B3(46,22,26,30) B3(47,23,27,31))
// three sources, one register named twice (+ one other bank)
KERNEL(k3_dup,      B3(40,16,17,17) B3(41,20,21,21) B3(42,24,25,25) B3(43,28,29,29) B3(44,17,18,18) B3(45,21,22,22) B3(46,25,26,26) B3(47,29,30,30))
// two sources, different banks / same bank
KERNEL(k2_distinct, A2(40,16,17) A2(41,20,21) A2(42,24,25) A2(43,28,29) A2(44,17,18) A2(45,21,22) A2(46,25,26) A2(47,29,30))
KERNEL(k2_same,     A2(40,16,20) A2(41,17,21) A2(42,18,22) A2(43,19,23) A2(44,24,28) A2(45,25,29) A2(46,26,30) A2(47,27,31))
// destination bank equal to a source bank (write/read conflict?)
KERNEL(k2_dst_same, A2(40,16,17) A2(41,17,18) A2(42,18,19) A2(43,19,16) A2(44,20,21) A2(45,21,22) A2(46,22,23) A2(47,23,20))
KERNEL(k3_dst_same, B3(40,16,17,18) B3(41,21,22,23) B3(42,26,27,24) B3(43,31,28,29) B3(44,20,21,22) B3(45,25,26,27) B3(46,30,31,28) B3(47,19,16,17))

// VOP3 encoding with two VGPR sources; three-source with an inline constant / an SGPR as the third
KERNEL(k2_e64,      A2E(40,16,17) A2E(41,20,21) A2E(42,24,25) A2E(43,28,29) A2E(44,17,18) A2E(45,21,22) A2E(46,25,26) A2E(47,29,30))
KERNEL(k3_const,    B2K(40,16,17) B2K(41,20,21) B2K(42,24,25) B2K(43,28,29) B2K(44,17,18) B2K(45,21,22) B2K(46,25,26) B2K(47,29,30))
KERNEL(k3_sgpr,     B2S(40,16,17) B2S(41,20,21) B2S(42,24,25) B2S(43,28,29) B2S(44,17,18) B2S(45,21,22) B2S(46,25,26) B2S(47,29,30))
// three VGPR sources where the destination is one of them (read-modify-write, as most body ops are)
KERNEL(k3_rmw,      B3RMW(40,16,17) B3RMW(41,20,21) B3RMW(42,24,25) B3RMW(43,28,29) B3RMW(44,17,18) B3RMW(45,21,22) B3RMW(46,25,26) B3RMW(47,29,30))
// alternating 3-source / 2-source
KERNEL(k_mix_3_2,   B3(40,16,17,18) A2(41,20,21) B3(42,24,25,26) A2(43,28,29) B3(44,17,18,19) A2(45,21,22) B3(46,25,26,27) A2(47,29,30))

struct Entry { const char *name; void (*fn)(uint32_t *, int); };

int main(int argc, char **argv)
{
    int waves_per_simd = argc > 1 ? atoi(argv[1]) : 4;
    int iters = argc > 2 ? atoi(argv[2]) : 4000;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    int blocks = cus * waves_per_simd;
    uint32_t *out;
    CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    Entry tests[] = {{"k3_distinct", k3_distinct}, {"k3_two_same", k3_two_same}, {"k3_all_same", k3_all_same}, {"k3_dup", k3_dup},
                     {"k2_distinct", k2_distinct}, {"k2_same", k2_same}, {"k2_dst_same", k2_dst_same}, {"k3_dst_same", k3_dst_same},
                     {"k2_e64", k2_e64}, {"k3_const", k3_const}, {"k3_sgpr", k3_sgpr}, {"k3_rmw", k3_rmw}, {"k_mix_3_2", k_mix_3_2}};
    printf("device %s, %d CUs, %d waves/SIMD, iters %d\n", prop.gcnArchName, cus, waves_per_simd, iters);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (auto &t : tests) {
        hipLaunchKernelGGL(t.fn, dim3(blocks), dim3(256), 0, 0, out, iters / 10);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(t.fn, dim3(blocks), dim3(256), 0, 0, out, iters);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        double insts = (double)iters * 16 * 8 * waves_per_simd;
        printf("%-14s %8.3f ms  %5.2f cycles/wave-instruction/SIMD @2.4GHz\n", t.name, ms, ms * 1e6 / insts * 2.4);
    }
    return 0;
}
