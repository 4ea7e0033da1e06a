// dep_rate.hip — what a wave pays for reading a VALU result right after it was written, on gfx950: wave64 instructions per
// SIMD cycle as a function of the distance between an instruction and its consumer (1 = the very next instruction) and
// of the number of waves a SIMD holds.  The long-subject Myers rows (two waves per SIMD) are what this prices: their
// phases are written word by word, each instruction consuming the previous one's result.
//   ./dep_rate [iters]      (every SIMD of the chip holds W waves, W = 1, 2, 3, 4, 8; LDS caps the workgroups per CU)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)

#define KERNEL(NAME, BODY)                                                                        \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, int iters)                         \
    {                                                                                             \
        extern __shared__ uint32_t lds[];                                                         \
        uint32_t a0 = threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 + 11,            \
                 a5 = a0 ^ 0x55, a6 = a0 + 99, a7 = ~a0, b = blockIdx.x, c = 0x9e3779b9u;          \
        if (iters < 0) lds[threadIdx.x] = b;                                                      \
        for (int i = 0; i < iters; i++) {                                                         \
            asm volatile(REP16(BODY)                                                              \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6),   \
                           "+v"(a7)                                                               \
                         : "v"(b), "v"(c)                                                         \
                         : "vcc");                                                                \
        }                                                                                         \
        out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;               \
    }

#define X(d, s) "v_xor_b32 %" #d ", %" #s ", %8\n\t"
#define B3(d, s) "v_bitop3_b32 %" #d ", %" #s ", %8, %9 bitop3:0xbe\n\t"
// distance 1: every instruction reads the previous one's result; 2, 3, 4, 8: that many chains in turn
KERNEL(xor_d1, X(0,0) X(0,0) X(0,0) X(0,0) X(0,0) X(0,0) X(0,0) X(0,0))
KERNEL(xor_d2, X(0,0) X(1,1) X(0,0) X(1,1) X(0,0) X(1,1) X(0,0) X(1,1))
KERNEL(xor_d3, X(0,0) X(1,1) X(2,2) X(0,0) X(1,1) X(2,2) X(0,0) X(1,1))   /* 8 per rep: the third chain is one short, same distances */
KERNEL(xor_d4, X(0,0) X(1,1) X(2,2) X(3,3) X(0,0) X(1,1) X(2,2) X(3,3))
KERNEL(xor_d8, X(0,0) X(1,1) X(2,2) X(3,3) X(4,4) X(5,5) X(6,6) X(7,7))
KERNEL(b3_d1, B3(0,0) B3(0,0) B3(0,0) B3(0,0) B3(0,0) B3(0,0) B3(0,0) B3(0,0))
KERNEL(b3_d2, B3(0,0) B3(1,1) B3(0,0) B3(1,1) B3(0,0) B3(1,1) B3(0,0) B3(1,1))
KERNEL(b3_d4, B3(0,0) B3(1,1) B3(2,2) B3(3,3) B3(0,0) B3(1,1) B3(2,2) B3(3,3))
KERNEL(b3_d8, B3(0,0) B3(1,1) B3(2,2) B3(3,3) B3(4,4) B3(5,5) B3(6,6) B3(7,7))
// the long-subject phase A, word by word (MATCH3 e; AND d,P,e; ADDC d,d,P; BITOP3 d; OR d,d,e — each reads the one before) ...
#define WORD(e, d, p) "v_bitop3_b32 %" #e ", %8, %9, %" #p " bitop3:0x96\n\tv_and_b32 %" #d ", %" #p ", %" #e "\n\tv_addc_co_u32 %" #d ", vcc, %" #d ", %" #p ", vcc\n\t" \
                      "v_bitop3_b32 %" #d ", %" #d ", %" #p ", %9 bitop3:0xbe\n\tv_or_b32 %" #d ", %" #d ", %" #e "\n\t"
KERNEL(phaseA_serial, WORD(0, 1, 2) WORD(0, 3, 4) WORD(0, 5, 6) WORD(0, 1, 2) WORD(0, 3, 4) WORD(0, 5, 6) WORD(0, 1, 2) WORD(0, 3, 4))   /* 40 per rep */
// ... and two words at a time, instruction by instruction in turn (second mask register), the two chain links two apart
#define PAIR(e0, d0, p0, e1, d1, p1) \
    "v_bitop3_b32 %" #e0 ", %8, %9, %" #p0 " bitop3:0x96\n\tv_bitop3_b32 %" #e1 ", %8, %9, %" #p1 " bitop3:0x96\n\t" \
    "v_and_b32 %" #d0 ", %" #p0 ", %" #e0 "\n\tv_and_b32 %" #d1 ", %" #p1 ", %" #e1 "\n\t" \
    "v_addc_co_u32 %" #d0 ", vcc, %" #d0 ", %" #p0 ", vcc\n\tv_or_b32 %" #e0 ", %" #e0 ", %8\n\tv_or_b32 %" #e1 ", %" #e1 ", %8\n\t" \
    "v_addc_co_u32 %" #d1 ", vcc, %" #d1 ", %" #p1 ", vcc\n\t" \
    "v_bitop3_b32 %" #d0 ", %" #d0 ", %" #p0 ", %9 bitop3:0xbe\n\tv_bitop3_b32 %" #d1 ", %" #d1 ", %" #p1 ", %9 bitop3:0xbe\n\t"
KERNEL(phaseA_paired, PAIR(0, 1, 2, 7, 3, 4) PAIR(0, 5, 6, 7, 1, 2) PAIR(0, 3, 4, 7, 5, 6) PAIR(0, 1, 2, 7, 3, 4))   /* 40 per rep, same instruction kinds (two of the ORs stand in for the final ORs) */

struct Entry { const char *name; void (*fn)(uint32_t *, int); int per_rep; };

int main(int argc, char **argv)
{
    int iters = argc > 1 ? atoi(argv[1]) : 2000;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    uint32_t *out;
    CHECK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
    Entry tests[] = {
#define E(n, per) {#n, n, per}
        E(xor_d1, 8), E(xor_d2, 8), E(xor_d3, 8), E(xor_d4, 8), E(xor_d8, 8), E(b3_d1, 8), E(b3_d2, 8), E(b3_d4, 8), E(b3_d8, 8),
        E(phaseA_serial, 40), E(phaseA_paired, 40),
    };
    printf("device %s, %d CUs, nominal clock %d MHz; cycles per wave64 instruction per SIMD at the nominal clock\n", prop.gcnArchName, cus, prop.clockRate / 1000);
    printf("%-16s", "waves per SIMD:");
    const int ws[] = {1, 2, 3, 4, 8};
    for (int w : ws) printf(" %7d", w);
    printf("\n");
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (auto &t : tests) {
        printf("%-16s", t.name);
        for (int w : ws) {
            const int blocks = cus * w;                       // 256 threads = one wave per SIMD of a CU
            const size_t lds = (size_t)(160 * 1024 / w) - 2048;   // at most w workgroups fit a CU
            CHECK(hipFuncSetAttribute((const void *)t.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(t.fn, dim3(blocks), dim3(256), lds, 0, out, iters / 10);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(t.fn, dim3(blocks), dim3(256), lds, 0, out, iters);
            CHECK(hipEventRecord(e1));
            CHECK(hipDeviceSynchronize());
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double insts_per_simd = (double)iters * 16 * t.per_rep * w;
            printf(" %7.2f", ms * 1e6 / insts_per_simd * (prop.clockRate / 1e6));
        }
        printf("\n");
    }
    return 0;
}
