// chain_rate.hip — what a carry chain through VCC costs on gfx950 as a function of the distance between its links, and what
// parking the chain costs (round 5).  A link is `v_addc_co_u32 d, vcc, d, d, vcc` (reads and writes VCC; the ISA asks for two
// wait states between a VALU write of VCC and a VALU read of it as carry-in); between two links sit N independent
// instructions on other registers (alternating v_xor / v_bitop3, no dependency on the chain or on each other within the gap).
//   link3 .. link8 : 3 .. 8 instructions from link to link (2 .. 7 fillers)
//   park_s8 / park_v8: the chain of link4 parked and resumed every 8 links — through a scalar pair (s_mov_b64 sX, vcc ;
//                      s_mov_b64 vcc, sY) or through a vector register (v_subb_co_u32 c, vcc, c, c, vcc ; v_add_co_u32 c, vcc, c, c)
//   ./chain_rate [iters]     cycles per wave64 VALU instruction per SIMD at the nominal clock, for 1, 2, 4, 8 waves per SIMD
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
#define REP4(x) x x x x
#define REP8(x) REP4(x) REP4(x)

#define KERNEL(NAME, BODY)                                                                        \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, int iters)                         \
    {                                                                                             \
        extern __shared__ uint32_t lds[];                                                         \
        uint32_t a0 = threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 + 11,            \
                 a5 = a0 ^ 0x55, a6 = a0 + 99, a7 = ~a0, b = blockIdx.x, c = 0x9e3779b9u;          \
        if (iters < 0) lds[threadIdx.x] = b;                                                      \
        for (int i = 0; i < iters; i++) {                                                         \
            asm volatile(BODY                                                                     \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6),   \
                           "+v"(a7)                                                               \
                         : "v"(b), "v"(c)                                                         \
                         : "vcc", "s40", "s41", "s42", "s43");                                    \
        }                                                                                         \
        out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;               \
    }

#define L "v_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
#define X(d) "v_xor_b32 %" #d ", %" #d ", %8\n\t"
#define B(d) "v_bitop3_b32 %" #d ", %" #d ", %8, %9 bitop3:0xbe\n\t"
// fillers: registers 1..6 in turn, so that no filler reads the result of one closer than six instructions back
#define F2 X(1) B(2)
#define F3 X(1) B(2) X(3)
#define F4 X(1) B(2) X(3) B(4)
#define F5 X(1) B(2) X(3) B(4) X(5)
#define F7 X(1) B(2) X(3) B(4) X(5) B(6) X(1)
KERNEL(link3, REP8(REP8(L F2)))     /* 64 links, 192 instructions */
KERNEL(link4, REP8(REP8(L F3)))     /* 256 */
KERNEL(link5, REP8(REP8(L F4)))     /* 320 */
KERNEL(link6, REP8(REP8(L F5)))     /* 384 */
KERNEL(link8, REP8(REP8(L F7)))     /* 512 */
// eight links of link4, then the chain is parked and the one parked before is taken up: scalar pair ...
#define PARK_S "s_mov_b64 s[40:41], vcc\n\ts_mov_b64 vcc, s[42:43]\n\t" X(1) B(2) X(3) REP8(L F3) "s_mov_b64 s[42:43], vcc\n\ts_mov_b64 vcc, s[40:41]\n\t" X(1) B(2) X(3)
KERNEL(park_s8, REP4(REP8(L F3) PARK_S))     /* per rep: 16 links x 4 + 6 = 70 VALU; x 4 = 280 VALU + 16 scalar moves */
// ... vector register (a7 parks one chain, a6 the other; the fillers keep away from them)
#define SV(r) "v_subb_co_u32 %" #r ", vcc, %" #r ", %" #r ", vcc\n\t"
#define LV(r) "v_add_co_u32 %" #r ", vcc, %" #r ", %" #r "\n\t"
#define PARK_V SV(7) LV(6) X(1) B(2) X(3) REP8(L F3) SV(6) LV(7) X(1) B(2) X(3)
KERNEL(park_v8, REP4(REP8(L F3) PARK_V))     /* 70 + 4 = 74 VALU per rep; x 4 = 296 */
// no chain at all: the filler mix alone
KERNEL(fill_only, REP8(REP8(X(1) B(2) X(3) B(4))))   /* 256 */

struct Entry { const char *name; void (*fn)(uint32_t *, int); int per_iter; };

int main(int argc, char **argv)
{
    int iters = argc > 1 ? atoi(argv[1]) : 4000;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    uint32_t *out;
    CHECK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
    Entry tests[] = {
#define E(n, per) {#n, n, per}
        E(fill_only, 256), E(link3, 192), E(link4, 256), E(link5, 320), E(link6, 384), E(link8, 512), E(park_s8, 280), E(park_v8, 296),
    };
    printf("device %s, %d CUs, nominal clock %d MHz; cycles per wave64 VALU instruction per SIMD at the nominal clock\n", prop.gcnArchName, cus, prop.clockRate / 1000);
    printf("%-12s", "waves/SIMD:");
    const int ws[] = {1, 2, 4, 8};
    for (int w : ws) printf(" %7d", w);
    printf("\n");
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (auto &t : tests) {
        printf("%-12s", t.name);
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
            const double insts_per_simd = (double)iters * t.per_iter * w;
            printf(" %7.2f", ms * 1e6 / insts_per_simd * (prop.clockRate / 1e6));
        }
        printf("\n");
    }
    return 0;
}
