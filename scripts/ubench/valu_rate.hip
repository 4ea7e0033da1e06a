// valu_rate.hip — per-instruction VALU issue cost on gfx950 (cycles per wave64 instruction per
// SIMD), measured with every SIMD of the chip holding `waves` waves that each run a long chain of
// one instruction kind.  Used to price the Myers/BitPAl row bodies.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

// Eight independent accumulators a0..a7 so that dependent-issue latency is not what is measured.
#define KERNEL(NAME, BODY)                                                                        \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, int iters)                         \
    {                                                                                             \
        uint32_t a0 = threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 + 11,            \
                 a5 = a0 ^ 0x55, a6 = a0 + 99, a7 = ~a0, b = blockIdx.x, c = 0x9e3779b9u;          \
        for (int i = 0; i < iters; i++) {                                                         \
            asm volatile(REP16(BODY)                                                              \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6),   \
                           "+v"(a7)                                                               \
                         : "v"(b), "v"(c)                                                         \
                         : "vcc", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81");                                                                \
        }                                                                                         \
        out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;               \
    }

// 2-source
KERNEL(k_and_self,  "v_and_b32 %0, %0, %8\n\tv_and_b32 %1, %1, %8\n\tv_and_b32 %2, %2, %8\n\tv_and_b32 %3, %3, %8\n\tv_and_b32 %4, %4, %8\n\tv_and_b32 %5, %5, %8\n\tv_and_b32 %6, %6, %8\n\tv_and_b32 %7, %7, %8\n\t")
KERNEL(k_xor,       "v_xor_b32 %0, %0, %8\n\tv_xor_b32 %1, %1, %8\n\tv_xor_b32 %2, %2, %8\n\tv_xor_b32 %3, %3, %8\n\tv_xor_b32 %4, %4, %8\n\tv_xor_b32 %5, %5, %8\n\tv_xor_b32 %6, %6, %8\n\tv_xor_b32 %7, %7, %8\n\t")
KERNEL(k_add,       "v_add_u32 %0, %0, %8\n\tv_add_u32 %1, %1, %8\n\tv_add_u32 %2, %2, %8\n\tv_add_u32 %3, %3, %8\n\tv_add_u32 %4, %4, %8\n\tv_add_u32 %5, %5, %8\n\tv_add_u32 %6, %6, %8\n\tv_add_u32 %7, %7, %8\n\t")
KERNEL(k_lshl,      "v_lshlrev_b32 %0, 1, %0\n\tv_lshlrev_b32 %1, 1, %1\n\tv_lshlrev_b32 %2, 1, %2\n\tv_lshlrev_b32 %3, 1, %3\n\tv_lshlrev_b32 %4, 1, %4\n\tv_lshlrev_b32 %5, 1, %5\n\tv_lshlrev_b32 %6, 1, %6\n\tv_lshlrev_b32 %7, 1, %7\n\t")
KERNEL(k_not,       "v_not_b32 %0, %0\n\tv_not_b32 %1, %1\n\tv_not_b32 %2, %2\n\tv_not_b32 %3, %3\n\tv_not_b32 %4, %4\n\tv_not_b32 %5, %5\n\tv_not_b32 %6, %6\n\tv_not_b32 %7, %7\n\t")
KERNEL(k_mov,       "v_mov_b32 %0, %1\n\tv_mov_b32 %1, %2\n\tv_mov_b32 %2, %3\n\tv_mov_b32 %3, %4\n\tv_mov_b32 %4, %5\n\tv_mov_b32 %5, %6\n\tv_mov_b32 %6, %7\n\tv_mov_b32 %7, %8\n\t")
KERNEL(k_bcnt,      "v_bcnt_u32_b32 %0, %0, %8\n\tv_bcnt_u32_b32 %1, %1, %8\n\tv_bcnt_u32_b32 %2, %2, %8\n\tv_bcnt_u32_b32 %3, %3, %8\n\tv_bcnt_u32_b32 %4, %4, %8\n\tv_bcnt_u32_b32 %5, %5, %8\n\tv_bcnt_u32_b32 %6, %6, %8\n\tv_bcnt_u32_b32 %7, %7, %8\n\t")
// carry ops
KERNEL(k_add_co,    "v_add_co_u32 %0, vcc, %0, %8\n\tv_add_co_u32 %1, vcc, %1, %8\n\tv_add_co_u32 %2, vcc, %2, %8\n\tv_add_co_u32 %3, vcc, %3, %8\n\tv_add_co_u32 %4, vcc, %4, %8\n\tv_add_co_u32 %5, vcc, %5, %8\n\tv_add_co_u32 %6, vcc, %6, %8\n\tv_add_co_u32 %7, vcc, %7, %8\n\t")
KERNEL(k_addc_chain,"v_addc_co_u32 %0, vcc, %0, %8, vcc\n\tv_addc_co_u32 %1, vcc, %1, %8, vcc\n\tv_addc_co_u32 %2, vcc, %2, %8, vcc\n\tv_addc_co_u32 %3, vcc, %3, %8, vcc\n\tv_addc_co_u32 %4, vcc, %4, %8, vcc\n\tv_addc_co_u32 %5, vcc, %5, %8, vcc\n\tv_addc_co_u32 %6, vcc, %6, %8, vcc\n\tv_addc_co_u32 %7, vcc, %7, %8, vcc\n\t")
KERNEL(k_addc_sp,   "v_addc_co_u32 %0, vcc, %0, %8, vcc\n\tv_and_b32 %1, %1, %8\n\tv_and_b32 %2, %2, %8\n\tv_and_b32 %3, %3, %8\n\tv_addc_co_u32 %4, vcc, %4, %8, vcc\n\tv_and_b32 %5, %5, %8\n\tv_and_b32 %6, %6, %8\n\tv_and_b32 %7, %7, %8\n\t")
// 3-source, all VGPR
KERNEL(k_bitop3,    "v_bitop3_b32 %0, %0, %8, %9 bitop3:0xbe\n\tv_bitop3_b32 %1, %1, %8, %9 bitop3:0xbe\n\tv_bitop3_b32 %2, %2, %8, %9 bitop3:0xbe\n\tv_bitop3_b32 %3, %3, %8, %9 bitop3:0xbe\n\tv_bitop3_b32 %4, %4, %8, %9 bitop3:0xbe\n\tv_bitop3_b32 %5, %5, %8, %9 bitop3:0xbe\n\tv_bitop3_b32 %6, %6, %8, %9 bitop3:0xbe\n\tv_bitop3_b32 %7, %7, %8, %9 bitop3:0xbe\n\t")
KERNEL(k_bitop3_3d, "v_bitop3_b32 %0, %1, %2, %3 bitop3:0xbe\n\tv_bitop3_b32 %1, %2, %3, %4 bitop3:0xbe\n\tv_bitop3_b32 %2, %3, %4, %5 bitop3:0xbe\n\tv_bitop3_b32 %3, %4, %5, %6 bitop3:0xbe\n\tv_bitop3_b32 %4, %5, %6, %7 bitop3:0xbe\n\tv_bitop3_b32 %5, %6, %7, %0 bitop3:0xbe\n\tv_bitop3_b32 %6, %7, %0, %1 bitop3:0xbe\n\tv_bitop3_b32 %7, %0, %1, %2 bitop3:0xbe\n\t")
KERNEL(k_bitop3_2v, "v_bitop3_b32 %0, %0, %8, 1 bitop3:0xbe\n\tv_bitop3_b32 %1, %1, %8, 1 bitop3:0xbe\n\tv_bitop3_b32 %2, %2, %8, 1 bitop3:0xbe\n\tv_bitop3_b32 %3, %3, %8, 1 bitop3:0xbe\n\tv_bitop3_b32 %4, %4, %8, 1 bitop3:0xbe\n\tv_bitop3_b32 %5, %5, %8, 1 bitop3:0xbe\n\tv_bitop3_b32 %6, %6, %8, 1 bitop3:0xbe\n\tv_bitop3_b32 %7, %7, %8, 1 bitop3:0xbe\n\t")
KERNEL(k_alignbit,  "v_alignbit_b32 %0, %0, %8, 31\n\tv_alignbit_b32 %1, %1, %8, 31\n\tv_alignbit_b32 %2, %2, %8, 31\n\tv_alignbit_b32 %3, %3, %8, 31\n\tv_alignbit_b32 %4, %4, %8, 31\n\tv_alignbit_b32 %5, %5, %8, 31\n\tv_alignbit_b32 %6, %6, %8, 31\n\tv_alignbit_b32 %7, %7, %8, 31\n\t")
KERNEL(k_lshl_or,   "v_lshl_or_b32 %0, %0, 1, %8\n\tv_lshl_or_b32 %1, %1, 1, %8\n\tv_lshl_or_b32 %2, %2, 1, %8\n\tv_lshl_or_b32 %3, %3, 1, %8\n\tv_lshl_or_b32 %4, %4, 1, %8\n\tv_lshl_or_b32 %5, %5, 1, %8\n\tv_lshl_or_b32 %6, %6, 1, %8\n\tv_lshl_or_b32 %7, %7, 1, %8\n\t")
KERNEL(k_and_or,    "v_and_or_b32 %0, %0, %8, %9\n\tv_and_or_b32 %1, %1, %8, %9\n\tv_and_or_b32 %2, %2, %8, %9\n\tv_and_or_b32 %3, %3, %8, %9\n\tv_and_or_b32 %4, %4, %8, %9\n\tv_and_or_b32 %5, %5, %8, %9\n\tv_and_or_b32 %6, %6, %8, %9\n\tv_and_or_b32 %7, %7, %8, %9\n\t")
KERNEL(k_or3,       "v_or3_b32 %0, %0, %8, %9\n\tv_or3_b32 %1, %1, %8, %9\n\tv_or3_b32 %2, %2, %8, %9\n\tv_or3_b32 %3, %3, %8, %9\n\tv_or3_b32 %4, %4, %8, %9\n\tv_or3_b32 %5, %5, %8, %9\n\tv_or3_b32 %6, %6, %8, %9\n\tv_or3_b32 %7, %7, %8, %9\n\t")
KERNEL(k_bfi,       "v_bfi_b32 %0, %0, %8, %9\n\tv_bfi_b32 %1, %1, %8, %9\n\tv_bfi_b32 %2, %2, %8, %9\n\tv_bfi_b32 %3, %3, %8, %9\n\tv_bfi_b32 %4, %4, %8, %9\n\tv_bfi_b32 %5, %5, %8, %9\n\tv_bfi_b32 %6, %6, %8, %9\n\tv_bfi_b32 %7, %7, %8, %9\n\t")
KERNEL(k_add3,      "v_add3_u32 %0, %0, %8, %9\n\tv_add3_u32 %1, %1, %8, %9\n\tv_add3_u32 %2, %2, %8, %9\n\tv_add3_u32 %3, %3, %8, %9\n\tv_add3_u32 %4, %4, %8, %9\n\tv_add3_u32 %5, %5, %8, %9\n\tv_add3_u32 %6, %6, %8, %9\n\tv_add3_u32 %7, %7, %8, %9\n\t")
KERNEL(k_xad,       "v_xad_u32 %0, %0, %8, %9\n\tv_xad_u32 %1, %1, %8, %9\n\tv_xad_u32 %2, %2, %8, %9\n\tv_xad_u32 %3, %3, %8, %9\n\tv_xad_u32 %4, %4, %8, %9\n\tv_xad_u32 %5, %5, %8, %9\n\tv_xad_u32 %6, %6, %8, %9\n\tv_xad_u32 %7, %7, %8, %9\n\t")
KERNEL(k_fma,       "v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\tv_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9\n\t")
KERNEL(k_mad_u32_u24,"v_mad_u32_u24 %0, %0, %8, %9\n\tv_mad_u32_u24 %1, %1, %8, %9\n\tv_mad_u32_u24 %2, %2, %8, %9\n\tv_mad_u32_u24 %3, %3, %8, %9\n\tv_mad_u32_u24 %4, %4, %8, %9\n\tv_mad_u32_u24 %5, %5, %8, %9\n\tv_mad_u32_u24 %6, %6, %8, %9\n\tv_mad_u32_u24 %7, %7, %8, %9\n\t")
KERNEL(k_cndmask,   "v_cndmask_b32 %0, %0, %8, vcc\n\tv_cndmask_b32 %1, %1, %8, vcc\n\tv_cndmask_b32 %2, %2, %8, vcc\n\tv_cndmask_b32 %3, %3, %8, vcc\n\tv_cndmask_b32 %4, %4, %8, vcc\n\tv_cndmask_b32 %5, %5, %8, vcc\n\tv_cndmask_b32 %6, %6, %8, vcc\n\tv_cndmask_b32 %7, %7, %8, vcc\n\t")
// packed / 64-bit candidates
KERNEL(k_salu_mix,  "v_and_b32 %0, %0, %8\n\ts_nop 0\n\tv_and_b32 %1, %1, %8\n\ts_nop 0\n\tv_and_b32 %2, %2, %8\n\ts_nop 0\n\tv_and_b32 %3, %3, %8\n\ts_nop 0\n\tv_and_b32 %4, %4, %8\n\ts_nop 0\n\tv_and_b32 %5, %5, %8\n\ts_nop 0\n\tv_and_b32 %6, %6, %8\n\ts_nop 0\n\tv_and_b32 %7, %7, %8\n\ts_nop 0\n\t")

KERNEL(k_addco_rot, "v_add_co_u32 %0, s[72:73], %0, %8\n\tv_add_co_u32 %1, s[74:75], %1, %8\n\tv_add_co_u32 %2, s[76:77], %2, %8\n\tv_add_co_u32 %3, s[78:79], %3, %8\n\tv_add_co_u32 %4, s[72:73], %4, %8\n\tv_add_co_u32 %5, s[74:75], %5, %8\n\tv_add_co_u32 %6, s[76:77], %6, %8\n\tv_add_co_u32 %7, s[78:79], %7, %8\n\t")
KERNEL(k_addc_const, "v_addc_co_u32 %0, s[72:73], %0, %8, s[80:81]\n\tv_addc_co_u32 %1, s[74:75], %1, %8, s[80:81]\n\tv_addc_co_u32 %2, s[76:77], %2, %8, s[80:81]\n\tv_addc_co_u32 %3, s[78:79], %3, %8, s[80:81]\n\tv_addc_co_u32 %4, s[72:73], %4, %8, s[80:81]\n\tv_addc_co_u32 %5, s[74:75], %5, %8, s[80:81]\n\tv_addc_co_u32 %6, s[76:77], %6, %8, s[80:81]\n\tv_addc_co_u32 %7, s[78:79], %7, %8, s[80:81]\n\t")
KERNEL(k_cnd_const, "v_cndmask_b32 %0, %0, %8, s[80:81]\n\tv_cndmask_b32 %1, %1, %8, s[80:81]\n\tv_cndmask_b32 %2, %2, %8, s[80:81]\n\tv_cndmask_b32 %3, %3, %8, s[80:81]\n\tv_cndmask_b32 %4, %4, %8, s[80:81]\n\tv_cndmask_b32 %5, %5, %8, s[80:81]\n\tv_cndmask_b32 %6, %6, %8, s[80:81]\n\tv_cndmask_b32 %7, %7, %8, s[80:81]\n\t")
KERNEL(k_cmp_rot, "v_cmp_lt_u32 s[72:73], %0, %8\n\tv_cmp_lt_u32 s[74:75], %1, %8\n\tv_cmp_lt_u32 s[76:77], %2, %8\n\tv_cmp_lt_u32 s[78:79], %3, %8\n\tv_cmp_lt_u32 s[72:73], %4, %8\n\tv_cmp_lt_u32 s[74:75], %5, %8\n\tv_cmp_lt_u32 s[76:77], %6, %8\n\tv_cmp_lt_u32 s[78:79], %7, %8\n\t")
KERNEL(k_cmp_vcc, "v_cmp_lt_u32 vcc, %0, %8\n\tv_cmp_lt_u32 vcc, %1, %8\n\tv_cmp_lt_u32 vcc, %2, %8\n\tv_cmp_lt_u32 vcc, %3, %8\n\tv_cmp_lt_u32 vcc, %4, %8\n\tv_cmp_lt_u32 vcc, %5, %8\n\tv_cmp_lt_u32 vcc, %6, %8\n\tv_cmp_lt_u32 vcc, %7, %8\n\t")
KERNEL(k_readlane, "v_readfirstlane_b32 s72, %0\n\tv_readfirstlane_b32 s74, %1\n\tv_readfirstlane_b32 s76, %2\n\tv_readfirstlane_b32 s78, %3\n\tv_readfirstlane_b32 s72, %4\n\tv_readfirstlane_b32 s74, %5\n\tv_readfirstlane_b32 s76, %6\n\tv_readfirstlane_b32 s78, %7\n\t")
KERNEL(k_mix_addco_1in2, "v_add_co_u32 %0, s[72:73], %0, %8\n\tv_and_b32 %1, %1, %8\n\tv_add_co_u32 %2, s[74:75], %2, %8\n\tv_and_b32 %3, %3, %8\n\tv_add_co_u32 %4, s[76:77], %4, %8\n\tv_and_b32 %5, %5, %8\n\tv_add_co_u32 %6, s[78:79], %6, %8\n\tv_and_b32 %7, %7, %8\n\t")
KERNEL(k_mix_addco_1in4, "v_add_co_u32 %0, s[72:73], %0, %8\n\tv_and_b32 %1, %1, %8\n\tv_and_b32 %2, %2, %8\n\tv_and_b32 %3, %3, %8\n\tv_add_co_u32 %4, s[74:75], %4, %8\n\tv_and_b32 %5, %5, %8\n\tv_and_b32 %6, %6, %8\n\tv_and_b32 %7, %7, %8\n\t")
KERNEL(k_mix_addco_1in8, "v_add_co_u32 %0, s[72:73], %0, %8\n\tv_and_b32 %1, %1, %8\n\tv_and_b32 %2, %2, %8\n\tv_and_b32 %3, %3, %8\n\tv_and_b32 %4, %4, %8\n\tv_and_b32 %5, %5, %8\n\tv_and_b32 %6, %6, %8\n\tv_and_b32 %7, %7, %8\n\t")
KERNEL(k_mix_addc_1in4, "v_addc_co_u32 %0, s[72:73], %0, %8, s[80:81]\n\tv_and_b32 %1, %1, %8\n\tv_and_b32 %2, %2, %8\n\tv_and_b32 %3, %3, %8\n\tv_addc_co_u32 %4, s[74:75], %4, %8, s[80:81]\n\tv_and_b32 %5, %5, %8\n\tv_and_b32 %6, %6, %8\n\tv_and_b32 %7, %7, %8\n\t")
KERNEL(k_mix_cnd_1in4, "v_cndmask_b32 %0, %0, %8, s[80:81]\n\tv_and_b32 %1, %1, %8\n\tv_and_b32 %2, %2, %8\n\tv_and_b32 %3, %3, %8\n\tv_cndmask_b32 %4, %4, %8, s[80:81]\n\tv_and_b32 %5, %5, %8\n\tv_and_b32 %6, %6, %8\n\tv_and_b32 %7, %7, %8\n\t")
KERNEL(k_mix_vccaddc_1in4, "v_addc_co_u32 %0, vcc, %0, %8, vcc\n\tv_and_b32 %1, %1, %8\n\tv_and_b32 %2, %2, %8\n\tv_and_b32 %3, %3, %8\n\tv_addc_co_u32 %4, vcc, %4, %8, vcc\n\tv_and_b32 %5, %5, %8\n\tv_and_b32 %6, %6, %8\n\tv_and_b32 %7, %7, %8\n\t")
KERNEL(k_mix_align_1in4, "v_alignbit_b32 %0, %0, %8, 31\n\tv_and_b32 %1, %1, %8\n\tv_and_b32 %2, %2, %8\n\tv_and_b32 %3, %3, %8\n\tv_alignbit_b32 %4, %4, %8, 31\n\tv_and_b32 %5, %5, %8\n\tv_and_b32 %6, %6, %8\n\tv_and_b32 %7, %7, %8\n\t")

struct Entry { const char *name; void (*fn)(uint32_t *, int); int per_rep; };

int main(int argc, char **argv)
{
    int waves_per_simd = argc > 1 ? atoi(argv[1]) : 8;
    int iters = argc > 2 ? atoi(argv[2]) : 2000;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    int blocks = cus * waves_per_simd;  // 256 threads = 4 waves = one per SIMD
    uint32_t *out;
    CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    Entry tests[] = {
#define E(n, per) {#n, n, per}
        E(k_and_self, 8), E(k_xor, 8), E(k_add, 8), E(k_lshl, 8), E(k_not, 8), E(k_mov, 8), E(k_bcnt, 8),
        E(k_add_co, 8), E(k_addc_chain, 8), E(k_addc_sp, 8),
        E(k_bitop3, 8), E(k_bitop3_3d, 8), E(k_bitop3_2v, 8), E(k_alignbit, 8), E(k_lshl_or, 8), E(k_and_or, 8), E(k_or3, 8),
        E(k_bfi, 8), E(k_add3, 8), E(k_xad, 8), E(k_fma, 8), E(k_mad_u32_u24, 8), E(k_cndmask, 8), E(k_salu_mix, 8), E(k_addco_rot, 8), E(k_addc_const, 8), E(k_cnd_const, 8), E(k_cmp_rot, 8), E(k_cmp_vcc, 8), E(k_readlane, 8), E(k_mix_addco_1in2, 8), E(k_mix_addco_1in4, 8), E(k_mix_addco_1in8, 8), E(k_mix_addc_1in4, 8), E(k_mix_cnd_1in4, 8), E(k_mix_vccaddc_1in4, 8), E(k_mix_align_1in4, 8),
    };
    printf("device %s, %d CUs, clock %d MHz, %d waves/SIMD, iters %d\n", prop.gcnArchName, cus, prop.clockRate / 1000, waves_per_simd, iters);
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
        double insts_per_simd = (double)iters * 16 * t.per_rep * waves_per_simd;  // wave-instructions issued on one SIMD
        double ns_per_inst = ms * 1e6 / insts_per_simd;
        printf("%-16s %8.3f ms  %6.3f ns/inst/SIMD  = %5.2f cycles @2.4GHz  (%6.1f Tlane-ops/s chip)\n", t.name, ms, ns_per_inst,
               ns_per_inst * 2.4, 64.0 / ns_per_inst * cus * 4 / 1e3);
    }
    return 0;
}
