// long_kernels.hip — subjects longer than the register-resident kernels cover
// (Myers > 1024 bp, BitPAl > 256 bp): the DP state lives in a per-wave slice of the workspace
// instead of VGPRs.  Compiler-scheduled C++, same recurrences on the same Peq layout (32 data
// bits per word, hardware carries), so the scores are identical; only the speed differs
// (every word step pays global loads/stores that hit L2).  The reference has no length limit
// (its scratch is per-thread memory too, original/BGSA_CPU/align_core.c:49-52), so neither does
// the library.
//
// Grid: a fixed number of workgroups (kLongBlocks) loop over (group, query-tile) tasks, one
// state slice per resident wave — the slice count does not grow with the problem.
#include "bgsa_common.h"

namespace bgsa {

constexpr int kLongBlocks = 1024;  // 4 workgroups per CU

// ---- Myers -------------------------------------------------------------------------------------
// state slice: [word][2 = VP, VN][lane]
__global__ __launch_bounds__(256) void myers_long_kernel(
    const char *__restrict__ content, const uint32_t *__restrict__ peq, int16_t *__restrict__ out,
    uint32_t *__restrict__ state_all, int ref_len, int read_len, long long ld, int n_groups, int word_num,
    int ref_start, int ref_end, int q_tile)
{
    const int lane = threadIdx.x & (kLanes - 1);
    const int wave = threadIdx.x >> 6;
    uint32_t *st = state_all + (static_cast<size_t>(blockIdx.x) * kWavesPerBlock + wave) * word_num * 2 * kLanes + lane;
    const int nq = ref_end - ref_start;
    const int q_tiles = (nq + q_tile - 1) / q_tile;
    const long long n_tasks = static_cast<long long>((n_groups + kWavesPerBlock - 1) / kWavesPerBlock) * q_tiles;
    for (long long task = blockIdx.x; task < n_tasks; task += gridDim.x) {
        const int group = static_cast<int>(task / q_tiles) * kWavesPerBlock + wave;
        const int tile = static_cast<int>(task % q_tiles);
        if (group >= n_groups) continue;  // wave-uniform
        const uint32_t *g = peq + static_cast<size_t>(group) * kChars * word_num * kLanes + lane;
        const int q0 = ref_start + tile * q_tile;
        const int q1 = (q0 + q_tile < ref_end) ? q0 + q_tile : ref_end;
        for (int q = q0; q < q1; q++) {
            for (int w = 0; w < word_num; w++) {
                st[(2 * w) * kLanes] = ~0u;
                st[(2 * w + 1) * kLanes] = 0u;
            }
            UniformBytes qs(content + static_cast<size_t>(q) * (ref_len + 1));
            for (int r = 0; r < ref_len; r++) {
                if ((r & 3) == 0) qs.refill(r, ref_len - r);
                uint32_t c = __builtin_amdgcn_readfirstlane(qs.next());
                if (c > 4) c = 0;
                const uint32_t *eq = g + static_cast<size_t>(c) * word_num * kLanes;
                uint32_t carry = 0, hp_in = 1, hn_in = 0;
                for (int w = 0; w < word_num; w++) {
                    const uint32_t e = eq[w * kLanes];
                    const uint32_t pv = st[(2 * w) * kLanes], mv = st[(2 * w + 1) * kLanes];
                    const unsigned long long s = static_cast<unsigned long long>(pv & e) + pv + carry;
                    carry = static_cast<uint32_t>(s >> 32);
                    const uint32_t d0 = ((static_cast<uint32_t>(s)) ^ pv) | e | mv;
                    const uint32_t hp = ~(d0 | pv) | mv;
                    const uint32_t hn = d0 & pv;
                    const uint32_t hps = (hp << 1) | hp_in;
                    const uint32_t hns = (hn << 1) | hn_in;
                    hp_in = hp >> 31;
                    hn_in = hn >> 31;
                    st[(2 * w) * kLanes] = ~(d0 | hps) | hns;
                    st[(2 * w + 1) * kLanes] = d0 & hps;
                }
            }
            int score = ref_len;
            for (int w = 0; w < word_num; w++) {
                const int rem = read_len - 32 * w;
                const uint32_t m = rem >= 32 ? ~0u : (rem <= 0 ? 0u : ((1u << rem) - 1u));
                score += __popc(st[(2 * w) * kLanes] & m) - __popc(st[(2 * w + 1) * kLanes] & m);
            }
            out[static_cast<size_t>(q - ref_start) * ld + static_cast<size_t>(group) * kLanes + lane] =
                static_cast<int16_t>(-score);
        }
    }
}

// ---- BitPAl (2, -3, -5) ----------------------------------------------------------------------------
struct BitpalCarry {
    uint32_t ov[5], seed[4], plane[4];
};

__device__ __forceinline__ uint32_t add_c(uint32_t a, uint32_t b, uint32_t &carry)
{
    const unsigned long long s = static_cast<unsigned long long>(a) + b + carry;
    carry = static_cast<uint32_t>(s >> 32);
    return static_cast<uint32_t>(s);
}

// One (row, word) step, the word-serial form of rows_ir.py:bitpal_body (reference
// original/BGSA_AVX2/align_core.c:183-428 on full 32-bit words).
__device__ __forceinline__ void bitpal_word_step(uint32_t match, uint32_t (&h)[5], BitpalCarry &cy)
{
    const uint32_t nm = ~match;
    const uint32_t top = h[4] & h[3];
    const uint32_t o3 = h[2] | h[1] | h[0];
    const uint32_t neg5 = ~(h[4] | h[3] | o3);
    const uint32_t bb = top & h[2];
    const uint32_t neg1 = bb & ~h[1] & ~h[0], neg2 = bb & ~h[1] & h[0], neg3 = bb & h[1] & ~h[0], neg4 = bb & h[1] & h[0];
    const uint32_t any = ((top & o3) | neg5) & nm;
    const uint32_t run = neg5 & nm;
    const uint32_t sum = add_c(neg5 & match, neg5, cy.ov[0]);
    const uint32_t dv7m = (sum ^ run) | match;
    auto shifted_run = [&](uint32_t seed, int i) {
        const uint32_t v = add_c(seed, seed, cy.seed[i]);       // seed << 1 across words
        const uint32_t s = add_c(v, run, cy.ov[i + 1]);
        return (s ^ run) & nm;
    };
    const uint32_t dv6 = shifted_run(neg4 & dv7m, 0);
    const uint32_t dv5 = shifted_run((neg3 & dv7m) | (neg4 & dv6), 1);
    const uint32_t dv4 = shifted_run((neg2 & dv7m) | (neg3 & dv6) | (neg4 & dv5), 2);
    const uint32_t dv3 = shifted_run((neg1 & dv7m) | (neg2 & dv6) | (neg3 & dv5) | (neg4 & dv4), 3);
    const uint32_t rest = ~(dv7m | dv6 | dv5 | dv4 | dv3);
    const uint32_t v[5] = {rest | dv4 | dv6, dv5 | dv6 | rest, rest | dv7m, ~rest, 0u};
    uint32_t t[5], c = 0;
    for (int i = 0; i < 5; i++) {  // bit-sliced 5-bit add, dH + dV
        const uint32_t x = h[i] ^ v[i];
        t[i] = x ^ c;
        c = (h[i] & v[i]) | (x & c);
    }
    for (int i = 0; i < 4; i++) t[i] = add_c(t[i] & ~t[4], t[i] & ~t[4], cy.plane[i]);  // clamp, shift one column up
    const uint32_t g[5] = {(h[0] | any) & nm, h[1] & ~any & nm, (h[2] & ~any) | match, (h[3] | any) & nm,
                           h[4] | any | match};
    uint32_t r[5];
    c = 0;
    for (int i = 0; i < 5; i++) {
        const uint32_t ti = i < 4 ? t[i] : 0u;
        const uint32_t x = g[i] ^ ti;
        r[i] = x ^ c;
        c = (g[i] & ti) | (x & c);
    }
    for (int i = 0; i < 4; i++) h[i] = r[i] & r[4];
    h[4] = r[4];
}

// state slice: [word][5 planes][lane]
__global__ __launch_bounds__(256) void bitpal_long_kernel(
    const char *__restrict__ content, const uint32_t *__restrict__ peq, int16_t *__restrict__ out,
    uint32_t *__restrict__ state_all, int ref_len, int read_len, long long ld, int n_groups, int word_num,
    int ref_start, int ref_end, int q_tile)
{
    const int lane = threadIdx.x & (kLanes - 1);
    const int wave = threadIdx.x >> 6;
    uint32_t *st = state_all + (static_cast<size_t>(blockIdx.x) * kWavesPerBlock + wave) * word_num * 5 * kLanes + lane;
    const int nq = ref_end - ref_start;
    const int q_tiles = (nq + q_tile - 1) / q_tile;
    const long long n_tasks = static_cast<long long>((n_groups + kWavesPerBlock - 1) / kWavesPerBlock) * q_tiles;
    for (long long task = blockIdx.x; task < n_tasks; task += gridDim.x) {
        const int group = static_cast<int>(task / q_tiles) * kWavesPerBlock + wave;
        const int tile = static_cast<int>(task % q_tiles);
        if (group >= n_groups) continue;
        const uint32_t *g = peq + static_cast<size_t>(group) * kChars * word_num * kLanes + lane;
        const int q0 = ref_start + tile * q_tile;
        const int q1 = (q0 + q_tile < ref_end) ? q0 + q_tile : ref_end;
        for (int q = q0; q < q1; q++) {
            for (int i = 0; i < 5 * word_num; i++) st[i * kLanes] = 0u;
            UniformBytes qs(content + static_cast<size_t>(q) * (ref_len + 1));
            for (int r = 0; r < ref_len; r++) {
                if ((r & 3) == 0) qs.refill(r, ref_len - r);
                uint32_t c = __builtin_amdgcn_readfirstlane(qs.next());
                if (c > 4) c = 0;
                const uint32_t *eq = g + static_cast<size_t>(c) * word_num * kLanes;
                BitpalCarry cy = {};
                for (int w = 0; w < word_num; w++) {
                    uint32_t h[5];
                    for (int i = 0; i < 5; i++) h[i] = st[(5 * w + i) * kLanes];
                    bitpal_word_step(eq[w * kLanes], h, cy);
                    for (int i = 0; i < 5; i++) st[(5 * w + i) * kLanes] = h[i];
                }
            }
            int score = -5 * ref_len - 5 * read_len;
            for (int w = 0; w < word_num; w++) {
                const int rem = read_len - 32 * w;
                const uint32_t m = rem >= 32 ? ~0u : (rem <= 0 ? 0u : ((1u << rem) - 1u));
                score += 16 * __popc(st[(5 * w + 4) * kLanes] & m) - 8 * __popc(st[(5 * w + 3) * kLanes] & m) -
                         4 * __popc(st[(5 * w + 2) * kLanes] & m) - 2 * __popc(st[(5 * w + 1) * kLanes] & m) -
                         __popc(st[(5 * w) * kLanes] & m);
            }
            out[static_cast<size_t>(q - ref_start) * ld + static_cast<size_t>(group) * kLanes + lane] =
                static_cast<int16_t>(score);
        }
    }
}

size_t long_state_bytes(int algo, int word_num)
{
    const size_t per_wave = static_cast<size_t>(word_num) * (algo == BGSA_ALGO_BITPAL ? 5 : 2) * kLanes * sizeof(uint32_t);
    return per_wave * kWavesPerBlock * kLongBlocks;
}

int launch_long(int algo, const char *d_content, const uint32_t *d_peq, int16_t *d_results, int ref_len,
                int read_len, int64_t read_count, int ref_start, int ref_end, int word_num, void *d_state,
                hipStream_t stream)
{
    const int n_groups = static_cast<int>(read_count / kLanes);
    const int q_tile = 4;
    if (algo == BGSA_ALGO_BITPAL)
        hipLaunchKernelGGL(bitpal_long_kernel, dim3(kLongBlocks), dim3(256), 0, stream, d_content, d_peq, d_results,
                           static_cast<uint32_t *>(d_state), ref_len, read_len, static_cast<long long>(read_count),
                           n_groups, word_num, ref_start, ref_end, q_tile);
    else
        hipLaunchKernelGGL(myers_long_kernel, dim3(kLongBlocks), dim3(256), 0, stream, d_content, d_peq, d_results,
                           static_cast<uint32_t *>(d_state), ref_len, read_len, static_cast<long long>(read_count),
                           n_groups, word_num, ref_start, ref_end, q_tile);
    BGSA_HIP_TRY(hipGetLastError());
    return BGSA_HIP_OK;
}

}  // namespace bgsa
