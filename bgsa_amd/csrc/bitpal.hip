// bitpal.hip — BitPAl packed scoring, match 2 / mismatch -3 / gap -5, one subject per lane, gfx950.
//
// Replaces the reference's align_avx hot loop (original/BGSA_AVX2/align_core.c:164-482).  Same
// decomposition as myers_global.hip: lane = subject, wave = group of 64, the wave keeps its Peq
// block in VGPRs and walks a tile of queries; the row loop is the generated threaded-code asm of
// bitpal_rows_gen.inc (rows_ir.py:bitpal_body — 76 fast-class VALU per (row, word) against the
// reference's 194, all fourteen inter-word carries as VCC add-with-carry chains on full 32-bit
// words).  The per-column state is five bit-planes per word (25 VGPRs at 150 bp).
//
// Final score (align_core.c:433-471): -5*qlen + sum over subject columns of
// (16*b16 - 8*b8 - 4*b4 - 2*b2 - b1 - 5), i.e. five masked popcounts per word.
#include <stdlib.h>

#include "bgsa_common.h"

namespace bgsa {

#include "bitpal_rows_gen.inc"
static_assert(kBitpalChains == 13, "capi.hip sizes the carry buffers for 13 chains");

template <int NW>
__global__ __launch_bounds__(256) void bitpal_asm_kernel(
    const unsigned char *__restrict__ streams, const uint32_t *__restrict__ peq,
    int16_t *__restrict__ out, int ref_len, int read_len, long long ld, int n_groups, int word_num,
    int n_queries, int q_tile, int stream_stride_bytes)
{
    const int lane = threadIdx.x & (kLanes - 1);
    const int group = __builtin_amdgcn_readfirstlane(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));
    if (group >= n_groups) return;

    uint32_t P[kChars][NW];
    const uint32_t *g = peq + static_cast<size_t>(group) * kChars * word_num * kLanes + lane;
#pragma unroll
    for (int c = 0; c < kChars; c++)
#pragma unroll
        for (int w = 0; w < NW; w++)
            P[c][w] = (w < word_num) ? g[(c * word_num + w) * kLanes] : 0u;

    const int q0 = blockIdx.y * q_tile;
    const int q1 = (q0 + q_tile < n_queries) ? q0 + q_tile : n_queries;
    int16_t *dst = out + static_cast<size_t>(group) * kLanes + lane;

    for (int q = q0; q < q1; q++) {
        uint32_t st[5 * NW];
#pragma unroll
        for (int i = 0; i < 5 * NW; i++) st[i] = 0u;  // every column starts at dH = -5 (:167-171)
        const unsigned long long s =
            reinterpret_cast<unsigned long long>(streams) + static_cast<unsigned long long>(q) * stream_stride_bytes;
        bitpal_rows_asm<NW>(st, P, uniform_u64(s), __builtin_amdgcn_readfirstlane(stream_stride_bytes / 8 - 2));
        int score = -5 * ref_len - 5 * read_len;
#pragma unroll
        for (int w = 0; w < NW; w++) {
            const int rem = read_len - 32 * w;
            const uint32_t m = rem >= 32 ? ~0u : (rem <= 0 ? 0u : ((1u << rem) - 1u));
            score += 16 * __popc(st[w * 5 + 4] & m) - 8 * __popc(st[w * 5 + 3] & m) -
                     4 * __popc(st[w * 5 + 2] & m) - 2 * __popc(st[w * 5 + 1] & m) -
                     __popc(st[w * 5 + 0] & m);
        }
        dst[static_cast<size_t>(q) * ld] = static_cast<int16_t>(score);
    }
}

// Subjects longer than 256 bp: column blocks of NW words, the thirteen inter-word carry chains of a
// row crossing block boundaries through per-wave carry words (same scheme as myers_blocked_kernel;
// rows_ir.py: make_blocked, CPU-simulated in tests/test_rows_ir.py).
template <int NW>
__global__ __launch_bounds__(256) void bitpal_blocked_kernel(
    const unsigned char *__restrict__ streams, const uint32_t *__restrict__ peq, int16_t *__restrict__ out,
    uint32_t *__restrict__ carry_all, int ref_len, int read_len, long long ld, int n_groups, int word_num,
    int n_queries, int q_tile, int stream_stride_bytes, int n_blocks)
{
    constexpr int NC = kBitpalChains;
    const int lane = threadIdx.x & (kLanes - 1);
    const int wave = threadIdx.x >> 6;
    const int n_chunks = (ref_len + 31) / 32;
    uint32_t *carry = carry_all + (static_cast<size_t>(blockIdx.x) * kWavesPerBlock + wave) * n_chunks * NC * kLanes;
    const unsigned long long carry_base = uniform_u64(reinterpret_cast<unsigned long long>(carry));
    const int q_tiles = (n_queries + q_tile - 1) / q_tile;
    const long long n_tasks = static_cast<long long>((n_groups + kWavesPerBlock - 1) / kWavesPerBlock) * q_tiles;
    const int tail_rows = ref_len & 31;

    for (long long task = blockIdx.x; task < n_tasks; task += gridDim.x) {
        const int group = __builtin_amdgcn_readfirstlane(static_cast<int>(task / q_tiles) * kWavesPerBlock + wave);
        const int tile = static_cast<int>(task % q_tiles);
        if (group >= n_groups) continue;  // wave-uniform
        const uint32_t *g = peq + static_cast<size_t>(group) * kChars * word_num * kLanes + lane;
        const int q0 = tile * q_tile;
        const int q1 = (q0 + q_tile < n_queries) ? q0 + q_tile : n_queries;
        for (int q = q0; q < q1; q++) {
            for (int i = 0; i < n_chunks * NC; i++) carry[i * kLanes + lane] = 0u;  // every chain starts at carry-in 0
            const unsigned long long s =
                reinterpret_cast<unsigned long long>(streams) + static_cast<unsigned long long>(q) * stream_stride_bytes;
            int score = -5 * ref_len - 5 * read_len;
            for (int blk = 0; blk < n_blocks; blk++) {
                uint32_t P[kChars][NW];
#pragma unroll
                for (int c = 0; c < kChars; c++)
#pragma unroll
                    for (int w = 0; w < NW; w++) {
                        const int gw = blk * NW + w;
                        P[c][w] = (gw < word_num) ? g[(c * word_num + gw) * kLanes] : 0u;
                    }
                uint32_t st[5 * NW + 2 * NC];
#pragma unroll
                for (int i = 0; i < 5 * NW; i++) st[i] = 0u;
#pragma unroll
                for (int i = 0; i < NC; i++) {
                    st[5 * NW + i] = carry[i * kLanes + lane];  // chunk 0
                    st[5 * NW + NC + i] = 0u;
                }
                uint32_t voff = static_cast<uint32_t>(lane * 4);
                bitpal_block_rows_asm<NW>(st, P, voff, carry_base, uniform_u64(s),
                                          __builtin_amdgcn_readfirstlane(stream_stride_bytes / 8 - 2));
#pragma unroll
                for (int i = 0; i < NC; i++) {
                    const uint32_t word = tail_rows ? (st[5 * NW + NC + i] << (32 - tail_rows)) : st[5 * NW + NC + i];
                    carry[((n_chunks - 1) * NC + i) * kLanes + lane] = word;
                }
#pragma unroll
                for (int w = 0; w < NW; w++) {
                    const int rem = read_len - 32 * (blk * NW + w);
                    const uint32_t m = rem >= 32 ? ~0u : (rem <= 0 ? 0u : ((1u << rem) - 1u));
                    score += 16 * __popc(st[w * 5 + 4] & m) - 8 * __popc(st[w * 5 + 3] & m) -
                             4 * __popc(st[w * 5 + 2] & m) - 2 * __popc(st[w * 5 + 1] & m) - __popc(st[w * 5 + 0] & m);
                }
            }
            out[static_cast<size_t>(q) * ld + static_cast<size_t>(group) * kLanes + lane] = static_cast<int16_t>(score);
        }
    }
}

namespace {

int bitpal_impl()
{
    static const int impl = [] {
        const char *e = getenv("BGSA_BITPAL_IMPL");
        return (e && e[0] == 'c') ? 1 : 0;
    }();
    return impl;
}

// Narrowest instantiated block width that covers word_num > 8 words with the fewest blocks.
int pick_block_nw(int word_num, int *n_blocks)
{
    const int blocks = (word_num + 7) / 8;
    const int need = (word_num + blocks - 1) / blocks;
    const int nw = need < 5 ? 5 : need;
    *n_blocks = (word_num + nw - 1) / nw;
    return nw;
}

template <int NW>
int launch_blocked(const char *d_content, const uint32_t *d_peq, int16_t *d_results, int ref_len, int read_len,
                   int64_t read_count, int ref_start, int ref_end, int word_num, int n_blocks, void *d_workspace,
                   hipStream_t stream)
{
    const int nq = ref_end - ref_start;
    const int stride = blocked_stream_layout(ref_len, nullptr, nullptr);
    const size_t stream_bytes = (static_cast<size_t>(stride) * nq + 255) & ~static_cast<size_t>(255);
    if (int rc = launch_pack_blocked(d_content, ref_len, ref_start, ref_end, d_workspace, stream)) return rc;
    uint32_t *carry = reinterpret_cast<uint32_t *>(static_cast<unsigned char *>(d_workspace) + stream_bytes);
    hipLaunchKernelGGL((bitpal_blocked_kernel<NW>), dim3(kBlockedBlocks), dim3(256), 0, stream,
                       static_cast<const unsigned char *>(d_workspace), d_peq, d_results, carry, ref_len, read_len,
                       static_cast<long long>(read_count), static_cast<int>(read_count / kLanes), word_num, nq, 2,
                       stride, n_blocks);
    BGSA_HIP_TRY(hipGetLastError());
    return BGSA_HIP_OK;
}

int pick_q_tile(int nq, int64_t n_groups)
{
    int q_tile = 16;
    while (q_tile > 1 && ((nq + q_tile - 1) / q_tile) * ((n_groups + 3) / 4) < 4096) q_tile >>= 1;
    return q_tile;
}

template <int NW>
int launch_nw(const char *d_content, const uint32_t *d_peq, int16_t *d_results, int ref_len,
              int read_len, int64_t read_count, int ref_start, int ref_end, int word_num,
              void *d_workspace, hipStream_t stream)
{
    const int nq = ref_end - ref_start;
    const int64_t n_groups = read_count / kLanes;
    const int q_tile = pick_q_tile(nq, n_groups);
    dim3 grid(static_cast<unsigned>((n_groups + kWavesPerBlock - 1) / kWavesPerBlock),
              static_cast<unsigned>((nq + q_tile - 1) / q_tile));
    if (grid.y > 65535u) {
        set_error_text("bitpal: too many query tiles for one launch");
        return BGSA_HIP_EUNSUPPORTED;
    }
    if (int rc = launch_pack_queries(d_content, ref_len, ref_start, ref_end, d_workspace, stream)) return rc;
    hipLaunchKernelGGL(bitpal_asm_kernel<NW>, grid, dim3(256), 0, stream,
                       static_cast<const unsigned char *>(d_workspace), d_peq, d_results, ref_len,
                       read_len, static_cast<long long>(read_count), static_cast<int>(n_groups), word_num,
                       nq, q_tile, static_cast<int>(stream_stride(ref_len)));
    BGSA_HIP_TRY(hipGetLastError());
    return BGSA_HIP_OK;
}

}  // namespace

const char *bitpal_kernel_name(int word_num)
{
    static thread_local char name[64];
    if (word_num > 8) {
        int n_blocks = 0;
        snprintf(name, sizeof name, bitpal_impl() ? "bitpal_long_kernel" : "bitpal_blocked_kernel<%d>",
                 pick_block_nw(word_num, &n_blocks));
        return name;
    }
    snprintf(name, sizeof name, "bitpal_asm_kernel<%d>", word_num);
    return name;
}

int launch_bitpal(const char *d_content, const uint32_t *d_peq, int16_t *d_results, int ref_len,
                  int read_len, int64_t read_count, int ref_start, int ref_end, int word_num,
                  void *d_workspace, hipStream_t stream)
{
    if (ref_end <= ref_start || read_count == 0) return BGSA_HIP_OK;
    if (word_num > 8 && bitpal_impl() == 1)  // A/B: the state-in-memory C++ kernel
        return launch_long(BGSA_ALGO_BITPAL, d_content, d_peq, d_results, ref_len, read_len, read_count, ref_start,
                           ref_end, word_num, d_workspace, stream);
    if (word_num > 8) {
        int n_blocks = 0;
        switch (pick_block_nw(word_num, &n_blocks)) {
#define BGSA_BLOCK_CASE(N)                                                                       \
    case N:                                                                                      \
        return launch_blocked<N>(d_content, d_peq, d_results, ref_len, read_len, read_count, ref_start, \
                                 ref_end, word_num, n_blocks, d_workspace, stream);
            BGSA_BLOCK_CASE(5) BGSA_BLOCK_CASE(6) BGSA_BLOCK_CASE(7) BGSA_BLOCK_CASE(8)
#undef BGSA_BLOCK_CASE
        default: break;
        }
    }
    switch (word_num) {
#define BGSA_CASE(N)                                                                            \
    case N:                                                                                     \
        return launch_nw<N>(d_content, d_peq, d_results, ref_len, read_len, read_count,         \
                            ref_start, ref_end, word_num, d_workspace, stream);
        BGSA_CASE(1) BGSA_CASE(2) BGSA_CASE(3) BGSA_CASE(4) BGSA_CASE(5) BGSA_CASE(6)
        BGSA_CASE(7) BGSA_CASE(8)
#undef BGSA_CASE
    default:
        set_error_text("bitpal: no kernel for this word count");
        return BGSA_HIP_EUNSUPPORTED;
    }
}

}  // namespace bgsa
