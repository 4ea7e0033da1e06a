// preprocess.hip — subject preprocess (ASCII rows -> per-lane Peq match masks) and query mapping.
//
// Replaces <arch>_handle_reads (reference original/BGSA_CPU/global.c:25-70; banded form
// banded/BGSA_CPU/global.c:25-84) and the in-place query map of get_ref_from_file
// (original/BGSA_CPU/file.c:134-139), both as a host routine (the reference's seam) and as GPU
// kernels so the host only ships raw rows.
//
// Layout produced, for every group of 64 consecutive subjects:
//     peq[group][char 0..4][word 0..word_num-1][lane 0..63]
// Global Myers / BitPAl: 32-bit words, bit (p mod D) of word (p div D) set iff subject[p] maps to
// `char`, D = 32 data bits per word (the reference keeps one bit per word as a software carry).
// Banded: the match bit-string offset by k+1 zero bits, 32-bit words (see "banded layout" below).
#include <algorithm>
#include <thread>
#include <vector>

#include "bgsa_common.h"

namespace bgsa {

// init_mapping_table, reference original/BGSA_CPU/global.c:9-15: A C G T N -> 0..4, rest -> 0.
__host__ __device__ __forceinline__ uint32_t map_char(uint32_t ch)
{
    return ch == 'C' ? 1u : ch == 'G' ? 2u : ch == 'T' ? 3u : ch == 'N' ? 4u : 0u;
}

// ---- global layout (Myers / BitPAl) -------------------------------------------------------------

// One lane builds all words of its own subject.  Lanes read their rows byte by byte (stride
// len+1 across lanes, sequential per lane, so every 128-B line fetched is fully consumed from L1
// over the following iterations) and write coalesced 256-B rows of the block.
__global__ __launch_bounds__(256) void preprocess_global_kernel(const char *__restrict__ rows,
                                                                uint32_t *__restrict__ peq, int len,
                                                                long long read_count, int word_num,
                                                                int data_bits)
{
    const long long subject = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (subject >= read_count) return;
    const long long group = subject >> 6;
    const int lane = static_cast<int>(subject & 63);
    const char *row = rows + subject * (len + 1);
    uint32_t *dst = peq + static_cast<size_t>(group) * kChars * word_num * kLanes + lane;
    for (int w = 0; w < word_num; w++) {
        uint32_t m[kChars] = {0, 0, 0, 0, 0};
        const int base = w * data_bits;
        const int n = min(data_bits, len - base);
        for (int b = 0; b < n; b++) {
            const uint32_t c = map_char(static_cast<uint8_t>(row[base + b]));
#pragma unroll
            for (uint32_t cc = 0; cc < kChars; cc++) m[cc] |= static_cast<uint32_t>(c == cc) << b;
        }
#pragma unroll
        for (int cc = 0; cc < kChars; cc++) dst[(cc * word_num + w) * kLanes] = m[cc];
    }
}

// Short reads (64 rows fit in 16 KB of LDS): the 64 rows of a group are one contiguous byte range
// of the input, so the wave stages them with fully coalesced 16-byte loads and each lane then
// walks its own row in LDS — the global reads are at streaming efficiency instead of 64 strided
// byte streams.  One wave per group, four groups per workgroup.
__global__ __launch_bounds__(256) void preprocess_global_lds_kernel(const char *__restrict__ rows,
                                                                    uint32_t *__restrict__ peq, int len,
                                                                    long long n_groups, int word_num,
                                                                    long long total_bytes)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char stage[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long group = static_cast<long long>(blockIdx.x) * kWavesPerBlock + wave;
    const int row_bytes = len + 1;
    const int group_bytes = 64 * row_bytes;
    const int slot_bytes = (group_bytes + 15 + 16) & ~15;  // +16: the group may start mid-16-byte line
    unsigned char *mine = stage + wave * slot_bytes;
    if (group < n_groups) {
        const long long first = group * group_bytes;                 // byte offset of the group's rows
        const long long aligned = first & ~15LL;
        const int skew = static_cast<int>(first - aligned);
        const uint4 *src = reinterpret_cast<const uint4 *>(rows + aligned);
        const int n16 = (skew + group_bytes + 15) / 16;
        for (int i = lane; i < n16; i += 64) {
            uint4 v = make_uint4(0, 0, 0, 0);
            const long long at = aligned + 16LL * i;
            if (at + 16 <= total_bytes) {
                v = src[i];
            } else if (at < total_bytes) {   // the input's last, partial 16-byte line: not one byte past avail_bytes
                uint32_t w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
                for (int b = 0; b < 16; b++)
                    if (at + b < total_bytes) w[b >> 2] |= static_cast<uint32_t>(static_cast<unsigned char>(rows[at + b])) << (8 * (b & 3));
                v = make_uint4(w[0], w[1], w[2], w[3]);
            }
            reinterpret_cast<uint4 *>(mine)[i] = v;
        }
        __builtin_amdgcn_wave_barrier();
        const unsigned char *row = mine + skew + lane * row_bytes;
        uint32_t *dst = peq + static_cast<size_t>(group) * kChars * word_num * kLanes + lane;
        for (int w = 0; w < word_num; w++) {
            uint32_t m[kChars] = {0, 0, 0, 0, 0};
            const int base = w * 32;
            const int n = min(32, len - base);
            for (int b = 0; b < n; b++) {
                const uint32_t c = map_char(row[base + b]);
#pragma unroll
                for (uint32_t cc = 0; cc < kChars; cc++) m[cc] |= static_cast<uint32_t>(c == cc) << b;
            }
#pragma unroll
            for (int cc = 0; cc < kChars; cc++) dst[(cc * word_num + w) * kLanes] = m[cc];
        }
    }
}

// ---- banded layout ------------------------------------------------------------------------------
// "Mext": per character class, the subject's match bit-string offset by k+1 zero bits — bit i is
// set iff i >= k+1 and subject[i-(k+1)] maps to the class — in 32-bit words.  Word 0 of the
// reference's layout (first k characters at bits k+1..2k, banded/BGSA_CPU/global.c:45-62) is Mext
// bits 0..2k, and its per-row shift-and-feed (align_core.c:35-62) walks the same string one bit per
// row, so the window the reference holds at row r is Mext bits r .. r+2k (banded.hip).
// word_num = ceil(len / 32) + 3: the 64-bit window of the last row and the prefetch stay in bounds.
__global__ __launch_bounds__(256) void preprocess_banded_kernel(const char *__restrict__ rows,
                                                                uint32_t *__restrict__ mext, int len,
                                                                long long read_count, int word_num, int k)
{
    const long long subject = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (subject >= read_count) return;
    const long long group = subject >> 6;
    const int lane = static_cast<int>(subject & 63);
    const char *row = rows + subject * (len + 1);
    uint32_t *dst = mext + static_cast<size_t>(group) * kChars * word_num * kLanes + lane;
    for (int w = 0; w < word_num; w++) {
        uint32_t m[kChars] = {0, 0, 0, 0, 0};
        for (int b = 0; b < 32; b++) {
            const int p = w * 32 + b - (k + 1);
            if (p < 0 || p >= len) continue;
            const uint32_t c = map_char(static_cast<uint8_t>(row[p]));
#pragma unroll
            for (uint32_t cc = 0; cc < kChars; cc++) m[cc] |= static_cast<uint32_t>(c == cc) << b;
        }
#pragma unroll
        for (int cc = 0; cc < kChars; cc++) dst[(cc * word_num + w) * kLanes] = m[cc];
    }
}

// ---- query map ------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void map_queries_kernel(char *content, long long bytes)
{
    const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= bytes) return;
    const uint8_t ch = static_cast<uint8_t>(content[i]);
    if (ch != '\n') content[i] = static_cast<char>(map_char(ch));
}

// ---- query packer -----------------------------------------------------------------------------------
// mapped query rows (codes 0..4, stride ref_len+1) -> code streams of 8-byte windows
// (bgsa_common.h "Packed query stream").  One thread per (query, window).
__global__ __launch_bounds__(256) void pack_queries_kernel(const char *__restrict__ content,
                                                           unsigned long long *__restrict__ streams,
                                                           int ref_len, int ref_start, int n_queries,
                                                           int n_windows, unsigned *__restrict__ zero_word)
{
    const long long tid = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (zero_word && tid == 0) *zero_word = 0u;   // the scoring launch's task counter: zeroed here instead of by a memset of its own
    const int per = n_windows + 1;  // + the spare all-END window
    if (tid >= static_cast<long long>(n_queries) * per) return;
    const int q = static_cast<int>(tid / per), i = static_cast<int>(tid % per);
    const char *row = content + static_cast<size_t>(ref_start + q) * (ref_len + 1);
    const unsigned long long win = plain_stream_window(row, ref_len, i);
    streams[tid] = win;
}

int launch_pack_queries(const char *d_content, int ref_len, int ref_start, int ref_end,
                        void *d_streams, hipStream_t stream, unsigned *d_zero_word)
{
    const int nq = ref_end - ref_start;
    if (nq <= 0) return BGSA_HIP_OK;
    const int n_windows = stream_windows(ref_len);
    const long long total = static_cast<long long>(nq) * (n_windows + 1);
    hipLaunchKernelGGL(pack_queries_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0,
                       stream, d_content, static_cast<unsigned long long *>(d_streams), ref_len, ref_start,
                       nq, n_windows, d_zero_word);
    BGSA_HIP_TRY(hipGetLastError());
    return BGSA_HIP_OK;
}

__global__ __launch_bounds__(256) void pack_query_pairs_kernel(const char *__restrict__ content,
                                                               unsigned long long *__restrict__ streams,
                                                               int ref_len, int ref_start, int n_queries,
                                                               int n_windows, unsigned *__restrict__ zero_word)
{
    const long long tid = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (zero_word && tid == 0) *zero_word = 0u;   // as in pack_queries_kernel
    const int per = n_windows + 1;  // + the spare all-END window
    if (tid >= static_cast<long long>(n_queries) * per) return;
    const int q = static_cast<int>(tid / per), i = static_cast<int>(tid % per);
    const char *row = content + static_cast<size_t>(ref_start + q) * (ref_len + 1);
    streams[tid] = pair_stream_window(row, ref_len, i);
}

int launch_pack_query_pairs(const char *d_content, int ref_len, int ref_start, int ref_end,
                            void *d_streams, hipStream_t stream, unsigned *d_zero_word)
{
    const int nq = ref_end - ref_start;
    if (nq <= 0) return BGSA_HIP_OK;
    const int n_windows = pair_stream_windows(ref_len);
    const long long total = static_cast<long long>(nq) * (n_windows + 1);
    hipLaunchKernelGGL(pack_query_pairs_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0,
                       stream, d_content, static_cast<unsigned long long *>(d_streams), ref_len, ref_start,
                       nq, n_windows, d_zero_word);
    BGSA_HIP_TRY(hipGetLastError());
    return BGSA_HIP_OK;
}

// One thread per query writes its banded stream (a few hundred bytes; the layout is sequential).
__global__ __launch_bounds__(64) void pack_banded_kernel(const char *__restrict__ content,
                                                         unsigned char *__restrict__ streams, int len, int k, int phase, int cut,
                                                         int ref_start, int n_queries, int stride)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n_queries) return;
    banded_stream_layout(len, k, phase, cut, content + static_cast<size_t>(ref_start + q) * (len + 1),
                         streams + static_cast<size_t>(q) * stride);
}

int launch_pack_banded(const char *d_content, int len, int k, int phase, int cut, int ref_start, int ref_end, void *d_streams,
                       hipStream_t stream)
{
    const int nq = ref_end - ref_start;
    if (nq <= 0) return BGSA_HIP_OK;
    const int stride = banded_stream_layout(len, k, phase, cut, nullptr, nullptr);
    hipLaunchKernelGGL(pack_banded_kernel, dim3((nq + 63) / 64), dim3(64), 0, stream, d_content,
                       static_cast<unsigned char *>(d_streams), len, k, phase, cut, ref_start, nq, stride);
    BGSA_HIP_TRY(hipGetLastError());
    return BGSA_HIP_OK;
}

__global__ __launch_bounds__(64) void pack_blocked_kernel(const char *__restrict__ content,
                                                          unsigned char *__restrict__ streams, int len,
                                                          int ref_start, int n_queries, int stride)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n_queries) return;
    blocked_stream_layout(len, content + static_cast<size_t>(ref_start + q) * (len + 1),
                          streams + static_cast<size_t>(q) * stride);
}

int launch_pack_blocked(const char *d_content, int len, int ref_start, int ref_end, void *d_streams,
                        hipStream_t stream)
{
    const int nq = ref_end - ref_start;
    if (nq <= 0) return BGSA_HIP_OK;
    const int stride = blocked_stream_layout(len, nullptr, nullptr);
    hipLaunchKernelGGL(pack_blocked_kernel, dim3((nq + 63) / 64), dim3(64), 0, stream, d_content,
                       static_cast<unsigned char *>(d_streams), len, ref_start, nq, stride);
    BGSA_HIP_TRY(hipGetLastError());
    return BGSA_HIP_OK;
}

int launch_preprocess(int algo, const char *d_rows, int64_t avail_bytes, int len,
                      int64_t read_count, int word_num, int k, uint32_t *d_peq, hipStream_t stream)
{
    if (read_count == 0) return BGSA_HIP_OK;
    const unsigned blocks = static_cast<unsigned>((read_count + 255) / 256);
    if (algo == BGSA_ALGO_BANDED) {
        hipLaunchKernelGGL(preprocess_banded_kernel, dim3(blocks), dim3(256), 0, stream, d_rows, d_peq, len,
                           static_cast<long long>(read_count), word_num, k);
    } else if (64 * (len + 1) + 32 <= 16 * 1024 && (reinterpret_cast<uintptr_t>(d_rows) & 15) == 0 &&
               avail_bytes >= read_count * (len + 1)) {
        const long long n_groups = read_count / kLanes;
        const int slot = (64 * (len + 1) + 15 + 16) & ~15;
        hipLaunchKernelGGL(preprocess_global_lds_kernel, dim3(static_cast<unsigned>((n_groups + 3) / 4)), dim3(256),
                           static_cast<size_t>(slot) * kWavesPerBlock, stream, d_rows, d_peq, len, n_groups, word_num,
                           static_cast<long long>(avail_bytes));
    } else {
        hipLaunchKernelGGL(preprocess_global_kernel, dim3(blocks), dim3(256), 0, stream, d_rows,
                           d_peq, len, static_cast<long long>(read_count), word_num, 32);
    }
    BGSA_HIP_TRY(hipGetLastError());
    return BGSA_HIP_OK;
}

int launch_map_queries(char *d_content, int64_t bytes, hipStream_t stream)
{
    if (bytes == 0) return BGSA_HIP_OK;
    const unsigned blocks = static_cast<unsigned>((bytes + 255) / 256);
    hipLaunchKernelGGL(map_queries_kernel, dim3(blocks), dim3(256), 0, stream, d_content,
                       static_cast<long long>(bytes));
    BGSA_HIP_TRY(hipGetLastError());
    return BGSA_HIP_OK;
}

// ---- score scaling: the generator's `factor` (Main.java:213-267) ---------------------------------
// The reference multiplies the final score by a constant when the kernel computed a reduced problem:
// BitPAl scores with a common factor f run as (M/f, I/f, G/f) and the result is multiplied by f
// (genPackedScore, BitPAlGenerator.java:121-127); Myers with weights (0, 1, 1) (`-m 1`) reports
// +distance where `-m 0` reports -distance (genMyersScore, MyersGenerator.java:43-45).  Here that is
// one streaming pass over the int16 tile after the scoring kernel, only when a factor is in play.
__global__ __launch_bounds__(256) void scale_scores_kernel(int16_t *__restrict__ scores, long long n_vec8, int factor)
{
    const long long i = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
    if (i >= n_vec8) return;
    uint4 v = reinterpret_cast<uint4 *>(scores)[i];
    uint32_t *w = reinterpret_cast<uint32_t *>(&v);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int lo = static_cast<int16_t>(w[k] & 0xffffu) * factor;
        const int hi = static_cast<int16_t>(w[k] >> 16) * factor;
        w[k] = (static_cast<uint32_t>(lo) & 0xffffu) | (static_cast<uint32_t>(hi) << 16);
    }
    reinterpret_cast<uint4 *>(scores)[i] = v;
}

int launch_scale_scores(int16_t *d_scores, int64_t count, int factor, hipStream_t stream)
{
    if (count == 0 || factor == 1) return BGSA_HIP_OK;
    if (count % 8) {  // tiles are [queries][multiple of 64 subjects]
        set_error_text("scale_scores: element count must be a multiple of 8");
        return BGSA_HIP_EINVAL;
    }
    const long long n_vec8 = count / 8;
    hipLaunchKernelGGL(scale_scores_kernel, dim3(static_cast<unsigned>((n_vec8 + 255) / 256)), dim3(256), 0, stream,
                       d_scores, n_vec8, factor);
    BGSA_HIP_TRY(hipGetLastError());
    return BGSA_HIP_OK;
}

// ---- host routine behind hip_handle_reads -------------------------------------------------------

void host_handle_reads(int algo, const char *rows, int64_t avail, int len, uint32_t *result_reads,
                       int word_num, int64_t read_count, int k, int threads)
{
    const int64_t n_groups = read_count / kLanes;
    if (threads < 1) threads = 1;
    threads = static_cast<int>(std::min<int64_t>(threads, std::max<int64_t>(n_groups, 1)));
    auto work = [&](int t) {
        for (int64_t g = t; g < n_groups; g += threads) {
            for (int lane = 0; lane < kLanes; lane++) {
                const int64_t off = (g * kLanes + lane) * static_cast<int64_t>(len + 1);
                const char *row = rows + off;
                if (algo == BGSA_ALGO_BANDED) {
                    const size_t base = static_cast<size_t>(g) * kChars * word_num * kLanes + lane;
                    for (int p = 0; p < len; p++) {
                        const int i = p + k + 1;  // Mext bit index
                        if (i / 32 >= word_num) break;
                        result_reads[base + (static_cast<size_t>(map_char(static_cast<uint8_t>(row[p]))) * word_num + i / 32) * kLanes] |=
                            1u << (i % 32);
                    }
                } else {
                    const int bits = 32;
                    uint32_t *dst = result_reads + static_cast<size_t>(g) * kChars * word_num * kLanes + lane;
                    for (int p = 0; p < len; p++) {
                        const uint32_t c = map_char(static_cast<uint8_t>(row[p]));
                        dst[(c * word_num + p / bits) * kLanes] |= 1u << (p % bits);
                    }
                }
            }
        }
    };
    if (threads == 1) {
        work(0);
        return;
    }
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; t++) pool.emplace_back(work, t);
    for (auto &th : pool) th.join();
}

}  // namespace bgsa
