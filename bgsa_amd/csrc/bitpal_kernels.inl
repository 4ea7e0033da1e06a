// bitpal_kernels.inl — the BitPAl kernels and launchers of ONE score set.  Included by a generated
// translation unit (_gen/bitpal_set_<tag>.hip, gen_bitpal_sets.py) inside `namespace bgsa::<set>`,
// right after that set's generated row loops (bitpal_rows_gen.inc for the default 2/-3/-5, or
// _gen/bitpal_rows_<tag>.inc), which define kBitpalPlanes, kBitpalChains, kBitpalGap, kBitpalWeights,
// the width lists and bitpal_rows_asm<NW> / bitpal_block_rows_asm<NW>.
//
// Replaces the reference's align_avx hot loop (original/BGSA_AVX2/align_core.c:164-482 for the
// committed 2/-3/-5 instance; generator/.../BitPAlGenerator.java:151-534 for any other scores).
// Same decomposition as myers_global.hip: lane = subject, wave = group of 64, the wave keeps its
// Peq block in VGPRs and walks a tile of queries; the row loop is generated threaded-code asm
// (rows_ir.py: bitpal_body), every inter-word carry a VCC add-with-carry chain on full 32-bit words.
//
// Final score (align_core.c:433-471 generalised): gap*(qlen + slen) + sum over subject columns of
// u = dH - gap, i.e. one masked popcount per plane and word, weighted 2^plane (kBitpalWeights).

template <int NW>
__device__ __forceinline__ int bitpal_column_sum(const uint32_t *st, int first_word, int read_len)
{
    int sum = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) {
        const int rem = read_len - 32 * (first_word + w);
        const uint32_t m = rem >= 32 ? ~0u : (rem <= 0 ? 0u : ((1u << rem) - 1u));
#pragma unroll
        for (int i = 0; i < kBitpalPlanes; i++) sum += kBitpalWeights[i] * __popc(st[w * kBitpalPlanes + i] & m);
    }
    return sum;
}

// Semi-global (generator option -s, BitPAlGenerator.java:78-116): walk the last DP row one subject
// column at a time, run = S[m][j], best = the maximum so far.  Once per (query, subject) pair — about
// 3 % of the row loop's work at 150 bp.
template <int NW>
__device__ __forceinline__ void bitpal_last_row_max(const uint32_t *st, int first_word, int read_len, int &run, int &best)
{
#pragma unroll
    for (int w = 0; w < NW; w++) {
        int cols = read_len - 32 * (first_word + w);
        cols = cols > 32 ? 32 : cols;
        for (int j = 0; j < cols; j++) {   // cols is wave-uniform
            int u = 0;
#pragma unroll
            for (int i = 0; i < kBitpalPlanes; i++) u += kBitpalWeights[i] * static_cast<int>((st[w * kBitpalPlanes + i] >> j) & 1u);
            run += u + kBitpalGap;
            best = run > best ? run : best;
        }
    }
}

// Row 0 of the DP: dH = gap everywhere (global: u = 0) or dH = 0 (semi-global: u = -gap,
// writeBitInitStr, BitPAlGenerator.java:2201-2218).  The planes hold u itself, unsigned.
__device__ __forceinline__ uint32_t bitpal_init_plane(int plane, int semi)
{
    constexpr uint32_t stored = static_cast<uint32_t>(-kBitpalGap);
    return (semi && ((stored >> plane) & 1u)) ? ~0u : 0u;
}

template <int NW, bool SEMI, bool DYN = false>
__global__ __launch_bounds__(256) void bitpal_asm_kernel(
    const unsigned char *__restrict__ streams, const uint32_t *__restrict__ peq,
    int16_t *__restrict__ out, int ref_len, int read_len, long long ld, int n_groups, int word_num,
    int n_queries, int q_tile, int stream_stride_bytes, unsigned *__restrict__ fault_word,
    unsigned *__restrict__ task_counter)
{
    constexpr int semi = SEMI;
    const int lane = threadIdx.x & (kLanes - 1);
    // DYN: the waves of a persistent grid take (group, tile) tasks from a counter (bgsa_common.h "dynamic task handout")
    const unsigned n_tasks = static_cast<unsigned>(n_groups) * ((static_cast<unsigned>(n_queries) + q_tile - 1) / q_tile);   // < 2^32: the launcher checked
    unsigned task = 0, task_issued = 0;
    if constexpr (DYN) {
        task = first_wave_task();
        if (task >= n_tasks) return;
    }
    do {
        int group, tile;
        if constexpr (DYN) {
            group = static_cast<int>(task % static_cast<unsigned>(n_groups));
            tile = static_cast<int>(task / static_cast<unsigned>(n_groups));
        } else {
            group = __builtin_amdgcn_readfirstlane(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));
            tile = blockIdx.y;
            if (group >= n_groups) return;
        }

        uint32_t P[kChars][NW];
        // In the task loop of the DYN instantiation everything that does not depend on the task is loop-invariant, and the
        // compiler keeps it in VGPRs across the row loop: five "word exists" flags and the two per-lane base pointers — ten
        // registers, 103 instead of 93 at five words, the 150 bp kernel's fifth wave per SIMD.  Values laundered through an
        // empty asm are redefined per task as far as the compiler can tell, so they are recomputed (a handful of scalar
        // instructions per task) instead of kept.
        int wn = word_num;
        const uint32_t *peq_t = peq;
        int16_t *out_t = out;
        unsigned lane_t = static_cast<unsigned>(lane);
        if constexpr (DYN) asm volatile("" : "+s"(wn), "+s"(peq_t), "+s"(out_t), "+v"(lane_t));
        const uint32_t *g = peq_t + static_cast<size_t>(group) * kChars * wn * kLanes + lane_t;
#pragma unroll
        for (int c = 0; c < kChars; c++)
#pragma unroll
            for (int w = 0; w < NW; w++)
                P[c][w] = (w < wn) ? g[(c * wn + w) * kLanes] : 0u;

        const int q0 = tile * q_tile;
        const int q1 = (q0 + q_tile < n_queries) ? q0 + q_tile : n_queries;
        // (the store address is formed at the store from this wave-uniform base and the lane: a per-lane 64-bit pointer kept
        // across the row loop is two more registers)
        int16_t *dst = out_t + static_cast<size_t>(group) * kLanes;

        for (int q = q0; q < q1; q++) {
            if constexpr (DYN) {   // the next task, asked for under this one's last query: late enough that the tail of a
                if (q == q1 - 1) task_issued = issue_wave_task(task_counter);   // launch is handed out as waves free up, early enough that the round trip is hidden
            }
            uint32_t st[kBitpalPlanes * NW];
#pragma unroll
            for (int i = 0; i < kBitpalPlanes * NW; i++) st[i] = bitpal_init_plane(i % kBitpalPlanes, semi);  // (:167-171)
            const unsigned long long s =
                reinterpret_cast<unsigned long long>(streams) + static_cast<unsigned long long>(q) * stream_stride_bytes;
            note_stream_fault(fault_word, bitpal_rows_asm<NW>(st, P, uniform_u64(s),
                                                              __builtin_amdgcn_readfirstlane(stream_stride_bytes / 8 - 2)));
            int score;
            if (semi) {
                int run = kBitpalGap * ref_len;
                score = run;
                bitpal_last_row_max<NW>(st, 0, read_len, run, score);
            } else {
                score = kBitpalGap * (ref_len + read_len) + bitpal_column_sum<NW>(st, 0, read_len);
            }
            unsigned lane_s = static_cast<unsigned>(lane);
            if constexpr (DYN) asm volatile("" : "+v"(lane_s));      // (the lane's byte offset is formed here, not kept)
            dst[static_cast<size_t>(q) * ld + lane_s] = static_cast<int16_t>(score);
        }
        if constexpr (DYN) task = resolve_wave_task(task_issued);
    } while (DYN && task < n_tasks);
}

// Subjects wider than kBitpalMaxPlain words: column blocks of NW words, the carry chains of a row
// crossing block boundaries through per-wave carry words (same scheme as myers_blocked_kernel;
// rows_ir.py: make_blocked, CPU-simulated in tests/test_rows_ir.py).
template <int NW, bool SEMI>
__global__ __launch_bounds__(256) void bitpal_blocked_kernel(
    const unsigned char *__restrict__ streams, const uint32_t *__restrict__ peq, int16_t *__restrict__ out,
    uint32_t *__restrict__ carry_all, int ref_len, int read_len, long long ld, int n_groups, int word_num,
    int n_queries, int q_tile, int stream_stride_bytes, int n_blocks, unsigned long long *task_counter,
    unsigned *__restrict__ fault_word)
{
    constexpr int semi = SEMI;
    constexpr int NC = kBitpalChains;
    constexpr int NS = kBitpalPlanes * NW;
    const int lane = threadIdx.x & (kLanes - 1);
    const int wave = threadIdx.x >> 6;
    const int n_chunks = (ref_len + 31) / 32;
    uint32_t *carry = carry_all + (static_cast<size_t>(blockIdx.x) * kWavesPerBlock + wave) * n_chunks * NC * kLanes;
    const unsigned long long carry_base = uniform_u64(reinterpret_cast<unsigned long long>(carry));
    const int q_tiles = (n_queries + q_tile - 1) / q_tile;
    const long long n_tasks = static_cast<long long>((n_groups + kWavesPerBlock - 1) / kWavesPerBlock) * q_tiles;
    const int tail_rows = ref_len & 31;
    dephase_persistent_workgroup();

    for (long long task = next_blocked_task(task_counter); task < n_tasks; task = next_blocked_task(task_counter)) {
        const int group = __builtin_amdgcn_readfirstlane(static_cast<int>(task / q_tiles) * kWavesPerBlock + wave);
        const int tile = static_cast<int>(task % q_tiles);
        if (group >= n_groups) continue;  // wave-uniform; the wave still meets the others at the next task fetch
        const uint32_t *g = peq + static_cast<size_t>(group) * kChars * word_num * kLanes + lane;
        const int q0 = tile * q_tile;
        const int q1 = (q0 + q_tile < n_queries) ? q0 + q_tile : n_queries;
        for (int q = q0; q < q1; q++) {
            for (int i = 0; i < n_chunks * NC; i++) carry[i * kLanes + lane] = 0u;  // every chain starts at carry-in 0
            const unsigned long long s =
                reinterpret_cast<unsigned long long>(streams) + static_cast<unsigned long long>(q) * stream_stride_bytes;
            int score = semi ? kBitpalGap * ref_len : kBitpalGap * (ref_len + read_len);
            int run = score;  // semi-global: S[m][j] walking right along the last row; score = its maximum
            for (int blk = 0; blk < n_blocks; blk++) {
                uint32_t P[kChars][NW];
#pragma unroll
                for (int c = 0; c < kChars; c++)
#pragma unroll
                    for (int w = 0; w < NW; w++) {
                        const int gw = blk * NW + w;
                        const int gwc = gw < word_num ? gw : word_num - 1;   // branch-free: see myers_blocked_kernel
                        P[c][w] = g[(c * word_num + gwc) * kLanes] & (gw < word_num ? ~0u : 0u);
                    }
                uint32_t st[NS + 2 * NC];
#pragma unroll
                for (int i = 0; i < NS; i++) st[i] = bitpal_init_plane(i % kBitpalPlanes, semi);
#pragma unroll
                for (int i = 0; i < NC; i++) {
                    st[NS + i] = carry[i * kLanes + lane];  // chunk 0
                    st[NS + NC + i] = 0u;
                }
                uint32_t voff = static_cast<uint32_t>(lane * 4);
                note_stream_fault(fault_word, bitpal_block_rows_asm<NW>(st, P, voff, carry_base, uniform_u64(s),
                                                                        __builtin_amdgcn_readfirstlane(stream_stride_bytes / 8 - 2)));
#pragma unroll
                for (int i = 0; i < NC; i++) {
                    const uint32_t word = tail_rows ? (st[NS + NC + i] << (32 - tail_rows)) : st[NS + NC + i];
                    carry[((n_chunks - 1) * NC + i) * kLanes + lane] = word;
                }
                if (semi)
                    bitpal_last_row_max<NW>(st, blk * NW, read_len, run, score);
                else
                    score += bitpal_column_sum<NW>(st, blk * NW, read_len);
            }
            out[static_cast<size_t>(q) * ld + static_cast<size_t>(group) * kLanes + lane] = static_cast<int16_t>(score);
        }
    }
}

// The same for score sets with so many carry chains that a register pair per chain does not fit (kBitpalPackedBlocks;
// rows_ir.py: make_blocked_packed): the carries of one row are bits of kBitpalCarryWords words per direction, and the
// row loop exchanges them with the neighbouring blocks every row — carry buffer [row][word][64 lanes] per wave, one row
// more than the query has (the loop fetches one row ahead), zeroed per query: every chain of block 0 starts at carry-in 0.
template <int NW, bool SEMI>
__global__ __launch_bounds__(256) void bitpal_packed_blocked_kernel(
    const unsigned char *__restrict__ streams, const uint32_t *__restrict__ peq, int16_t *__restrict__ out,
    uint32_t *__restrict__ carry_all, int ref_len, int read_len, long long ld, int n_groups, int word_num,
    int n_queries, int q_tile, int stream_stride_bytes, int n_blocks, unsigned long long *task_counter,
    unsigned *__restrict__ fault_word)
{
    constexpr int semi = SEMI;
    constexpr int CW = kBitpalCarryWords;
    constexpr int NS = kBitpalPlanes * NW;
    const int lane = threadIdx.x & (kLanes - 1);
    const int wave = threadIdx.x >> 6;
    const int n_rows = ref_len + 1;
    uint32_t *carry = carry_all + (static_cast<size_t>(blockIdx.x) * kWavesPerBlock + wave) * n_rows * CW * kLanes;
    const unsigned long long carry_base = uniform_u64(reinterpret_cast<unsigned long long>(carry));
    const int q_tiles = (n_queries + q_tile - 1) / q_tile;
    const long long n_tasks = static_cast<long long>((n_groups + kWavesPerBlock - 1) / kWavesPerBlock) * q_tiles;
    dephase_persistent_workgroup();

    for (long long task = next_blocked_task(task_counter); task < n_tasks; task = next_blocked_task(task_counter)) {
        const int group = __builtin_amdgcn_readfirstlane(static_cast<int>(task / q_tiles) * kWavesPerBlock + wave);
        const int tile = static_cast<int>(task % q_tiles);
        if (group >= n_groups) continue;
        const uint32_t *g = peq + static_cast<size_t>(group) * kChars * word_num * kLanes + lane;
        const int q0 = tile * q_tile;
        const int q1 = (q0 + q_tile < n_queries) ? q0 + q_tile : n_queries;
        for (int q = q0; q < q1; q++) {
            for (int i = 0; i < n_rows * CW; i++) carry[i * kLanes + lane] = 0u;
            const unsigned long long s =
                reinterpret_cast<unsigned long long>(streams) + static_cast<unsigned long long>(q) * stream_stride_bytes;
            int score = semi ? kBitpalGap * ref_len : kBitpalGap * (ref_len + read_len);
            int run = score;
            for (int blk = 0; blk < n_blocks; blk++) {
                uint32_t P[kChars][NW];
#pragma unroll
                for (int c = 0; c < kChars; c++)
#pragma unroll
                    for (int w = 0; w < NW; w++) {
                        const int gw = blk * NW + w;
                        const int gwc = gw < word_num ? gw : word_num - 1;
                        P[c][w] = g[(c * word_num + gwc) * kLanes] & (gw < word_num ? ~0u : 0u);
                    }
                uint32_t st[NS + 2 * CW], next_in[CW];
#pragma unroll
                for (int i = 0; i < NS; i++) st[i] = bitpal_init_plane(i % kBitpalPlanes, semi);
#pragma unroll
                for (int j = 0; j < CW; j++) {
                    st[NS + j] = 0u;
                    st[NS + CW + j] = 0u;
                    next_in[j] = carry[j * kLanes + lane];   // row 0
                }
                uint32_t voff = static_cast<uint32_t>(lane * 4);
                note_stream_fault(fault_word, bitpal_packed_block_rows_asm<NW>(st, P, next_in, voff, carry_base, uniform_u64(s),
                                                                               __builtin_amdgcn_readfirstlane(stream_stride_bytes / 8 - 2)));
                if (semi)
                    bitpal_last_row_max<NW>(st, blk * NW, read_len, run, score);
                else
                    score += bitpal_column_sum<NW>(st, blk * NW, read_len);
            }
            out[static_cast<size_t>(q) * ld + static_cast<size_t>(group) * kLanes + lane] = static_cast<int16_t>(score);
        }
    }
}

namespace {

// Narrowest instantiated block width that covers word_num words with the fewest blocks.
inline int pick_block_nw(int word_num, int *n_blocks)
{
    const int blocks = (word_num + kBitpalBlockMax - 1) / kBitpalBlockMax;
    const int need = (word_num + blocks - 1) / blocks;
    const int nw = need < kBitpalBlockMin ? kBitpalBlockMin : need;
    *n_blocks = (word_num + nw - 1) / nw;
    return nw;
}

template <int NW>
int launch_blocked(const char *d_content, const uint32_t *d_peq, int16_t *d_results, int ref_len, int read_len,
                   int64_t read_count, int ref_start, int ref_end, int word_num, int n_blocks, void *d_workspace,
                   hipStream_t stream, int semi)
{
    const int nq = ref_end - ref_start;
    const int stride = blocked_stream_layout(ref_len, nullptr, nullptr);
    const size_t stream_bytes = (static_cast<size_t>(stride) * nq + 255) & ~static_cast<size_t>(255);
    if (int rc = launch_pack_blocked(d_content, ref_len, ref_start, ref_end, d_workspace, stream)) return rc;
    uint32_t *carry = reinterpret_cast<uint32_t *>(static_cast<unsigned char *>(d_workspace) + stream_bytes);
    unsigned long long *counter = reinterpret_cast<unsigned long long *>(
        reinterpret_cast<unsigned char *>(carry) + blocked_carry_bytes(ref_len, kBitpalChains));
    BGSA_HIP_TRY(hipMemsetAsync(counter, 0, sizeof(unsigned long long), stream));
    unsigned *fault = nullptr;
    if (int rc = stream_guard(d_workspace, stride, kCodeRefill, -1, stream, &fault)) return rc;
    auto kernel = semi ? bitpal_blocked_kernel<NW, true> : bitpal_blocked_kernel<NW, false>;
    hipLaunchKernelGGL(kernel, dim3(blocked_workgroups()), dim3(256), 0, stream,
                       static_cast<const unsigned char *>(d_workspace), d_peq, d_results, carry, ref_len, read_len,
                       static_cast<long long>(read_count), static_cast<int>(read_count / kLanes), word_num, nq,
                       (note_query_tile(blocked_q_tile(nq, read_count / kLanes)), blocked_q_tile(nq, read_count / kLanes)),
                       stride, n_blocks, counter, fault);
    BGSA_HIP_TRY(hipGetLastError());
    return BGSA_HIP_OK;
}

// Bytes of the packed-carry buffers (capi.hip sizes the workspace with the same formula: BitpalSet::carry_words).
inline size_t packed_carry_bytes(int ref_len)
{
    return static_cast<size_t>(ref_len + 1) * kBitpalCarryWords * kLanes * sizeof(uint32_t) * kWavesPerBlock * blocked_workgroups();
}

template <int NW>
int launch_packed_blocked(const char *d_content, const uint32_t *d_peq, int16_t *d_results, int ref_len, int read_len,
                          int64_t read_count, int ref_start, int ref_end, int word_num, int n_blocks, void *d_workspace,
                          hipStream_t stream, int semi)
{
    const int nq = ref_end - ref_start;
    const int stride = static_cast<int>(stream_stride(ref_len));
    // the same workspace split as launch_blocked (capi.hip sizes the stream part for the longer CARRY-token stream)
    const size_t stream_bytes = (static_cast<size_t>(blocked_stream_layout(ref_len, nullptr, nullptr)) * nq + 255) & ~static_cast<size_t>(255);
    if (int rc = launch_pack_queries(d_content, ref_len, ref_start, ref_end, d_workspace, stream)) return rc;
    uint32_t *carry = reinterpret_cast<uint32_t *>(static_cast<unsigned char *>(d_workspace) + stream_bytes);
    unsigned long long *counter = reinterpret_cast<unsigned long long *>(reinterpret_cast<unsigned char *>(carry) + packed_carry_bytes(ref_len));
    BGSA_HIP_TRY(hipMemsetAsync(counter, 0, sizeof(unsigned long long), stream));
    unsigned *fault = nullptr;
    if (int rc = stream_guard(d_workspace, stride, kCodeRefill, 7, stream, &fault)) return rc;
    auto kernel = semi ? bitpal_packed_blocked_kernel<NW, true> : bitpal_packed_blocked_kernel<NW, false>;
    hipLaunchKernelGGL(kernel, dim3(blocked_workgroups()), dim3(256), 0, stream,
                       static_cast<const unsigned char *>(d_workspace), d_peq, d_results, carry, ref_len, read_len,
                       static_cast<long long>(read_count), static_cast<int>(read_count / kLanes), word_num, nq,
                       (note_query_tile(blocked_q_tile(nq, read_count / kLanes)), blocked_q_tile(nq, read_count / kLanes)), stride, n_blocks, counter, fault);
    BGSA_HIP_TRY(hipGetLastError());
    return BGSA_HIP_OK;
}

// Column blocks in the set's carry form (a template so that only that form's row loops are instantiated).
template <bool PACKED>
int launch_blocks(const char *d_content, const uint32_t *d_peq, int16_t *d_results, int ref_len, int read_len,
                  int64_t read_count, int ref_start, int ref_end, int word_num, void *d_workspace, hipStream_t stream, int semi)
{
    int n_blocks = 0;
    switch (pick_block_nw(word_num, &n_blocks)) {
#define X(N)                                                                                                    \
    case N:                                                                                                     \
        if constexpr (PACKED)                                                                                   \
            return launch_packed_blocked<N>(d_content, d_peq, d_results, ref_len, read_len, read_count, ref_start, \
                                            ref_end, word_num, n_blocks, d_workspace, stream, semi);           \
        else                                                                                                    \
            return launch_blocked<N>(d_content, d_peq, d_results, ref_len, read_len, read_count, ref_start,       \
                                     ref_end, word_num, n_blocks, d_workspace, stream, semi);
        BGSA_BITPAL_BLOCK_WIDTHS(X)
#undef X
    default: break;
    }
    set_error_text("bitpal: no column-block kernel for this word count");
    return BGSA_HIP_EUNSUPPORTED;
}


template <int NW>
int launch_nw(const char *d_content, const uint32_t *d_peq, int16_t *d_results, int ref_len,
              int read_len, int64_t read_count, int ref_start, int ref_end, int word_num,
              void *d_workspace, hipStream_t stream, int semi)
{
    const int nq = ref_end - ref_start;
    const int64_t n_groups = read_count / kLanes;
    // a BitPAl row is 2-4x a Myers row.  Counter where the loop's registers cost no wave — asked of the runtime per score set,
    // width and mode, once (2/-3/-5: 5 words 98 -> 108 VGPRs, four waves per SIMD either way; 8 words 155 -> 169 would be
    // three -> two: 256 bp ran 10 % slower with it, profiles/r03_length_sweep.txt)
    static const bool counter_costs_no_wave[2] = {
        [] { int a = 0, b = 0;
             return hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, bitpal_asm_kernel<NW, false, false>, 256, 0) == hipSuccess &&
                    hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, bitpal_asm_kernel<NW, false, true>, 256, 0) == hipSuccess && b >= a && a > 0; }(),
        [] { int a = 0, b = 0;
             return hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, bitpal_asm_kernel<NW, true, false>, 256, 0) == hipSuccess &&
                    hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, bitpal_asm_kernel<NW, true, true>, 256, 0) == hipSuccess && b >= a && a > 0; }()};
    const TaskPlan plan = plan_tasks(nq, n_groups, static_cast<long long>(ref_len) * NW * 2, 16, NW <= 8 && counter_costs_no_wave[semi ? 1 : 0],
                                     query_tile_max());
    const int q_tile = plan.q_tile;
    note_query_tile(q_tile);
    dim3 grid(static_cast<unsigned>((n_groups + kWavesPerBlock - 1) / kWavesPerBlock),
              static_cast<unsigned>((nq + q_tile - 1) / q_tile));
    if (grid.y > 65535u && !plan.dynamic) {
        set_error_text("bitpal: too many query tiles for one launch");
        return BGSA_HIP_EUNSUPPORTED;
    }
    unsigned *counter = nullptr;   // zeroed by the packer
    if (plan.dynamic) {
        const long long blocks = static_cast<long long>(grid.x) * grid.y;
        counter = task_counter_in(d_workspace, stream_stride(ref_len) * static_cast<size_t>(nq));
        const int resident = semi ? persistent_blocks_for(bitpal_asm_kernel<NW, true, true>) : persistent_blocks_for(bitpal_asm_kernel<NW, false, true>);
        grid = dim3(static_cast<unsigned>(blocks < resident ? blocks : resident), 1u);
    }
    if (int rc = launch_pack_queries(d_content, ref_len, ref_start, ref_end, d_workspace, stream, counter)) return rc;
    unsigned *fault = nullptr;
    if (int rc = stream_guard(d_workspace, static_cast<int>(stream_stride(ref_len)), kCodeRefill, 7, stream, &fault)) return rc;
    auto kernel = counter ? (semi ? bitpal_asm_kernel<NW, true, true> : bitpal_asm_kernel<NW, false, true>)
                          : (semi ? bitpal_asm_kernel<NW, true, false> : bitpal_asm_kernel<NW, false, false>);
    hipLaunchKernelGGL(kernel, grid, dim3(256), 0, stream,
                       static_cast<const unsigned char *>(d_workspace), d_peq, d_results, ref_len,
                       read_len, static_cast<long long>(read_count), static_cast<int>(n_groups), word_num,
                       nq, q_tile, static_cast<int>(stream_stride(ref_len)), fault, counter);
    BGSA_HIP_TRY(hipGetLastError());
    return BGSA_HIP_OK;
}

}  // namespace

const char *set_kernel_name(int word_num)
{
    static thread_local char name[64];
    if (word_num > kBitpalMaxPlain) {
        int n_blocks = 0;
        snprintf(name, sizeof name, kBitpalPackedBlocks ? "bitpal_packed_blocked_kernel<%d>" : "bitpal_blocked_kernel<%d>",
                 pick_block_nw(word_num, &n_blocks));
        return name;
    }
    snprintf(name, sizeof name, "bitpal_asm_kernel<%d>", word_num);
    return name;
}

int set_launch(const char *d_content, const uint32_t *d_peq, int16_t *d_results, int ref_len,
               int read_len, int64_t read_count, int ref_start, int ref_end, int word_num,
               void *d_workspace, hipStream_t stream, int semi)
{
    if (word_num > kBitpalMaxPlain)
        return launch_blocks<kBitpalPackedBlocks>(d_content, d_peq, d_results, ref_len, read_len, read_count, ref_start, ref_end,
                                                  word_num, d_workspace, stream, semi);
    switch (word_num) {
#define X(N)                                                                                    \
    case N:                                                                                     \
        return launch_nw<N>(d_content, d_peq, d_results, ref_len, read_len, read_count,         \
                            ref_start, ref_end, word_num, d_workspace, stream, semi);
        BGSA_BITPAL_PLAIN_WIDTHS(X)
#undef X
    default:
        set_error_text("bitpal: no kernel for this word count");
        return BGSA_HIP_EUNSUPPORTED;
    }
}
