// myers_global.hip — Myers unit-cost global alignment, one subject per lane, gfx950.
//
// Replaces the reference's align_cpu / align_sse hot loop (original/BGSA_CPU/align_core.c:54-146,
// original/BGSA_SSE/align_core.c:55-150) and its OpenMP grid (cal_cpu.c:66-84).
//
// Mapping to the machine
//   * lane  = one subject; a wavefront = one "group" of HIP_V_NUM = 64 subjects, which is exactly
//     the reference's SIMD-lane layout [group][char][word][lane] widened from 4/8/16 to 64 lanes.
//   * The subject's match masks stay in VGPRs for the whole task; a task scores a tile of
//     queries against the group, so each Peq block is read from HBM once per tile.
//   * The query character is wave-uniform: it is fetched through the scalar cache and selects
//     one of five copies of the row body by a scalar jump, so `Eq = Peq[c][w]` costs no VALU
//     work (the reference pays a pointer add + a vector load per word, align_core.c:67,74).
//   * Words are full 32-bit (the reference keeps bit W-1 free as a software carry,
//     align_core.c:79-83,91-96): the addition and the 1-bit shift of HP are add-with-carry chains
//     through VCC; HN needs no shift — it is read off the addition's carries (rows_ir.py: myers_body).
//   * The score is not tracked per row (align_core.c:121-124); after the last row
//     D[m][n] = m + popcount(VP & mask) - popcount(VN & mask), two v_bcnt per word.
//
// Three kernels, chosen by launch_myers():
//   myers_global_asm_kernel<NW,1>   1..1024 bp   generated asm row loop, Peq planes resident, 8 VALU per (row, word); 30 and 32 words
//                                                (897..1024 bp) with the two carry chains in turns over blocks of 9 words
//   myers_global_planes_kernel<NW>  (A/B)        generated asm row loop on 3-bit character-code planes, 9 VALU per (row, word):
//                                                897..1024 bp until round 5, now under BGSA_MYERS_PEQ_MAX_WORDS
//   myers_global_kernel<NW,1>       compiler-scheduled C++ of the same recurrence: the A/B
//                                   reference for the asm (BGSA_MYERS_IMPL=c), 124 vs 216 TCUPS
//   myers_blocked_kernel<NW>        > 1024 bp    column blocks of the planes body, carries between
//                                                blocks through per-wave carry words
//
// Integer/bitwise only; no LDS, no MFMA.  The kernels are VALU-issue bound.
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "bgsa_common.h"

namespace bgsa {

// One DP row: in-place update of the vertical delta vectors for query character class `eq`.
// SEMI (the generator's `-s` for Myers, MyersGenerator.java:56-223): the row boundary feeds 0 instead
// of +1 into word 0 (D[0][i] = 0: the subject may start anywhere in the query) and the horizontal delta
// leaving the last subject column (bit `last_bit` of the last word's HP / HN) is handed back, so the
// caller can follow D[n][i] row by row.
template <int NW, int G, bool SEMI = false>
__device__ __forceinline__ void myers_row(uint32_t (&vp)[G * NW], uint32_t (&vn)[G * NW],
                                          const uint32_t (&eq)[G * NW], int last_word = 0, int last_bit = 0,
                                          int *delta = nullptr)
{
#pragma unroll
  for (int gi = 0; gi < G; gi++) {
    uint32_t hp_prev = 0, hn_prev = 0;
    unsigned carry = 0;
#pragma unroll
    for (int ww = 0; ww < NW; ww++) {
        const int w = gi * NW + ww;
        const uint32_t pv = vp[w], mv = vn[w], e = eq[w];
        const uint32_t pm = e | mv;
        // (pv & pm) == (pv & e) because pv & mv == 0 is an invariant of the recurrence.
        unsigned cout;
        const uint32_t sum = __builtin_addc(pv & e, pv, carry, &cout);
        carry = cout;
        const uint32_t d0 = (sum ^ pv) | pm;
        const uint32_t hp = ~(d0 | pv) | mv;
        const uint32_t hn = d0 & pv;
        // Shift one column along the subject; row boundary D[i][0]-D[i-1][0] = +1 enters word 0.
        const uint32_t hps = (ww == 0) ? ((hp << 1) | (SEMI ? 0u : 1u)) : ((hp << 1) | (hp_prev >> 31));
        const uint32_t hns = (ww == 0) ? (hn << 1) : ((hn << 1) | (hn_prev >> 31));
        hp_prev = hp;
        hn_prev = hn;
        vp[w] = ~(d0 | hps) | hns;
        vn[w] = d0 & hps;
        if (SEMI && ww == last_word)
            delta[gi] = static_cast<int>((hp >> last_bit) & 1u) - static_cast<int>((hn >> last_bit) & 1u);
    }
  }
}

// grid.x = ceil(n_groups / (4*G)), grid.y = number of query tiles; block = 4 waves, each wave
// owns G consecutive groups (G subjects per lane): the scalar work of a row (character fetch,
// 5-way branch) is shared by G x NW word updates.
template <int NW, int G, bool SEMI = false>
__global__ __launch_bounds__(256) void myers_global_kernel(
    const char *__restrict__ content, const uint32_t *__restrict__ peq, int16_t *__restrict__ out,
    int ref_len, int read_len, long long ld, int n_groups, int word_num, int ref_start,
    int ref_end, int q_tile)
{
    const int lane = threadIdx.x & (kLanes - 1);
    const int group0 = __builtin_amdgcn_readfirstlane((blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6)) * G);
    if (group0 >= n_groups) return;

    // Peq blocks of this wave's groups: [char][word][lane], coalesced 256-B rows.  A group past
    // the end of the bucket is computed on zero masks and not stored.
    uint32_t P[kChars][G * NW];
#pragma unroll
    for (int gi = 0; gi < G; gi++) {
        const bool live = group0 + gi < n_groups;
        const uint32_t *g = peq + static_cast<size_t>(group0 + gi) * kChars * word_num * kLanes + lane;
#pragma unroll
        for (int c = 0; c < kChars; c++)
#pragma unroll
            for (int w = 0; w < NW; w++)
                P[c][gi * NW + w] = (live && w < word_num) ? g[(c * word_num + w) * kLanes] : 0u;
    }

    const int q0 = ref_start + blockIdx.y * q_tile;
    const int q1 = (q0 + q_tile < ref_end) ? q0 + q_tile : ref_end;
    int16_t *dst = out + static_cast<size_t>(group0) * kLanes + lane;

    for (int q = q0; q < q1; q++) {
        uint32_t vp[G * NW], vn[G * NW];
#pragma unroll
        for (int w = 0; w < G * NW; w++) {
            vp[w] = ~0u;
            vn[w] = 0u;
        }
        UniformBytes qs(content + static_cast<size_t>(q) * (ref_len + 1));
        // semi-global: run[gi] = D[n][i], the subject against the query prefix ending at row i;
        // best = its minimum, starting from D[n][0] = n (genSemiGlobal: score = read_len, min_score = score)
        const int last_word = (read_len - 1) >> 5, last_bit = (read_len - 1) & 31;
        int run[G], best[G], delta[G];
#pragma unroll
        for (int gi = 0; gi < G; gi++) run[gi] = best[gi] = read_len;
        for (int r = 0; r < ref_len; r++) {
            if ((r & 3) == 0) qs.refill(r, ref_len - r);
            const uint32_t c = __builtin_amdgcn_readfirstlane(qs.next());
            switch (c) {
            case 0: myers_row<NW, G, SEMI>(vp, vn, P[0], last_word, last_bit, delta); break;
            case 1: myers_row<NW, G, SEMI>(vp, vn, P[1], last_word, last_bit, delta); break;
            case 2: myers_row<NW, G, SEMI>(vp, vn, P[2], last_word, last_bit, delta); break;
            case 3: myers_row<NW, G, SEMI>(vp, vn, P[3], last_word, last_bit, delta); break;
            default: myers_row<NW, G, SEMI>(vp, vn, P[4], last_word, last_bit, delta); break;
            }
            if (SEMI) {
#pragma unroll
                for (int gi = 0; gi < G; gi++) {
                    run[gi] += delta[gi];
                    best[gi] = run[gi] < best[gi] ? run[gi] : best[gi];
                }
            }
        }
        if (SEMI) {
#pragma unroll
            for (int gi = 0; gi < G; gi++)
                if (group0 + gi < n_groups)
                    dst[static_cast<size_t>(q - ref_start) * ld + gi * kLanes] = static_cast<int16_t>(-best[gi]);
            continue;
        }
        // D[m][n] = m + sum over the n subject columns of (VP - VN).
#pragma unroll
        for (int gi = 0; gi < G; gi++) {
            int score = ref_len;
#pragma unroll
            for (int w = 0; w < NW; w++) {
                const int rem = read_len - 32 * w;
                const uint32_t m = rem >= 32 ? ~0u : (rem <= 0 ? 0u : ((1u << rem) - 1u));
                score += __popc(vp[gi * NW + w] & m) - __popc(vn[gi * NW + w] & m);
            }
            if (group0 + gi < n_groups)
                dst[static_cast<size_t>(q - ref_start) * ld + gi * kLanes] = static_cast<int16_t>(-score);
        }
    }
}

// ---- generated row loop (gen_rows_asm.py) ----------------------------------------------------------
// default of myers_peq_max_words(): the widest subject with its five Peq planes resident.  28 until round 5; 30 and 32 words since then
// — their rows run the two carry chains in turns over blocks of nine words (rows_ir.myers_body(split = 9)), which holds 18 temporaries
// where the row-long phases hold 64: 255 VGPRs, eight instructions per word against nine on the code planes (config 5: 4,214 -> 3,980 ms,
// profiles/r05_balance_ab.txt, r05_split_k9.txt).  BGSA_MYERS_PEQ_MAX_WORDS=28 puts 29 .. 32 words back on the code planes (A/B).
constexpr int kPeqMaxWords = 32;
constexpr int kSemiPeqMaxWords = 32;  // widest semi-global kernel with resident Peq planes (myers_semi_rows_asm; 26..32 words: chains in turns, round 5)
constexpr int kPairMaxWords = 2;  // widths instantiated as myers_pair_rows_asm (gen_rows_asm.py: MYERS_PAIR_NW)
#include "myers_rows_gen.inc"

// Same task decomposition as above, but all rows of a query run inside one generated asm block:
// five in-place row bodies selected by a scalar jump per row, every VALU instruction full rate
// (the inter-word shifts are add-with-carry chains instead of v_alignbit_b32), query characters
// from the packed code stream.  This is the kernel the launcher picks whenever NW <= 8.
template <int NW, int G, bool DYN = false>
__global__ __launch_bounds__(256) void myers_global_asm_kernel(
    const unsigned char *__restrict__ streams, const uint32_t *__restrict__ peq,
    int16_t *__restrict__ out, int ref_len, int read_len, long long ld, int n_groups, int word_num,
    int n_queries, int q_tile, int stream_stride_bytes, unsigned *__restrict__ fault_word,
    unsigned *__restrict__ task_counter)
{
    const int lane = threadIdx.x & (kLanes - 1);
    // Static mapping (DYN = false): workgroup (x, y) = (four wave-groups, query tile).  Dynamic (bgsa_common.h "dynamic task
    // handout"; a separate instantiation, so that the static kernels keep their register counts: the loop costs 5-8 VGPRs,
    // which the widest widths do not have): a persistent grid whose waves take (wave-group, tile) tasks, tile-major like the
    // static order.
    const unsigned wave_groups = (static_cast<unsigned>(n_groups) + G - 1) / G;
    const unsigned n_tasks = wave_groups * ((static_cast<unsigned>(n_queries) + q_tile - 1) / q_tile);   // < 2^32: the launcher checked
    unsigned task = 0, task_issued = 0;
    if constexpr (DYN) {
        task = first_wave_task();
        if (task >= n_tasks) return;
    }
    do {
        int group0, tile;
        if constexpr (DYN) {
            group0 = static_cast<int>((task % wave_groups) * G);
            tile = static_cast<int>(task / wave_groups);
        } else {
            group0 = __builtin_amdgcn_readfirstlane((blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6)) * G);
            tile = blockIdx.y;
            if (group0 >= n_groups) return;
        }

        uint32_t P[kChars][G * NW];
#pragma unroll
        for (int gi = 0; gi < G; gi++) {
            const bool live = group0 + gi < n_groups;
            const uint32_t *g = peq + static_cast<size_t>(group0 + gi) * kChars * word_num * kLanes + lane;
#pragma unroll
            for (int c = 0; c < kChars; c++)
#pragma unroll
                for (int w = 0; w < NW; w++)
                    P[c][gi * NW + w] = (live && w < word_num) ? g[(c * word_num + w) * kLanes] : 0u;
        }

        const int q0 = tile * q_tile;
        const int q1 = (q0 + q_tile < n_queries) ? q0 + q_tile : n_queries;
        int16_t *dst = out + static_cast<size_t>(group0) * kLanes + lane;

        for (int q = q0; q < q1; q++) {
            if constexpr (DYN) {   // the next task, asked for under this one's last query: late enough that the tail of a
                if (q == q1 - 1) task_issued = issue_wave_task(task_counter);   // launch is handed out as waves free up, early enough that the round trip is hidden
            }
            uint32_t st[2 * G * NW];  // {VP, VN} per word
#pragma unroll
            for (int w = 0; w < G * NW; w++) {
                st[2 * w] = ~0u;
                st[2 * w + 1] = 0u;
            }
            const unsigned long long s =
                reinterpret_cast<unsigned long long>(streams) + static_cast<unsigned long long>(q) * stream_stride_bytes;
            int left;
            if constexpr (NW <= kPairMaxWords)  // short rows: two per stream token (launch_asm packs it so)
                left = myers_pair_rows_asm<NW, G>(st, P, uniform_u64(s), __builtin_amdgcn_readfirstlane(stream_stride_bytes / 8 - 2));
            else
                left = myers_rows_asm<NW, G>(st, P, uniform_u64(s), __builtin_amdgcn_readfirstlane(stream_stride_bytes / 8 - 2));
            note_stream_fault(fault_word, left);
#pragma unroll
            for (int gi = 0; gi < G; gi++) {
                int score = ref_len;  // D[m][n] = m + sum over the n subject columns of (VP - VN)
#pragma unroll
                for (int w = 0; w < NW; w++) {
                    const int rem = read_len - 32 * w;
                    const uint32_t m = rem >= 32 ? ~0u : (rem <= 0 ? 0u : ((1u << rem) - 1u));
                    score += __popc(st[2 * (gi * NW + w)] & m) - __popc(st[2 * (gi * NW + w) + 1] & m);
                }
                if (group0 + gi < n_groups)
                    dst[static_cast<size_t>(q) * ld + gi * kLanes] = static_cast<int16_t>(-score);
            }
        }
        if constexpr (DYN) task = resolve_wave_task(task_issued);
    } while (DYN && task < n_tasks);
}

// ---- semi-global (the generator's -m 0 -s, MyersGenerator.java:56-223) -------------------------------------
// The subject end to end inside the query: D[i][0] = 0 for every query row, result = -min over rows of D[i][n].
// What the semi-global kernels do to a subject once per task (rows_ir.py: semi_align): its n columns are moved
// to the top of the `total_words` words they occupy, so that column n is bit 31 of the last word and the
// HP / HN shift chains drop D[i][n] - D[i-1][n] out as their final carries; the s = 32 * total_words - n unused
// low columns match every character and start at VP = 0, which keeps them at D = 0 — the row edge of the mode,
// delivered to the first real column.
// Source word `i` of class plane `row` (words row[0 .. word_num), stride kLanes), zero outside the subject.
// Unconditional load, index clamped and result masked: a guarded load becomes a branch per word and the
// loads of a block then complete one after the other instead of together.
__device__ __forceinline__ uint32_t semi_source_word(const uint32_t *row, int word_num, int i)
{
    const int ic = i < 0 ? 0 : (i >= word_num ? word_num - 1 : i);
    return row[ic * kLanes] & ((i >= 0 && i < word_num) ? ~0u : 0u);
}
// Aligned word from its two source words, s = 32 q + r: hi = source word aw - q, lo = source word aw - q - 1.
__device__ __forceinline__ uint32_t semi_funnel(uint32_t hi, uint32_t lo, int r)
{
    return r ? __builtin_amdgcn_alignbit(hi, lo, 32 - r) : hi;
}
// the unused low columns of aligned word aw (wave-uniform)
__device__ __forceinline__ uint32_t semi_dummy_mask(int aw, int s)
{
    const int d = s - 32 * aw;
    return d >= 32 ? ~0u : (d <= 0 ? 0u : ((1u << d) - 1u));
}

// Subjects up to 1024 bp (resident Peq planes at every width; 26..32 words with the two carry chains in turns): generated asm row loop of
// myers_semi_body, 8 VALU per word + 3 per row.
template <int NW>
__global__ __launch_bounds__(256) void myers_semi_asm_kernel(
    const unsigned char *__restrict__ streams, const uint32_t *__restrict__ peq,
    int16_t *__restrict__ out, int ref_len, int read_len, long long ld, int n_groups, int word_num,
    int n_queries, int q_tile, int stream_stride_bytes, unsigned *__restrict__ fault_word)
{
    const int lane = threadIdx.x & (kLanes - 1);
    const int group = __builtin_amdgcn_readfirstlane(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));
    if (group >= n_groups) return;
    const int s_cols = 32 * NW - read_len, sq = s_cols >> 5, sr = s_cols & 31;

    uint32_t P[kChars][NW];
    const uint32_t *g = peq + static_cast<size_t>(group) * kChars * word_num * kLanes + lane;
#pragma unroll
    for (int c = 0; c < kChars; c++) {
        const uint32_t *row = g + static_cast<size_t>(c) * word_num * kLanes;
        uint32_t src[NW + 1];
#pragma unroll
        for (int w = 0; w <= NW; w++) src[w] = semi_source_word(row, word_num, w - sq - 1);
#pragma unroll
        for (int w = 0; w < NW; w++) P[c][w] = semi_funnel(src[w + 1], src[w], sr) | semi_dummy_mask(w, s_cols);
    }

    const int q0 = blockIdx.y * q_tile;
    const int q1 = (q0 + q_tile < n_queries) ? q0 + q_tile : n_queries;
    int16_t *dst = out + static_cast<size_t>(group) * kLanes + lane;

    for (int q = q0; q < q1; q++) {
        uint32_t st[2 * NW + 2];
#pragma unroll
        for (int w = 0; w < NW; w++) {
            st[2 * w] = ~semi_dummy_mask(w, s_cols);
            st[2 * w + 1] = 0u;
        }
        st[2 * NW] = st[2 * NW + 1] = static_cast<uint32_t>(read_len);   // D[0][n] = n (genSemiGlobal: min_score = score = read_len)
        const unsigned long long s =
            reinterpret_cast<unsigned long long>(streams) + static_cast<unsigned long long>(q) * stream_stride_bytes;
        note_stream_fault(fault_word, myers_semi_rows_asm<NW>(st, P, uniform_u64(s),
                                                              __builtin_amdgcn_readfirstlane(stream_stride_bytes / 8 - 2)));
        dst[static_cast<size_t>(q) * ld] = static_cast<int16_t>(-static_cast<int>(st[2 * NW + 1]));
    }
}

// Long subjects (769..1024 bp; 257..1024 before the Peq-resident kernels were widened): the wave turns its five Peq planes into the subject's 3-bit
// character-code planes once per task (B0 = C|T, B1 = G|T, B2 = N) and the row body rebuilds the
// match mask of its class with one v_bitop3 per word (rows_ir.py:myers_planes_body): 9 VALU per
// word, 7*NW+1 registers, two waves per SIMD at NW = 32.
#ifdef BGSA_PLANES_WAVES_PER_EU   // measurement builds (scripts/build_variant.sh ... EXTRA=-DBGSA_PLANES_WAVES_PER_EU=3): ask for an occupancy
#define BGSA_PLANES_OCCUPANCY __attribute__((amdgpu_waves_per_eu(BGSA_PLANES_WAVES_PER_EU, BGSA_PLANES_WAVES_PER_EU)))
#else
#define BGSA_PLANES_OCCUPANCY
#endif
template <int NW, bool DYN = false>
__global__ __launch_bounds__(256) BGSA_PLANES_OCCUPANCY void myers_global_planes_kernel(
    const unsigned char *__restrict__ streams, const uint32_t *__restrict__ peq,
    int16_t *__restrict__ out, int ref_len, int read_len, long long ld, int n_groups, int word_num,
    int n_queries, int q_tile, int stream_stride_bytes, unsigned *__restrict__ fault_word,
    unsigned *__restrict__ task_counter)
{
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

        uint32_t Bp[3 * NW];
        const uint32_t *g = peq + static_cast<size_t>(group) * kChars * word_num * kLanes + lane;
#pragma unroll
        for (int w = 0; w < NW; w++) {
            uint32_t p[kChars];
#pragma unroll
            for (int c = 0; c < kChars; c++) p[c] = (w < word_num) ? g[(c * word_num + w) * kLanes] : 0u;
            Bp[3 * w + 0] = p[1] | p[3];
            Bp[3 * w + 1] = p[2] | p[3];
            Bp[3 * w + 2] = p[4];
        }

        const int q0 = tile * q_tile;
        const int q1 = (q0 + q_tile < n_queries) ? q0 + q_tile : n_queries;
        int16_t *dst = out + static_cast<size_t>(group) * kLanes + lane;

        for (int q = q0; q < q1; q++) {
            if constexpr (DYN) {   // the next task, asked for under this one's last query: late enough that the tail of a
                if (q == q1 - 1) task_issued = issue_wave_task(task_counter);   // launch is handed out as waves free up, early enough that the round trip is hidden
            }
            uint32_t st[2 * NW];
#pragma unroll
            for (int w = 0; w < NW; w++) {
                st[2 * w] = ~0u;
                st[2 * w + 1] = 0u;
            }
            const unsigned long long s =
                reinterpret_cast<unsigned long long>(streams) + static_cast<unsigned long long>(q) * stream_stride_bytes;
            note_stream_fault(fault_word, myers_planes_rows_asm<NW>(st, Bp, uniform_u64(s),
                                                                    __builtin_amdgcn_readfirstlane(stream_stride_bytes / 8 - 2)));
            int score = ref_len;
#pragma unroll
            for (int w = 0; w < NW; w++) {
                const int rem = read_len - 32 * w;
                const uint32_t m = rem >= 32 ? ~0u : (rem <= 0 ? 0u : ((1u << rem) - 1u));
                score += __popc(st[2 * w] & m) - __popc(st[2 * w + 1] & m);
            }
            dst[static_cast<size_t>(q) * ld] = static_cast<int16_t>(-score);
        }
        if constexpr (DYN) task = resolve_wave_task(task_issued);
    } while (DYN && task < n_tasks);
}

// Semi-global on the code planes (801..1024 bp until round 5; since then those widths run myers_semi_asm_kernel with the chains in turns and
// this kernel is what BGSA_MYERS_PEQ_MAX_WORDS selects): the code planes right-aligned like the Peq planes of myers_semi_asm_kernel;
// the unused low columns get code 7 (all three planes set), which MATCH3's truth tables treat as "matches every
// class" (rows_ir.py: myers_semi_planes_body) — 9 VALU per word + 3 per row, two waves per SIMD at NW = 32.
template <int NW>
__global__ __launch_bounds__(256) void myers_semi_planes_kernel(
    const unsigned char *__restrict__ streams, const uint32_t *__restrict__ peq,
    int16_t *__restrict__ out, int ref_len, int read_len, long long ld, int n_groups, int word_num,
    int n_queries, int q_tile, int stream_stride_bytes, unsigned *__restrict__ fault_word)
{
    const int lane = threadIdx.x & (kLanes - 1);
    const int group = __builtin_amdgcn_readfirstlane(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));
    if (group >= n_groups) return;
    const int s_cols = 32 * NW - read_len, sq = s_cols >> 5, sr = s_cols & 31;

    uint32_t Bp[3 * NW];
    const uint32_t *g = peq + static_cast<size_t>(group) * kChars * word_num * kLanes + lane;
    {
        const uint32_t *rc = g + static_cast<size_t>(1) * word_num * kLanes, *rg = g + static_cast<size_t>(2) * word_num * kLanes;
        const uint32_t *rt = g + static_cast<size_t>(3) * word_num * kLanes, *rn = g + static_cast<size_t>(4) * word_num * kLanes;
        uint32_t lo[3];   // source word (w - sq - 1) of the three code planes
        {
            const uint32_t t = semi_source_word(rt, word_num, -sq - 1);
            lo[0] = semi_source_word(rc, word_num, -sq - 1) | t;
            lo[1] = semi_source_word(rg, word_num, -sq - 1) | t;
            lo[2] = semi_source_word(rn, word_num, -sq - 1);
        }
#pragma unroll
        for (int w = 0; w < NW; w++) {
            const uint32_t t = semi_source_word(rt, word_num, w - sq);
            const uint32_t hi[3] = {semi_source_word(rc, word_num, w - sq) | t, semi_source_word(rg, word_num, w - sq) | t,
                                    semi_source_word(rn, word_num, w - sq)};
            const uint32_t dummy = semi_dummy_mask(w, s_cols);
#pragma unroll
            for (int i = 0; i < 3; i++) {
                Bp[3 * w + i] = semi_funnel(hi[i], lo[i], sr) | dummy;
                lo[i] = hi[i];
            }
        }
    }

    const int q0 = blockIdx.y * q_tile;
    const int q1 = (q0 + q_tile < n_queries) ? q0 + q_tile : n_queries;
    int16_t *dst = out + static_cast<size_t>(group) * kLanes + lane;

    for (int q = q0; q < q1; q++) {
        uint32_t st[2 * NW + 2];
#pragma unroll
        for (int w = 0; w < NW; w++) {
            st[2 * w] = ~semi_dummy_mask(w, s_cols);
            st[2 * w + 1] = 0u;
        }
        st[2 * NW] = st[2 * NW + 1] = static_cast<uint32_t>(read_len);
        const unsigned long long s =
            reinterpret_cast<unsigned long long>(streams) + static_cast<unsigned long long>(q) * stream_stride_bytes;
        note_stream_fault(fault_word, myers_semi_planes_rows_asm<NW>(st, Bp, uniform_u64(s),
                                                                     __builtin_amdgcn_readfirstlane(stream_stride_bytes / 8 - 2)));
        dst[static_cast<size_t>(q) * ld] = static_cast<int16_t>(-static_cast<int>(st[2 * NW + 1]));
    }
}

// Subjects longer than 1024 bp: column blocks of NW words.  For each query the wave runs the
// generated row loop once per block; the three carry chains of row r cross the block boundary
// through its carry buffer ([32-row chunk][add, HP, HN][lane] words in the workspace, first row in
// bit 31 — rows_ir.py: myers_block_body).  A fixed number of workgroups loops over the tasks so
// that the buffer count does not grow with the problem.
// SEMI (PEQ blocks only): the subject right-aligned over the n_blocks x NW words (semi_aligned_word above),
// HP carry-in 0, and D[i][n] followed through the last block's HP / HN carry-out words after the row loop.
template <int NW, bool PEQ = false, bool SEMI = false>
__global__ __launch_bounds__(256) void myers_blocked_kernel(
    const unsigned char *__restrict__ streams, const uint32_t *__restrict__ peq, int16_t *__restrict__ out,
    uint32_t *__restrict__ carry_all, int ref_len, int read_len, long long ld, int n_groups, int word_num,
    int n_queries, int q_tile, int stream_stride_bytes, int n_blocks, unsigned long long *task_counter,
    unsigned *__restrict__ fault_word)
{
    const int lane = threadIdx.x & (kLanes - 1);
    const int wave = threadIdx.x >> 6;
    const int n_chunks = (ref_len + 31) / 32;
    uint32_t *carry = carry_all + (static_cast<size_t>(blockIdx.x) * kWavesPerBlock + wave) * n_chunks * 3 * kLanes;
    const unsigned long long carry_base = uniform_u64(reinterpret_cast<unsigned long long>(carry));
    const int q_tiles = (n_queries + q_tile - 1) / q_tile;
    const long long n_tasks = static_cast<long long>((n_groups + kWavesPerBlock - 1) / kWavesPerBlock) * q_tiles;
    const int tail_rows = ref_len & 31;
    static_assert(!SEMI || PEQ, "semi-global column blocks use the Peq-resident body");
    const int s_cols = 32 * NW * n_blocks - read_len, sq = s_cols >> 5, sr = s_cols & 31;   // SEMI: unused low columns
    dephase_persistent_workgroup();

    for (long long task = next_blocked_task(task_counter); task < n_tasks; task = next_blocked_task(task_counter)) {
        const int group = __builtin_amdgcn_readfirstlane(static_cast<int>(task / q_tiles) * kWavesPerBlock + wave);
        const int tile = static_cast<int>(task % q_tiles);
        if (group >= n_groups) continue;  // wave-uniform; the wave still meets the others at the next task fetch
        const uint32_t *g = peq + static_cast<size_t>(group) * kChars * word_num * kLanes + lane;
        const int q0 = tile * q_tile;
        const int q1 = (q0 + q_tile < n_queries) ? q0 + q_tile : n_queries;
        for (int q = q0; q < q1; q++) {
            // carry-in of block 0: addition 0, HP 1 (the row edge D[i][0] - D[i-1][0] = +1; semi-global: 0), HN 0
            for (int c = 0; c < n_chunks; c++) {
                carry[(c * 3 + 0) * kLanes + lane] = 0u;
                carry[(c * 3 + 1) * kLanes + lane] = SEMI ? 0u : ~0u;
                carry[(c * 3 + 2) * kLanes + lane] = 0u;
            }
            const unsigned long long s =
                reinterpret_cast<unsigned long long>(streams) + static_cast<unsigned long long>(q) * stream_stride_bytes;
            int score = ref_len;
            for (int blk = 0; blk < n_blocks; blk++) {
                uint32_t Bp[PEQ ? 1 : 3 * NW];       // 3-bit character-code planes of the block, or
                uint32_t Pq[kChars][PEQ ? NW : 1];   // its five Peq planes (PEQ: 8 VALU per word, narrower blocks)
                if constexpr (SEMI) {
#pragma unroll
                    for (int c = 0; c < kChars; c++) {
                        const uint32_t *row = g + static_cast<size_t>(c) * word_num * kLanes;
                        uint32_t src[NW + 1];
#pragma unroll
                        for (int w = 0; w <= NW; w++) src[w] = semi_source_word(row, word_num, blk * NW + w - sq - 1);
#pragma unroll
                        for (int w = 0; w < NW; w++)
                            Pq[c][w] = semi_funnel(src[w + 1], src[w], sr) | semi_dummy_mask(blk * NW + w, s_cols);
                    }
                } else {
#pragma unroll
                    for (int w = 0; w < NW; w++) {
                        const int gw = blk * NW + w;
                        uint32_t p[kChars];
                        // unconditional loads (index clamped, result masked): a guarded load becomes a branch per
                        // word, and this runs per query and block, not once per task as in the plain kernels
                        const int gwc = gw < word_num ? gw : word_num - 1;
                        const uint32_t keep = gw < word_num ? ~0u : 0u;
#pragma unroll
                        for (int c = 0; c < kChars; c++) p[c] = g[(c * word_num + gwc) * kLanes] & keep;
                        if constexpr (PEQ) {
#pragma unroll
                            for (int c = 0; c < kChars; c++) Pq[c][w] = p[c];
                        } else {
                            Bp[3 * w + 0] = p[1] | p[3];
                            Bp[3 * w + 1] = p[2] | p[3];
                            Bp[3 * w + 2] = p[4];
                        }
                    }
                }
                uint32_t st[2 * NW + 6];
#pragma unroll
                for (int w = 0; w < NW; w++) {
                    st[2 * w] = SEMI ? ~semi_dummy_mask(blk * NW + w, s_cols) : ~0u;
                    st[2 * w + 1] = 0u;
                }
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    st[2 * NW + i] = carry[i * kLanes + lane];  // chunk 0
                    st[2 * NW + 3 + i] = 0u;
                }
                uint32_t voff = static_cast<uint32_t>(lane * 4);
                int left;
                if constexpr (PEQ)
                    left = myers_peq_block_rows_asm<NW>(st, Pq, voff, carry_base, uniform_u64(s),
                                                        __builtin_amdgcn_readfirstlane(stream_stride_bytes / 8 - 2));
                else
                    left = myers_block_rows_asm<NW>(st, Bp, voff, carry_base, uniform_u64(s),
                                                    __builtin_amdgcn_readfirstlane(stream_stride_bytes / 8 - 2));
                note_stream_fault(fault_word, left);
                // carry-out words of the last (possibly partial) chunk, first row left-aligned
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    const uint32_t word = tail_rows ? (st[2 * NW + 3 + i] << (32 - tail_rows)) : st[2 * NW + 3 + i];
                    carry[((n_chunks - 1) * 3 + i) * kLanes + lane] = word;
                }
#pragma unroll
                for (int w = 0; w < NW; w++) {
                    const int rem = read_len - 32 * (blk * NW + w);
                    const uint32_t m = rem >= 32 ? ~0u : (rem <= 0 ? 0u : ((1u << rem) - 1u));
                    score += __popc(st[2 * w] & m) - __popc(st[2 * w + 1] & m);
                }
            }
            if constexpr (SEMI) {
                // The carry buffer now holds what left the LAST block: per 32-row chunk the HP and HN bits of
                // column n, first row in bit 31 — D[i][n] - D[i-1][n] row by row.  D[0][n] = n.
                int run = read_len, best = read_len;
                for (int c = 0; c < n_chunks; c++) {
                    const uint32_t hpw = carry[(c * 3 + 1) * kLanes + lane], hnw = carry[(c * 3 + kMyersBlockHnPair) * kLanes + lane];
                    const int rows = ref_len - 32 * c < 32 ? ref_len - 32 * c : 32;
                    for (int b2 = 0; b2 < rows; b2++) {
                        run += static_cast<int>((hpw >> (31 - b2)) & 1u) - static_cast<int>((hnw >> (31 - b2)) & 1u);
                        best = run < best ? run : best;
                    }
                }
                score = best;
            }
            out[static_cast<size_t>(q) * ld + static_cast<size_t>(group) * kLanes + lane] = static_cast<int16_t>(-score);
        }
    }
}

namespace {

// Register-resident word counts that are instantiated; a subject uses the smallest one that
// holds it (extra words are all-zero Peq and masked out of the score).
constexpr int kMyersNW[] = {1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16, 20, 24, 28, 32};

int pick_nw(int word_num)
{
    for (int nw : kMyersNW)
        if (nw >= word_num) return nw;
    return -1;
}

// Queries per task.  Small enough that the grid has >> 256 CUs x 8 waves of tasks even for a
// few thousand subjects, large enough that the 5*NW Peq loads are noise next to
// q_tile * ref_len * 10 * NW VALU ops.
int pick_q_tile(int nq, int64_t n_wave_tasks, int ref_len, int words)
{
    return pick_query_tile(nq, n_wave_tasks, static_cast<long long>(ref_len) * words, 32);
}

// 0 = generated-asm row loop (default), 1 = compiler-scheduled C++ kernel (A/B and NW > 8).
int myers_impl()
{
    static const int impl = [] {
        const char *e = getenv("BGSA_MYERS_IMPL");
        return (e && e[0] == 'c') ? 1 : 0;
    }();
    return impl;
}

// Subject groups per wave of the two-rows-per-token kernels (<= 64 bp): 2 by default, BGSA_MYERS_PAIR_GROUPS=1 for the A/B.
static int pair_groups()
{
    static const int g = [] {
        const char *e = getenv("BGSA_MYERS_PAIR_GROUPS");
        return (e && e[0] == '1') ? 1 : 2;
    }();
    return g;
}

// Measurement knob: bytes of unused dynamic LDS per workgroup of the counter kernels — caps the waves per SIMD (160 KB of LDS
// per CU: 40960 -> four workgroups = four waves per SIMD).  Round 4 asked whether the 150 bp kernel, which issues at 98 % of
// the sustained clock with eight waves per SIMD, holds a higher clock with fewer (LABNOTES 9.5).
static unsigned myers_lds_pad()
{
    static const unsigned v = [] { const char *e = getenv("BGSA_MYERS_LDS_PAD"); return e ? static_cast<unsigned>(atoi(e)) : 0u; }();
    return v;
}

// Most queries per task of the 30- and 32-word kernels.  A task loads the group's 160 Peq words once and walks its queries, so the
// tile is what the launch's HBM traffic hangs on: 8 (the code-plane kernels' choice: their tasks are long) re-read the 660 B per
// subject 125 times per 1,000 queries — 83 GB per config-5 pass, 9.7 x SURVEY's algorithmic bytes —, 32 a quarter of that.  Neither costs
// time at < 1 % of HBM peak; BGSA_MYERS_LONG_TILE (8 | 16 | 32) is the measurement knob.
static int long_query_tile()
{
    static const int v = [] {
        const char *e = getenv("BGSA_MYERS_LONG_TILE");
        const int t = e ? atoi(e) : 32;
        return (t == 8 || t == 16 || t == 32) ? t : 32;
    }();
    return v;
}

template <int NW, int G>
int launch_asm(const char *d_content, const uint32_t *d_peq, int16_t *d_results, int ref_len,
               int read_len, int64_t read_count, int ref_start, int ref_end, int word_num,
               void *d_workspace, hipStream_t stream)
{
    const int nq = ref_end - ref_start;
    const int64_t n_groups = read_count / kLanes;
    // the widths with registers to spare have a counter instantiation, and so have the split-chain widths (30, 32 words: two
    // waves per SIMD with or without the task loop's registers)
    constexpr bool kCounter = NW <= 8 || NW >= 30;
    const TaskPlan plan = plan_tasks(nq, (n_groups + G - 1) / G, static_cast<long long>(ref_len) * NW * G, NW >= 30 ? 8 : 32, kCounter,
                                     NW >= 30 ? long_query_tile() : query_tile_max());
    const int q_tile = plan.q_tile;
    note_query_tile(q_tile);
    dim3 grid(static_cast<unsigned>((n_groups + kWavesPerBlock * G - 1) / (kWavesPerBlock * G)),
              static_cast<unsigned>((nq + q_tile - 1) / q_tile));
    if (grid.y > 65535u && !plan.dynamic) {
        set_error_text("myers: too many query tiles for one launch");
        return BGSA_HIP_EUNSUPPORTED;
    }
    constexpr bool kPairs = NW <= kPairMaxWords;
    const int stride = static_cast<int>(kPairs ? pair_stream_stride(ref_len) : stream_stride(ref_len));
    unsigned *counter = nullptr;   // the task counter behind the streams; the packer zeroes it
    if (plan.dynamic) {
        const long long blocks = static_cast<long long>(grid.x) * grid.y;
        counter = task_counter_in(d_workspace, static_cast<size_t>(stride) * nq);
        int resident = persistent_blocks();
        if constexpr (kCounter) resident = persistent_blocks_for(myers_global_asm_kernel<NW, G, true>, myers_lds_pad());
        grid = dim3(static_cast<unsigned>(blocks < resident ? blocks : resident), 1u);
    }
    if (int rc = kPairs ? launch_pack_query_pairs(d_content, ref_len, ref_start, ref_end, d_workspace, stream, counter)
                        : launch_pack_queries(d_content, ref_len, ref_start, ref_end, d_workspace, stream, counter))
        return rc;
    unsigned *fault = nullptr;
    if (int rc = stream_guard(d_workspace, stride, kPairs ? kPairRefill : kCodeRefill, kPairs ? -1 : 7, stream, &fault)) return rc;
    if constexpr (kCounter) {
        if (counter) {
            hipLaunchKernelGGL((myers_global_asm_kernel<NW, G, true>), grid, dim3(256), myers_lds_pad(), stream,
                               static_cast<const unsigned char *>(d_workspace), d_peq, d_results, ref_len,
                               read_len, static_cast<long long>(read_count), static_cast<int>(n_groups), word_num,
                               nq, q_tile, stride, fault, counter);
            BGSA_HIP_TRY(hipGetLastError());
            return BGSA_HIP_OK;
        }
    }
    hipLaunchKernelGGL((myers_global_asm_kernel<NW, G, false>), grid, dim3(256), 0, stream,
                       static_cast<const unsigned char *>(d_workspace), d_peq, d_results, ref_len,
                       read_len, static_cast<long long>(read_count), static_cast<int>(n_groups), word_num,
                       nq, q_tile, stride, fault, counter);
    BGSA_HIP_TRY(hipGetLastError());
    return BGSA_HIP_OK;
}

template <int NW>
int launch_semi_asm(const char *d_content, const uint32_t *d_peq, int16_t *d_results, int ref_len,
                    int read_len, int64_t read_count, int ref_start, int ref_end, int word_num,
                    void *d_workspace, hipStream_t stream)
{
    const int nq = ref_end - ref_start;
    const int64_t n_groups = read_count / kLanes;
    const int q_tile = pick_q_tile(nq, n_groups, ref_len, NW);
    note_query_tile(q_tile);
    dim3 grid(static_cast<unsigned>((n_groups + kWavesPerBlock - 1) / kWavesPerBlock),
              static_cast<unsigned>((nq + q_tile - 1) / q_tile));
    if (grid.y > 65535u) {
        set_error_text("myers: too many query tiles for one launch");
        return BGSA_HIP_EUNSUPPORTED;
    }
    if (int rc = launch_pack_queries(d_content, ref_len, ref_start, ref_end, d_workspace, stream)) return rc;
    unsigned *fault = nullptr;
    if (int rc = stream_guard(d_workspace, static_cast<int>(stream_stride(ref_len)), kCodeRefill, 7, stream, &fault)) return rc;
    hipLaunchKernelGGL((myers_semi_asm_kernel<NW>), grid, dim3(256), 0, stream,
                       static_cast<const unsigned char *>(d_workspace), d_peq, d_results, ref_len,
                       read_len, static_cast<long long>(read_count), static_cast<int>(n_groups), word_num,
                       nq, q_tile, static_cast<int>(stream_stride(ref_len)), fault);
    BGSA_HIP_TRY(hipGetLastError());
    return BGSA_HIP_OK;
}

template <int NW, bool SEMI = false>
int launch_planes(const char *d_content, const uint32_t *d_peq, int16_t *d_results, int ref_len,
                  int read_len, int64_t read_count, int ref_start, int ref_end, int word_num,
                  void *d_workspace, hipStream_t stream)
{
    const int nq = ref_end - ref_start;
    const int64_t n_groups = read_count / kLanes;
    // a task is already long: 8 queries x ref_len rows x 11*NW instructions.  Counter: two waves per SIMD with or without the
    // loop's registers from 26 words up; narrower A/B widths keep theirs
    const TaskPlan plan = plan_tasks(nq, n_groups, static_cast<long long>(ref_len) * NW, 8, !SEMI && NW >= 26);
    const int q_tile = plan.q_tile;
    note_query_tile(q_tile);
    dim3 grid(static_cast<unsigned>((n_groups + kWavesPerBlock - 1) / kWavesPerBlock),
              static_cast<unsigned>((nq + q_tile - 1) / q_tile));
    const bool dynamic = plan.dynamic;
    if (grid.y > 65535u && !dynamic) {
        set_error_text("myers: too many query tiles for one launch");
        return BGSA_HIP_EUNSUPPORTED;
    }
    unsigned *counter = dynamic ? task_counter_in(d_workspace, stream_stride(ref_len) * static_cast<size_t>(nq)) : nullptr;
    if (int rc = launch_pack_queries(d_content, ref_len, ref_start, ref_end, d_workspace, stream, counter)) return rc;
    unsigned *fault = nullptr;
    if (int rc = stream_guard(d_workspace, static_cast<int>(stream_stride(ref_len)), kCodeRefill, 7, stream, &fault)) return rc;
    if constexpr (SEMI)
        hipLaunchKernelGGL((myers_semi_planes_kernel<NW>), grid, dim3(256), 0, stream,
                           static_cast<const unsigned char *>(d_workspace), d_peq, d_results, ref_len,
                           read_len, static_cast<long long>(read_count), static_cast<int>(n_groups), word_num,
                           nq, q_tile, static_cast<int>(stream_stride(ref_len)), fault);
    else if (dynamic) {
        const long long blocks = static_cast<long long>(grid.x) * grid.y;
        const int resident = persistent_blocks_for(myers_global_planes_kernel<NW, true>);
        hipLaunchKernelGGL((myers_global_planes_kernel<NW, true>), dim3(static_cast<unsigned>(blocks < resident ? blocks : resident)),
                           dim3(256), 0, stream, static_cast<const unsigned char *>(d_workspace), d_peq, d_results, ref_len,
                           read_len, static_cast<long long>(read_count), static_cast<int>(n_groups), word_num,
                           nq, q_tile, static_cast<int>(stream_stride(ref_len)), fault, counter);
    } else
        hipLaunchKernelGGL((myers_global_planes_kernel<NW, false>), grid, dim3(256), 0, stream,
                           static_cast<const unsigned char *>(d_workspace), d_peq, d_results, ref_len,
                           read_len, static_cast<long long>(read_count), static_cast<int>(n_groups), word_num,
                           nq, q_tile, static_cast<int>(stream_stride(ref_len)), fault, nullptr);
    BGSA_HIP_TRY(hipGetLastError());
    return BGSA_HIP_OK;
}

template <int NW, bool PEQ = false, bool SEMI = false>
int launch_blocked(const char *d_content, const uint32_t *d_peq, int16_t *d_results, int ref_len, int read_len,
                   int64_t read_count, int ref_start, int ref_end, int word_num, int n_blocks, void *d_workspace,
                   hipStream_t stream)
{
    const int nq = ref_end - ref_start;
    const int stride = blocked_stream_layout(ref_len, nullptr, nullptr);
    const size_t stream_bytes = (static_cast<size_t>(stride) * nq + 255) & ~static_cast<size_t>(255);
    if (int rc = launch_pack_blocked(d_content, ref_len, ref_start, ref_end, d_workspace, stream)) return rc;
    uint32_t *carry = reinterpret_cast<uint32_t *>(static_cast<unsigned char *>(d_workspace) + stream_bytes);
    unsigned long long *counter = reinterpret_cast<unsigned long long *>(
        reinterpret_cast<unsigned char *>(carry) + blocked_carry_bytes(ref_len, 3));
    BGSA_HIP_TRY(hipMemsetAsync(counter, 0, sizeof(unsigned long long), stream));
    unsigned *fault = nullptr;
    if (int rc = stream_guard(d_workspace, stride, kCodeRefill, -1, stream, &fault)) return rc;
    hipLaunchKernelGGL((myers_blocked_kernel<NW, PEQ, SEMI>), dim3(blocked_workgroups()), dim3(256), 0, stream,
                       static_cast<const unsigned char *>(d_workspace), d_peq, d_results, carry, ref_len, read_len,
                       static_cast<long long>(read_count), static_cast<int>(read_count / kLanes), word_num, nq,
                       (note_query_tile(blocked_q_tile(nq, read_count / kLanes)), blocked_q_tile(nq, read_count / kLanes)),
                       stride, n_blocks, counter, fault);
    BGSA_HIP_TRY(hipGetLastError());
    return BGSA_HIP_OK;
}

// Block width for a subject of word_num > 32 words: the narrowest instantiated width that covers
// it with the fewest blocks (35 words -> 2 x 20, not 2 x 28).  28 words is the widest block that
// keeps two waves per SIMD (32 needs 256 VGPRs: measured 93 vs 168 TCUPS).
int pick_block_nw(int word_num, int *n_blocks)
{
    constexpr int kWidest = 28;
    const int blocks = (word_num + kWidest - 1) / kWidest;
    const int need = (word_num + blocks - 1) / blocks;
    for (int nw : {12, 14, 16, 18, 20, 22, 24, 26, 28})
        if (nw >= need) {
            *n_blocks = (word_num + nw - 1) / nw;
            return nw;
        }
    *n_blocks = blocks;
    return kWidest;
}

// Column blocks with resident Peq planes: 12..20 words (BGSA_MYERS_BLOCK_FORM=planes selects the
// code-plane blocks of up to 28 words instead, the A/B reference).
bool peq_blocks()
{
    static const bool on = [] {
        const char *e = getenv("BGSA_MYERS_BLOCK_FORM");
        return !(e && strcmp(e, "planes") == 0);
    }();
    return on;
}
int pick_peq_block_nw(int word_num, int *n_blocks)
{
    constexpr int kWidest = 20;  // 238 VGPRs: two waves per SIMD (22 words would need 256)
    const int blocks = (word_num + kWidest - 1) / kWidest;
    const int need = (word_num + blocks - 1) / blocks;
    for (int nw : {12, 14, 16, 18, 20})
        if (nw >= need) {
            *n_blocks = (word_num + nw - 1) / nw;
            return nw;
        }
    *n_blocks = blocks;
    return kWidest;
}

int pick_planes_nw(int word_num)
{
    for (int nw : {10, 12, 14, 16, 18, 20, 22, 24, 26, 28, 30, 32})
        if (nw >= word_num) return nw;
    return -1;
}

template <int NW, int G, bool SEMI = false>
int launch_nw(const char *d_content, const uint32_t *d_peq, int16_t *d_results, int ref_len,
              int read_len, int64_t read_count, int ref_start, int ref_end, int word_num,
              hipStream_t stream)
{
    const int nq = ref_end - ref_start;
    const int64_t n_groups = read_count / kLanes;
    const int q_tile = pick_q_tile(nq, (n_groups + G - 1) / G, ref_len, NW * G);
    note_query_tile(q_tile);
    dim3 grid(static_cast<unsigned>((n_groups + kWavesPerBlock * G - 1) / (kWavesPerBlock * G)),
              static_cast<unsigned>((nq + q_tile - 1) / q_tile));
    if (grid.y > 65535u) {
        set_error_text("myers: too many query tiles for one launch");
        return BGSA_HIP_EUNSUPPORTED;
    }
    hipLaunchKernelGGL((myers_global_kernel<NW, G, SEMI>), grid, dim3(256), 0, stream, d_content, d_peq,
                       d_results, ref_len, read_len, static_cast<long long>(read_count),
                       static_cast<int>(n_groups), word_num, ref_start, ref_end, q_tile);
    BGSA_HIP_TRY(hipGetLastError());
    return BGSA_HIP_OK;
}

}  // namespace

int myers_max_plain_words()
{
    static const int limit = [] {
        const char *e = getenv("BGSA_MYERS_MAX_PLAIN_WORDS");   // measurement knob
        const int v = e ? atoi(e) : kMaxWords;
        return (v >= 8 && v <= kMaxWords) ? v : kMaxWords;
    }();
    return limit;
}

// Widest subject (words) that keeps its five Peq planes in registers (8 VALU per word); wider ones use
// the 3-bit code planes (9 per word, fewer registers).  BGSA_MYERS_PEQ_MAX_WORDS overrides (measurement).
int myers_peq_max_words()
{
    static const int limit = [] {
        const char *e = getenv("BGSA_MYERS_PEQ_MAX_WORDS");
        const int v = e ? atoi(e) : kPeqMaxWords;
        return (v >= 8 && v <= 32) ? v : kPeqMaxWords;
    }();
    return limit;
}

// Semi-global: widest subject (words) scored without column blocks — resident Peq planes (myers_semi_asm_kernel) up
// to myers_peq_max_words(), the code planes (myers_semi_planes_kernel) up to 32 words; wider ones run as column blocks.
int myers_semi_max_plain_words() { return 32; }

int pick_semi_planes_nw(int word_num)
{
    for (int nw : {26, 28, 30, 32})
        if (nw >= word_num) return nw;
    return -1;
}

int pick_peq_nw(int word_num)
{
    if (word_num > myers_peq_max_words()) return -1;
    for (int nw : {1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16, 18, 20, 22, 24, 25, 26, 28, 30, 32})
        if (nw >= word_num) return nw;
    return -1;
}

const char *myers_kernel_name(int word_num, int semi_global)
{
    static thread_local char name[64];
    const int nw = pick_nw(word_num);
    if (semi_global) {
        int n_blocks = 0;
        if (myers_impl() != 0)
            snprintf(name, sizeof name, "myers_global_kernel<%d, 1, true>", nw);
        else if (word_num <= std::min(myers_peq_max_words(), kSemiPeqMaxWords))
            snprintf(name, sizeof name, "myers_semi_asm_kernel<%d>", pick_peq_nw(word_num));
        else if (word_num <= myers_semi_max_plain_words())
            snprintf(name, sizeof name, "myers_semi_planes_kernel<%d>", pick_semi_planes_nw(word_num));
        else
            snprintf(name, sizeof name, "myers_blocked_kernel<%d, true, true>", pick_peq_block_nw(word_num, &n_blocks));
        return name;
    }
    if (word_num > myers_max_plain_words()) {
        int n_blocks = 0;
        if (peq_blocks())
            snprintf(name, sizeof name, "myers_blocked_kernel<%d, true>", pick_peq_block_nw(word_num, &n_blocks));
        else
            snprintf(name, sizeof name, "myers_blocked_kernel<%d>", pick_block_nw(word_num, &n_blocks));
        return name;
    }
    if (myers_impl() == 0 && pick_peq_nw(word_num) > 0)
        snprintf(name, sizeof name, "myers_global_asm_kernel<%d, %d>", pick_peq_nw(word_num),
                 (word_num <= kPairMaxWords && pair_groups() == 2) ? 2 : 1);
    else if (myers_impl() == 0 && pick_planes_nw(word_num) > 0)
        snprintf(name, sizeof name, "myers_global_planes_kernel<%d>", pick_planes_nw(word_num));
    else
        snprintf(name, sizeof name, "myers_global_kernel<%d, 1>", nw);
    return name;
}

int launch_myers(const char *d_content, const uint32_t *d_peq, int16_t *d_results, int ref_len,
                 int read_len, int64_t read_count, int ref_start, int ref_end, int word_num,
                 void *d_workspace, hipStream_t stream, int semi_global)
{
    if (ref_end <= ref_start || read_count == 0) return BGSA_HIP_OK;
    if (semi_global && myers_impl() == 0) {
        // generated-asm kernels: resident Peq planes up to 24 words, code planes up to 32, column blocks (any length) beyond
        if (word_num > std::min(myers_peq_max_words(), kSemiPeqMaxWords) && word_num <= myers_semi_max_plain_words()) {
            switch (pick_semi_planes_nw(word_num)) {
#define BGSA_SEMI_PLANES_CASE(N)                                                                \
    case N:                                                                                     \
        return launch_planes<N, true>(d_content, d_peq, d_results, ref_len, read_len, read_count, \
                                      ref_start, ref_end, word_num, d_workspace, stream);
                BGSA_SEMI_PLANES_CASE(26) BGSA_SEMI_PLANES_CASE(28) BGSA_SEMI_PLANES_CASE(30) BGSA_SEMI_PLANES_CASE(32)
#undef BGSA_SEMI_PLANES_CASE
            default: break;
            }
        }
        if (word_num <= std::min(myers_peq_max_words(), kSemiPeqMaxWords)) {
            switch (pick_peq_nw(word_num)) {
#define BGSA_SEMI_CASE(N)                                                                       \
    case N:                                                                                     \
        return launch_semi_asm<N>(d_content, d_peq, d_results, ref_len, read_len, read_count,   \
                                  ref_start, ref_end, word_num, d_workspace, stream);
                BGSA_SEMI_CASE(1) BGSA_SEMI_CASE(2) BGSA_SEMI_CASE(3) BGSA_SEMI_CASE(4) BGSA_SEMI_CASE(5)
                BGSA_SEMI_CASE(6) BGSA_SEMI_CASE(7) BGSA_SEMI_CASE(8) BGSA_SEMI_CASE(10) BGSA_SEMI_CASE(12)
                BGSA_SEMI_CASE(14) BGSA_SEMI_CASE(16) BGSA_SEMI_CASE(18) BGSA_SEMI_CASE(20) BGSA_SEMI_CASE(22)
                BGSA_SEMI_CASE(24) BGSA_SEMI_CASE(25) BGSA_SEMI_CASE(26) BGSA_SEMI_CASE(28) BGSA_SEMI_CASE(30) BGSA_SEMI_CASE(32)
#undef BGSA_SEMI_CASE
            default: break;
            }
        }
        int n_blocks = 0;
        switch (pick_peq_block_nw(word_num, &n_blocks)) {
#define BGSA_BLOCK_CASE(N)                                                                       \
    case N:                                                                                      \
        return launch_blocked<N, true, true>(d_content, d_peq, d_results, ref_len, read_len, read_count, ref_start, \
                                             ref_end, word_num, n_blocks, d_workspace, stream);
            BGSA_BLOCK_CASE(12) BGSA_BLOCK_CASE(14) BGSA_BLOCK_CASE(16) BGSA_BLOCK_CASE(18) BGSA_BLOCK_CASE(20)
#undef BGSA_BLOCK_CASE
        default: break;
        }
    }
#if !BGSA_AB_KERNELS
    if (myers_impl() == 1) return ab_knob_refused("BGSA_MYERS_IMPL=c");
    if (semi_global) {
        set_error_text("myers: no semi-global kernel for this word count");
        return BGSA_HIP_EUNSUPPORTED;
    }
#else
    if (semi_global) {  // BGSA_MYERS_IMPL=c: the compiler-scheduled kernel, subjects up to 1024 bp (A/B reference)
        switch (pick_nw(word_num)) {
#define BGSA_CASE(N)                                                                            \
    case N:                                                                                     \
        return launch_nw<N, 1, true>(d_content, d_peq, d_results, ref_len, read_len, read_count, \
                                     ref_start, ref_end, word_num, stream);
            BGSA_CASE(1) BGSA_CASE(2) BGSA_CASE(3) BGSA_CASE(4) BGSA_CASE(5) BGSA_CASE(6)
            BGSA_CASE(7) BGSA_CASE(8) BGSA_CASE(10) BGSA_CASE(12) BGSA_CASE(14) BGSA_CASE(16)
            BGSA_CASE(20) BGSA_CASE(24) BGSA_CASE(28) BGSA_CASE(32)
#undef BGSA_CASE
        default:
            set_error_text("myers: the compiler-scheduled semi-global kernel (BGSA_MYERS_IMPL=c) covers subjects up to 1024 bp");
            return BGSA_HIP_EUNSUPPORTED;
        }
    }
    if (word_num > myers_max_plain_words() && myers_impl() == 1)  // A/B: the state-in-memory C++ kernel
        return launch_long(BGSA_ALGO_MYERS, d_content, d_peq, d_results, ref_len, read_len, read_count, ref_start,
                           ref_end, word_num, d_workspace, stream);
#endif
    if (word_num > myers_max_plain_words() && peq_blocks()) {
        int n_blocks = 0;
        switch (pick_peq_block_nw(word_num, &n_blocks)) {
#define BGSA_BLOCK_CASE(N)                                                                       \
    case N:                                                                                      \
        return launch_blocked<N, true>(d_content, d_peq, d_results, ref_len, read_len, read_count, ref_start, \
                                       ref_end, word_num, n_blocks, d_workspace, stream);
            BGSA_BLOCK_CASE(12) BGSA_BLOCK_CASE(14) BGSA_BLOCK_CASE(16) BGSA_BLOCK_CASE(18) BGSA_BLOCK_CASE(20)
#undef BGSA_BLOCK_CASE
        default: break;
        }
    }
#if !BGSA_AB_KERNELS
    if (word_num > myers_max_plain_words()) return ab_knob_refused("BGSA_MYERS_BLOCK_FORM=planes");
#else
    if (word_num > myers_max_plain_words()) {
        int n_blocks = 0;
        switch (pick_block_nw(word_num, &n_blocks)) {
#define BGSA_BLOCK_CASE(N)                                                                       \
    case N:                                                                                      \
        return launch_blocked<N>(d_content, d_peq, d_results, ref_len, read_len, read_count, ref_start, \
                                 ref_end, word_num, n_blocks, d_workspace, stream);
            BGSA_BLOCK_CASE(12) BGSA_BLOCK_CASE(14) BGSA_BLOCK_CASE(16) BGSA_BLOCK_CASE(18) BGSA_BLOCK_CASE(20)
            BGSA_BLOCK_CASE(22) BGSA_BLOCK_CASE(24) BGSA_BLOCK_CASE(26) BGSA_BLOCK_CASE(28)
#undef BGSA_BLOCK_CASE
        default: break;
        }
    }
#endif
    if (myers_impl() == 0) {
        if (word_num <= kPairMaxWords && pair_groups() == 2 && read_count >= 2 * kLanes) {   // short subjects: two groups per wave
            if (word_num == 1)
                return launch_asm<1, 2>(d_content, d_peq, d_results, ref_len, read_len, read_count, ref_start, ref_end, word_num,
                                        d_workspace, stream);
            return launch_asm<2, 2>(d_content, d_peq, d_results, ref_len, read_len, read_count, ref_start, ref_end, word_num,
                                    d_workspace, stream);
        }
        switch (pick_peq_nw(word_num)) {
#define BGSA_ASM_CASE(N)                                                                        \
    case N:                                                                                     \
        return launch_asm<N, 1>(d_content, d_peq, d_results, ref_len, read_len, read_count,     \
                                ref_start, ref_end, word_num, d_workspace, stream);
            BGSA_ASM_CASE(1) BGSA_ASM_CASE(2) BGSA_ASM_CASE(3) BGSA_ASM_CASE(4) BGSA_ASM_CASE(5)
            BGSA_ASM_CASE(6) BGSA_ASM_CASE(7) BGSA_ASM_CASE(8) BGSA_ASM_CASE(10) BGSA_ASM_CASE(12)
            BGSA_ASM_CASE(14) BGSA_ASM_CASE(16) BGSA_ASM_CASE(18) BGSA_ASM_CASE(20) BGSA_ASM_CASE(22)
            BGSA_ASM_CASE(24) BGSA_ASM_CASE(25) BGSA_ASM_CASE(26) BGSA_ASM_CASE(28) BGSA_ASM_CASE(30) BGSA_ASM_CASE(32)
#undef BGSA_ASM_CASE
        default: break;
        }
        switch (pick_planes_nw(word_num)) {
#define BGSA_PLANES_CASE(N)                                                                     \
    case N:                                                                                     \
        return launch_planes<N>(d_content, d_peq, d_results, ref_len, read_len, read_count,     \
                                ref_start, ref_end, word_num, d_workspace, stream);
#if BGSA_AB_KERNELS   // the code planes below 29 words: only under BGSA_MYERS_PEQ_MAX_WORDS (resident Peq planes measured faster)
            BGSA_PLANES_CASE(10) BGSA_PLANES_CASE(12) BGSA_PLANES_CASE(14) BGSA_PLANES_CASE(16)
            BGSA_PLANES_CASE(18) BGSA_PLANES_CASE(20) BGSA_PLANES_CASE(22) BGSA_PLANES_CASE(24)
            BGSA_PLANES_CASE(26) BGSA_PLANES_CASE(28)
#endif
            BGSA_PLANES_CASE(30) BGSA_PLANES_CASE(32)
#undef BGSA_PLANES_CASE
        default: break;
        }
    }
#if !BGSA_AB_KERNELS
    if (myers_impl() == 1) return ab_knob_refused("BGSA_MYERS_IMPL=c");
    if (myers_peq_max_words() != kPeqMaxWords) return ab_knob_refused("BGSA_MYERS_PEQ_MAX_WORDS");
    set_error_text("myers: no kernel for this word count");
    return BGSA_HIP_EUNSUPPORTED;
#else
    switch (pick_nw(word_num)) {
#define BGSA_CASE(N)                                                                            \
    case N:                                                                                     \
        return launch_nw<N, 1>(d_content, d_peq, d_results, ref_len, read_len, read_count,         \
                            ref_start, ref_end, word_num, stream);
        BGSA_CASE(1) BGSA_CASE(2) BGSA_CASE(3) BGSA_CASE(4) BGSA_CASE(5) BGSA_CASE(6)
        BGSA_CASE(7) BGSA_CASE(8) BGSA_CASE(10) BGSA_CASE(12) BGSA_CASE(14) BGSA_CASE(16)
        BGSA_CASE(20) BGSA_CASE(24) BGSA_CASE(28) BGSA_CASE(32)
#undef BGSA_CASE
    default:
        set_error_text("myers: no kernel for this word count");
        return BGSA_HIP_EUNSUPPORTED;
    }
#endif
}

}  // namespace bgsa
