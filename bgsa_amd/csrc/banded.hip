// banded.hip — BGSA's banded Myers filter, one subject per lane, gfx950.
//
// Replaces the reference's banded align_cpu (banded/BGSA_CPU/align_core.c:69-252) and its
// preprocess (banded/BGSA_CPU/global.c:25-84).  Output: int8, the band-limited distance
// (minimum over the last row inside the band) or MAX_ERROR = 127 when the running error on the
// lowest diagonal exceeds k + h + 1 at the reference's last checkpoint.
//
// What the reference does per row, per pair: one Hyyrö step on a single 64-bit word that slides
// down the diagonal (cpu_cal_D0, :19-33), then shifts FIVE 64-bit match windows right by one and
// feeds each a new bit (cpu_move_peq / cpu_or_peq, :35-62) — 30 of its 42 ops per row move
// windows.  Here the windows are never moved: the preprocess stores, per character class, the
// subject's match bit-string offset by k+1 zero bits ("Mext": bit i = subject[i-(k+1)] == c), and
// the window of row r is just the band_length bits of Mext starting at bit r — one funnel shift
// with a wave-uniform amount, for the current query character only.  That is exactly the window
// the reference holds at row r (DESIGN.md "banded"): word 0 of its layout is Mext bits 0..2k,
// and each row's shift-and-feed advances the same bit-string by one.
//
// Early exit: the reference tests err > max_err after row min(64, m), then every 16 rows, last at
// row max(64, n-h) (:136-140,170-174,199-203,216-220).  err never decreases, so a pair is rejected iff the
// test holds at that LAST checkpoint; any earlier test that fires implies it.  A wave therefore
// tests every 16 rows from the start and stops as soon as all 64 lanes are past the limit.
//
// Band of 2k+1 (+1 carry) bits: one 32-bit word for k <= 15, a 64-bit pair for k <= 31; any length.
// Supported domain: query_len == subject_len (the reference's band is mis-aligned otherwise,
// SURVEY.md §8(a) A5) — anything else is refused loudly.
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "bgsa_common.h"

namespace bgsa {

template <typename T>
struct BandWord;
template <>
struct BandWord<uint32_t> {
    static constexpr int bits = 32;
    // bits [sh, sh+32) of the 64-bit pair {hi:lo}; sh is wave-uniform in 0..31
    static __device__ __forceinline__ uint32_t funnel(uint32_t hi, uint32_t lo, int sh)
    {
        return static_cast<uint32_t>(((static_cast<uint64_t>(hi) << 32) | lo) >> sh);
    }
};
template <>
struct BandWord<uint64_t> {
    static constexpr int bits = 64;
    static __device__ __forceinline__ uint64_t funnel(uint64_t hi, uint64_t lo, int sh)
    {
        return sh == 0 ? lo : ((lo >> sh) | (hi << (64 - sh)));
    }
};

// One row (cpu_cal_D0 :19-33 + cpu_cal_score :64-67).  `win` already masked to the band.
template <typename T>
__device__ __forceinline__ void band_row(T win, T &vp, T &vn, uint32_t &acc)
{
    const T x = win | vn;
    const T d0 = (((x & vp) + vp) ^ vp) | x;
    const T hn = d0 & vp;
    const T hp = ~(d0 | vp) | vn;
    const T x2 = d0 >> 1;
    vn = x2 & hp;
    vp = ~(hp | x2) | hn;
    acc += 1u - static_cast<uint32_t>(d0 & 1);
}

#if BGSA_AB_KERNELS
// Compiler-scheduled kernel of the same algorithm (A/B reference for the asm kernels, selected by
// BGSA_BANDED_IMPL=c).  T = band word: uint32_t for k <= 15, uint64_t for k <= 31.  The first four
// 32-bit Mext words per class are register-resident (rows 0..31 need words 0..2); beyond that the
// wave re-reads, every 32 rows, the word the next windows straddle.
template <typename T>
__global__ __launch_bounds__(256) void banded_kernel(
    const char *__restrict__ content, const uint32_t *__restrict__ mext, int8_t *__restrict__ out,
    int len, long long ld, int n_groups, int word_num, int ref_start, int ref_end, int q_tile, int k)
{
    constexpr int W = BandWord<T>::bits;
    const int lane = threadIdx.x & (kLanes - 1);
    const int group = __builtin_amdgcn_readfirstlane(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));
    if (group >= n_groups) return;
    const uint32_t *g = mext + static_cast<size_t>(group) * kChars * word_num * kLanes + lane;

    uint32_t first[kChars][4];
#pragma unroll
    for (int c = 0; c < kChars; c++)
#pragma unroll
        for (int w = 0; w < 4; w++) first[c][w] = (w < word_num) ? g[(c * word_num + w) * kLanes] : 0u;

    const int h = k;                       // h_threshold = k + n - m with n == m (:70)
    const int band_down = k + h;           // band_length - 1 (:71-72)
    const T band_mask = (band_down + 1 >= W) ? ~T(0) : ((T(1) << (band_down + 1)) - 1);
    const uint32_t max_err = static_cast<uint32_t>(k + h + 1);  // :114
    // Row count after which the reference runs its last test: after the first min(64, m) rows
    // (:125-140) and, for m > 64, again when its feed index reaches n, i.e. after n-h rows
    // (:216-220) — whichever is later; the final k rows are never tested (:221-226).
    const int last_check = (len <= 64) ? len : ((len - h > 64) ? len - h : 64);

    const int q0 = ref_start + blockIdx.y * q_tile;
    const int q1 = (q0 + q_tile < ref_end) ? q0 + q_tile : ref_end;
    int8_t *dst = out + static_cast<size_t>(group) * kLanes + lane;

    for (int q = q0; q < q1; q++) {
        T vp = 0, vn = 0;                  // :96-97
        uint32_t acc = 0;                  // err - k; rows < k do not score (:116-123)
        bool dead = false, all_dead = false;
        UniformBytes qs(content + static_cast<size_t>(q) * (len + 1));
        uint32_t x0[kChars], x1[kChars], x2[kChars];  // words wi, wi+1, wi+2 of every class
#pragma unroll
        for (int c = 0; c < kChars; c++) { x1[c] = first[c][0]; x2[c] = first[c][1]; x0[c] = 0u; }
        for (int r = 0; r < len && !all_dead; r++) {
            const int j = r & 31;
            if (j == 0) {
                const int wi = r >> 5;
#pragma unroll
                for (int c = 0; c < kChars; c++) {
                    x0[c] = x1[c];
                    x1[c] = x2[c];
                    if (wi + 2 < 4) x2[c] = first[c][(wi + 2) & 3];
                    else x2[c] = (wi + 2 < word_num) ? g[(c * word_num + wi + 2) * kLanes] : 0u;
                }
            }
            if ((r & 3) == 0) qs.refill(r, len - r);
            const uint32_t c = __builtin_amdgcn_readfirstlane(qs.next());
            uint32_t a, b, d;
            switch (c) {
            case 0: a = x0[0]; b = x1[0]; d = x2[0]; break;
            case 1: a = x0[1]; b = x1[1]; d = x2[1]; break;
            case 2: a = x0[2]; b = x1[2]; d = x2[2]; break;
            case 3: a = x0[3]; b = x1[3]; d = x2[3]; break;
            default: a = x0[4]; b = x1[4]; d = x2[4]; break;
            }
            const uint32_t lo = BandWord<uint32_t>::funnel(b, a, j);
            T win = lo;
            if constexpr (W == 64) win |= static_cast<T>(BandWord<uint32_t>::funnel(d, b, j)) << 32;
            if (r == k) acc = 0;           // scoring starts at row k with err = k (:116-134)
            band_row<T>(win & band_mask, vp, vn, acc);
            const int done = r + 1;
            if (done <= last_check && ((done & 15) == 0 || done == last_check)) {
                const bool over = done > k && (static_cast<uint32_t>(k) + acc > max_err);
                if (done == last_check) dead = over;
                if (__builtin_amdgcn_ballot_w64(!over) == 0) {  // every lane is past the limit
                    dead = true;
                    all_dead = true;
                }
            }
        }
        int8_t result = static_cast<int8_t>(HIP_MAX_ERROR);
        if (!all_dead) {
            // :230-245 — walk the last row across the band, keep the minimum.
            uint32_t err = static_cast<uint32_t>(k) + acc, best = err;
            for (int i = 0; i <= h; i++) {
                err += static_cast<uint32_t>((vp >> i) & 1);
                err -= static_cast<uint32_t>((vn >> i) & 1);
                best = err < best ? err : best;
            }
            if (!dead) result = static_cast<int8_t>(best);
        }
        dst[static_cast<size_t>(q - ref_start) * ld] = result;
    }
}
#endif  // BGSA_AB_KERNELS

namespace { int banded_impl(); }

// One listed pair of the regroup pass (see "regrouping of sparse survivors" below): entry = (query - q0) << 8 | lane of
// the wave's group; scored from row 0 with the recurrence, tests and final walk of banded_kernel<T>, the query
// character selected per lane.  g = the group's Mext block, out_tile = &out[q0][group * 64].
template <typename T>
__device__ __forceinline__ void banded_finish_pair(uint32_t entry, const uint32_t *__restrict__ g, const char *__restrict__ content,
                                                   int first_query_row, int len, int word_num, int k,
                                                   int8_t *__restrict__ out_tile, long long ld)
{
    constexpr int W = BandWord<T>::bits;
    const int ql = static_cast<int>(entry >> 8);
    const uint32_t sl = entry & 63u;
    const uint32_t *gl = g + sl;
    const unsigned char *qrow = reinterpret_cast<const unsigned char *>(content) + static_cast<size_t>(first_query_row + ql) * (len + 1);
    const int h = k;
    const T band_mask = (k + h + 1 >= W) ? ~T(0) : ((T(1) << (k + h + 1)) - 1);
    const uint32_t max_err = static_cast<uint32_t>(k + h + 1);
    const int last_check = (len <= 64) ? len : ((len - h > 64) ? len - h : 64);
    T vp = 0, vn = 0;
    uint32_t acc = 0;
    bool dead = false;
    uint32_t x0[kChars], x1[kChars], x2[kChars];
#pragma unroll
    for (int c = 0; c < kChars; c++) {
        x0[c] = 0u;
        x1[c] = gl[(c * word_num + 0) * kLanes];
        x2[c] = gl[(c * word_num + 1) * kLanes];
    }
    // The lane's query characters, 32 rows at a time and one chunk ahead of the rows that use them: a byte load per row put a
    // memory round trip into every row (round 2's form of this pass: ~600 cycles per row; 1 % dense survivors 128.5 -> 116.6 ms).
    // A row is len + 1 bytes ('\n' included), rows lie back to back: a 4-byte load at offset o stays inside the row iff o + 4 <= len + 1.
    auto load_chars = [&](int r0, uint32_t (&q)[8]) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int o = r0 + 4 * j;
            uint32_t w = 0;
            if (o + 4 <= len + 1) {
                __builtin_memcpy(&w, qrow + o, 4);
            } else {
                for (int b = 0; b < 4; b++)
                    if (o + b < len) w |= static_cast<uint32_t>(qrow[o + b]) << (8 * b);
            }
            q[j] = w;
        }
    };
    uint32_t qnext[8];
    load_chars(0, qnext);
    for (int r0 = 0; r0 < len; r0 += 32) {
        const int wi = r0 >> 5;
        uint32_t qc[8];
#pragma unroll
        for (int j = 0; j < 8; j++) qc[j] = qnext[j];
        if (r0 + 32 < len) load_chars(r0 + 32, qnext);
#pragma unroll
        for (int c = 0; c < kChars; c++) {
            x0[c] = x1[c];
            x1[c] = x2[c];
            x2[c] = (wi + 2 < word_num) ? gl[(c * word_num + wi + 2) * kLanes] : 0u;
        }
        const int rows = len - r0 < 32 ? len - r0 : 32;
#pragma unroll
        for (int jj = 0; jj < 8; jj++) {       // unrolled over the chunk's eight character words: the word is a register, not a select
        const uint32_t word = qc[jj];
#pragma unroll 1
        for (int jb = 0; jb < 4; jb++) {
            const int j = 4 * jj + jb;
            if (j >= rows) break;                // wave-uniform
            const int r = r0 + j;
            uint32_t c = (word >> (8 * jb)) & 0xffu;
            c = c > 4u ? 0u : c;
            // the lane's class picks its words: compare-and-select, no control flow (lanes hold different queries)
            uint32_t a = x0[0], b = x1[0], d = x2[0];
#pragma unroll
            for (uint32_t cc = 1; cc < kChars; cc++) {
                const bool is = c == cc;
                a = is ? x0[cc] : a;
                b = is ? x1[cc] : b;
                if constexpr (W == 64) d = is ? x2[cc] : d;
            }
            T win = BandWord<uint32_t>::funnel(b, a, j);
            if constexpr (W == 64) win |= static_cast<T>(BandWord<uint32_t>::funnel(d, b, j)) << 32;
            if (r == k) acc = 0;
            band_row<T>(win & band_mask, vp, vn, acc);
            if (r + 1 == last_check) dead = static_cast<uint32_t>(k) + acc > max_err;
        }
        }
    }
    int8_t result = static_cast<int8_t>(HIP_MAX_ERROR);
    if (!dead) {
        uint32_t err = static_cast<uint32_t>(k) + acc, best = err;
        for (int i = 0; i <= h; i++) {
            err += static_cast<uint32_t>((vp >> i) & 1);
            err -= static_cast<uint32_t>((vn >> i) & 1);
            best = err < best ? err : best;
        }
        result = static_cast<int8_t>(best);
    }
    out_tile[static_cast<size_t>(ql) * ld + sl] = result;
}

// The dense pass of the one-word-window kernels (k <= 12): the same pair-per-lane rescoring as banded_finish_pair with
// every instruction of its row in the fast issue class — a half-rate-class instruction makes its whole row issue at the
// slow rate, and banded_finish_pair's row holds a 64-bit shift, four v_cmp and eight v_cndmask.
//   * windows as in the wave's own loop: per class the 32 bits of the match string cut every `cut` rows (five funnel shifts
//     per cut, wave-uniform amount), the row shifts one word right by row mod cut;
//   * the lane's class picks its word through three masks made of the class code's bits (0 - bit) and a four-deep tree of
//     bitwise selects (v_bitop3) instead of compares and conditional moves;
//   * characters that are no class (> 4) are cleared four at a time when the chunk is loaded (they score as class 0, as the
//     stream packers have it), not compared per row.
__device__ __forceinline__ uint32_t banded_clear_foreign_bytes(uint32_t w)
{
    // per byte: >= 5 (or >= 128) -> 0.  (b & 0x7f) + 123 carries into bit 7 iff (b & 0x7f) >= 5; no carry crosses a byte.
    const uint32_t flag = (((w & 0x7f7f7f7fu) + 0x7b7b7b7bu) | w) & 0x80808080u;
    const uint32_t ones = (flag - (flag >> 7)) | flag;   // 0xff where flagged
    return w & ~ones;
}

// E[class of the lane], the class code in the low bits of `cs` (0..4), in twelve fast-class instructions.  Inline asm because
// the compiler turns the same arithmetic back into v_bfe / v_cmp / v_cndmask / v_and_or — all half-rate class.
// v_bitop3 0xd8: (a, b, c) -> c ? b : a.
__device__ __forceinline__ uint32_t banded_pick_word(const uint32_t (&E)[kChars], uint32_t cs)
{
    uint32_t m0, m1, m2, t01, t23;
    asm("v_and_b32 %0, 1, %5\n\t"
        "v_lshrrev_b32 %1, 1, %5\n\t"
        "v_lshrrev_b32 %2, 2, %5\n\t"
        "v_sub_u32 %0, 0, %0\n\t"
        "v_and_b32 %1, 1, %1\n\t"
        "v_and_b32 %2, 1, %2\n\t"
        "v_sub_u32 %1, 0, %1\n\t"
        "v_sub_u32 %2, 0, %2\n\t"
        "v_bitop3_b32 %3, %6, %7, %0 bitop3:0xd8\n\t"
        "v_bitop3_b32 %4, %8, %9, %0 bitop3:0xd8\n\t"
        "v_bitop3_b32 %3, %3, %4, %1 bitop3:0xd8\n\t"
        "v_bitop3_b32 %3, %3, %10, %2 bitop3:0xd8"
        : "=&v"(m0), "=&v"(m1), "=&v"(m2), "=&v"(t01), "=&v"(t23)
        : "v"(cs), "v"(E[0]), "v"(E[1]), "v"(E[2]), "v"(E[3]), "v"(E[4]));
    return t01;
}

__device__ __forceinline__ void banded_finish_pair_cut(uint32_t entry, const uint32_t *__restrict__ g, const char *__restrict__ content,
                                                       int first_query_row, int len, int word_num, int k, int cut,
                                                       int8_t *__restrict__ out_tile, long long ld)
{
    const int ql = static_cast<int>(entry >> 8);
    const uint32_t sl = entry & 63u;
    const uint32_t *gl = g + sl;
    const unsigned char *qrow = reinterpret_cast<const unsigned char *>(content) + static_cast<size_t>(first_query_row + ql) * (len + 1);
    const int h = k;
    const uint32_t band_mask = (1u << (k + h + 1)) - 1u;   // k <= 12: at most 25 bits
    const uint32_t max_err = static_cast<uint32_t>(k + h + 1);
    const int last_check = (len <= 64) ? len : ((len - h > 64) ? len - h : 64);
    uint32_t vp = 0, vn = 0, acc = 0;
    bool dead = false;
    uint32_t x0[kChars], x1[kChars];   // the two match-string words the chunk's 32 rows straddle
#pragma unroll
    for (int c = 0; c < kChars; c++) {
        x0[c] = 0u;
        x1[c] = gl[(c * word_num + 0) * kLanes];
    }
    // Four characters of the query row per load, the next four fetched while these are scored: two registers.  (Until the
    // row loop went from 91 to 71 registers this pass held sixteen — the characters of 32 rows and of the next 32 — and was
    // then what set the kernel's register count; it runs once per ~56 listed pairs, the row loop always.)
    auto load_chars = [&](int o) {
        uint32_t w = 0;
        if (o + 4 <= len + 1) {
            __builtin_memcpy(&w, qrow + o, 4);
        } else {
            for (int b = 0; b < 4; b++)
                if (o + b < len) w |= static_cast<uint32_t>(qrow[o + b]) << (8 * b);
        }
        return w;
    };
    uint32_t qnext = load_chars(0);
    for (int r0 = 0; r0 < len; r0 += 32) {
        const int wi = r0 >> 5;
#pragma unroll
        for (int c = 0; c < kChars; c++) {
            x0[c] = x1[c];
            x1[c] = (wi + 1 < word_num) ? gl[(c * word_num + wi + 1) * kLanes] : 0u;
        }
        const int rows = len - r0 < 32 ? len - r0 : 32;
        uint32_t E[kChars];
#pragma unroll 1
        for (int jj = 0; jj < 8; jj++) {       // the chunk's eight character words
            const int j0 = 4 * jj;
            if (j0 >= rows) break;               // wave-uniform
            if ((j0 & (cut - 1)) == 0) {         // a cut: cut is 16 or 8, so cuts fall on word boundaries of the characters
#pragma unroll
                for (int c = 0; c < kChars; c++) E[c] = BandWord<uint32_t>::funnel(x1[c], x0[c], j0);
            }
            const uint32_t word = banded_clear_foreign_bytes(qnext);
            if (r0 + j0 + 4 < len) qnext = load_chars(r0 + j0 + 4);
#pragma unroll 1
            for (int jb = 0; jb < 4; jb++) {
                const int j = j0 + jb;
                if (j >= rows) break;            // wave-uniform
                const int r = r0 + j;
                // the class code's three bits as masks (the characters are 0..4 here)
                const uint32_t w = banded_pick_word(E, word >> (8 * jb)) >> (j & (cut - 1));
                // rows k and last_check are wave-uniform events: real branches (the empty asm keeps the compiler from turning
                // them into a compare and a conditional move in EVERY row)
                if (r == k) { acc = 0; asm volatile("" : "+v"(acc)); }
                band_row<uint32_t>(w & band_mask, vp, vn, acc);
                if (r + 1 == last_check) { dead = static_cast<uint32_t>(k) + acc > max_err; asm volatile("" ::: "memory"); }
            }
        }
    }
    int8_t result = static_cast<int8_t>(HIP_MAX_ERROR);
    if (!dead) {
        uint32_t err = static_cast<uint32_t>(k) + acc, best = err;
        for (int i = 0; i <= h; i++) {
            err += (vp >> i) & 1u;
            err -= (vn >> i) & 1u;
            best = err < best ? err : best;
        }
        result = static_cast<int8_t>(best);
    }
    out_tile[static_cast<size_t>(ql) * ld + sl] = result;
}

// ---- generated row loop (gen_rows_asm.py: gen_banded_function) -----------------------------------
#include "banded_rows_gen.inc"

// The whole query runs inside one generated asm block — 12 VALU per row for k <= 15 (32-bit band
// word), 22 for k <= 31 (64-bit pair) — with the window advance / checkpoints / early exit driven by
// EVENT tokens of the packed stream (bgsa_common.h: banded_stream_layout).
// PHASE (BGSA_BANDED_IMPL=p; k <= 11, 32-bit band only; measured slower, kept as the A/B alternative): the band held
// in place for `phase` = banded_phase_rows(k) rows at a time (rows_ir.py: banded_phase_body) — 13 fast-class VALU per
// row, no half-rate instruction in the row, but a re-anchoring event per phase.
template <bool WIDE, bool PHASE = false>
__global__ __launch_bounds__(256) void banded_asm_kernel(
    const unsigned char *__restrict__ streams, const uint32_t *__restrict__ mext, int8_t *__restrict__ out,
    long long ld, int n_groups, int word_num, int n_queries, int q_tile, int k, int stream_stride_bytes,
    unsigned *__restrict__ fault_word, const char *__restrict__ content, int ref_start, int len,
    uint32_t push_row, uint32_t push_max, int phase)
{
    static_assert(!(WIDE && PHASE), "the band is held in place in the 32-bit word only");
    // survivors of this wave's query tile waiting for their dense pass: (query - q0) << 8 | lane
    __shared__ uint32_t s_regroup[kWavesPerBlock][kLanes];
    uint32_t *regroup = s_regroup[threadIdx.x >> 6];
    int n_regroup = 0;   // wave-uniform
    constexpr int NM = WIDE ? 4 : 3;  // resident 32-bit words per class, the last one is the prefetch target
    const int lane = threadIdx.x & (kLanes - 1);
    const int group = __builtin_amdgcn_readfirstlane(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));
    if (group >= n_groups) return;
    const uint32_t *g = mext + static_cast<size_t>(group) * kChars * word_num * kLanes;

    uint32_t first[kChars][NM];
    unsigned long long base[kChars];
#pragma unroll
    for (int c = 0; c < kChars; c++) {
        base[c] = uniform_u64(reinterpret_cast<unsigned long long>(g + static_cast<size_t>(c) * word_num * kLanes));
#pragma unroll
        for (int w = 0; w < NM; w++) first[c][w] = g[(c * word_num + w) * kLanes + lane];
    }
    const int h = k;
    const unsigned long long band = (k + h + 1 >= 64) ? ~0ull : ((1ull << (k + h + 1)) - 1ull);
    const uint32_t limit = static_cast<uint32_t>(h + 1);   // err > k+h+1  <=>  errors since row k > h+1

    const int q0 = blockIdx.y * q_tile;
    const int q1 = (q0 + q_tile < n_queries) ? q0 + q_tile : n_queries;
    int8_t *dst = out + static_cast<size_t>(group) * kLanes + lane;

    for (int q = q0; q <= q1; q++) {
        // Regrouping (one call site: the iteration q == q1 only flushes).  When the list could overflow, or the
        // tile is done, the waiting pairs are scored one per lane, densely, from row 0: same recurrence, tests
        // and final walk as banded_kernel<T>, with the query character selected per lane.  The lanes of this pass
        // are pairs of THIS wave's group and THIS tile's queries, so every load below stays inside the group's
        // Mext block and the tile's consecutive query rows.
        if (n_regroup > kLanes - static_cast<int>(push_max) || (q == q1 && n_regroup > 0)) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (lane < n_regroup)
                banded_finish_pair<typename std::conditional<WIDE, uint64_t, uint32_t>::type>(
                    regroup[lane], g, content, ref_start + q0, len, word_num, k,
                    out + static_cast<size_t>(q0) * ld + static_cast<size_t>(group) * kLanes, ld);
            __builtin_amdgcn_wave_barrier();
            n_regroup = 0;
        }
        if (q == q1) break;
        constexpr int NS = PHASE ? 6 : (WIDE ? 5 : 3);
        uint32_t st[NS];
#pragma unroll
        for (int i = 0; i < NS; i++) st[i] = 0u;
        uint32_t M[kChars][NM];
#pragma unroll
        for (int c = 0; c < kChars; c++)
#pragma unroll
            for (int w = 0; w < NM; w++) M[c][w] = first[c][w];
        uint32_t voff = static_cast<uint32_t>(lane * 4 + NM * kLanes * 4);  // word NM of this lane
        const unsigned long long s =
            reinterpret_cast<unsigned long long>(streams) + static_cast<unsigned long long>(q) * stream_stride_bytes;
        const int n_windows = __builtin_amdgcn_readfirstlane(stream_stride_bytes / 8 - 2);
        unsigned long long dead_mask;
        unsigned long long vp, vn;
        uint32_t acc;
        int left, early;
        if constexpr (WIDE) {
            dead_mask = banded_rows_asm64(st, M, voff, base, uniform_u64(s), n_windows, static_cast<uint32_t>(band),
                                          static_cast<uint32_t>(band >> 32), limit, push_row, push_max, left, early);
            vp = st[0] | (static_cast<unsigned long long>(st[1]) << 32);
            vn = st[2] | (static_cast<unsigned long long>(st[3]) << 32);
            acc = st[4];
        } else if constexpr (PHASE) {
            st[3] = static_cast<uint32_t>(band);   // the band mask at offset 0
            st[4] = 1u;                            // its lowest bit: the diagonal the errors are counted on
            uint32_t W[kChars];                    // row 0: the window is the first word itself
#pragma unroll
            for (int c = 0; c < kChars; c++) W[c] = first[c][0];
            dead_mask = banded_rows_phase_asm32(st, W, M, voff, base, uniform_u64(s), n_windows, static_cast<uint32_t>(band),
                                                static_cast<uint32_t>(phase), limit, push_row, push_max, left, early);
            // the state sits at the offset of the rows of the last phase; that phase's error bits are still uncounted
            const int final_shift = len - phase * ((len - 1) / phase);
            vp = st[0] >> final_shift;
            vn = st[1] >> final_shift;
            acc = st[2] + static_cast<uint32_t>(__popc(st[5]));
        } else {
            dead_mask = banded_rows_asm32(st, M, voff, base, uniform_u64(s), n_windows, static_cast<uint32_t>(band), 0u, limit,
                                          push_row, push_max, left, early);
            vp = st[0];
            vn = st[1];
            acc = st[2];
        }
        note_stream_fault(fault_word, left);
        const bool dead = (dead_mask >> lane) & 1ull;
        if (early) {
            // The wave stopped at a late test that found only a few lanes within the limit: those pairs wait in
            // the regroup list for the dense pass above; the other lanes are rejected here.
            const unsigned long long alive = ~dead_mask;
            if (dead)
                dst[static_cast<size_t>(q) * ld] = static_cast<int8_t>(HIP_MAX_ERROR);
            else
                regroup[n_regroup + __popcll(alive & ((1ull << lane) - 1ull))] = (static_cast<uint32_t>(q - q0) << 8) | lane;
            n_regroup += __builtin_amdgcn_readfirstlane(static_cast<int>(__popcll(alive)));
            continue;
        }
        int8_t result = static_cast<int8_t>(HIP_MAX_ERROR);
        if (dead_mask != ~0ull) {
            // :230-245 — walk the last row across the band, keep the minimum.
            uint32_t err = static_cast<uint32_t>(k) + acc, best = err;
            for (int i = 0; i <= h; i++) {
                err += static_cast<uint32_t>((vp >> i) & 1ull);
                err -= static_cast<uint32_t>((vn >> i) & 1ull);
                best = err < best ? err : best;
            }
            if (!dead) result = static_cast<int8_t>(best);
        }
        dst[static_cast<size_t>(q) * ld] = result;
    }
}

// ---- one-word windows, one or two subject groups per wave (k <= 12): banded_cut_rows_asm_g{1,2} ------------------------
// The default for k <= 12.  What round 3's microbenchmarks showed (scripts/ubench/gen_banded_mix.py,
// profiles/r03_ubench_banded_mix.txt): a single half-rate-class instruction makes its whole row issue at ~4.2 cycles per
// instruction instead of ~2.2 — the 12-instruction sliding row costs 52.8 cycles of vector issue with its v_alignbit and
// 28.4 without — so the funnel shift leaves the row: per class the wave holds the 32 bits of the match string that start
// at the last multiple of banded_cut_rows(k) rows (one v_alignbit per class and cut, in an event), and a row shifts that
// word by `row mod cut` with v_lshrrev_b32, which is fast class.  With the vector side that short the loop's scalar side
// (one scalar unit per CU) would bound it, so two subject groups share a wave and every dispatch, shift counter and event
// (G = 2, the default; BGSA_BANDED_GROUPS=1 is the A/B): groups 2w and 2w+1, rows interleaved instruction by instruction.
// The wave stops when all 128 lanes are past the limit; the regroup list takes (query, group, lane).
// Registers: the loop lives on the waves a SIMD can choose from (4 -> 5 -> 6 waves: +6-11 %, +4-6 %), so the row
// loop keeps two registers per class and group (gen_rows_asm.py: gen_banded_cut_function) and the dense pass is kept narrow:
// 73 VGPRs = six waves per SIMD.
// FORM (round 4): 0 = the one-word windows above (k <= 12); 1 / 2 = the SAME kernel around the funnel-shift rows of thresholds
// 13 .. 15 (32-bit band) / 16 .. 31 (64-bit pair) — banded_funnel{32,64}_rows_asm_g2: what two groups per wave, the woven dispatch,
// the solid-survivor rule and the task counter are worth to the rows that have to keep their v_alignbit.  Those
// forms keep three / four match-string words per class and group (the last one the prefetch target) and use the funnel-shift
// dense pass (banded_finish_pair<T>).  The pair row of the default (one group per wave) shifts D0 with ONE v_lshrrev_b64 on a fixed
// register pair (banded_funnel64s_rows_asm_g1: 21 VALU per row, -2 %; LABNOTES 9.6 has what else was tried on these rows).
template <int G, bool DYN = false, int FORM = 0>
__global__ __launch_bounds__(256) void banded_cut_kernel(
    const unsigned char *__restrict__ streams, const uint32_t *__restrict__ mext, int8_t *__restrict__ out,
    long long ld, int n_groups, int word_num, int n_queries, int q_tile, int k, int stream_stride_bytes,
    unsigned *__restrict__ fault_word, const char *__restrict__ content, int ref_start, int len,
    uint32_t push_row, uint32_t push_row_solid, uint32_t solid_limit, uint32_t push_max, uint32_t cut_rows,
    unsigned *__restrict__ task_counter)
{
    __shared__ uint32_t s_regroup[kWavesPerBlock][kLanes];   // (query - q0) << 8 | group in wave << 6 | lane
    uint32_t *regroup = s_regroup[threadIdx.x >> 6];
    const int lane = threadIdx.x & (kLanes - 1);
    // DYN: the waves of a persistent grid take (wave-group, tile) tasks from a counter (bgsa_common.h "dynamic task handout")
    const unsigned wave_groups = (static_cast<unsigned>(n_groups) + G - 1) / G;
    const unsigned n_tasks = wave_groups * ((static_cast<unsigned>(n_queries) + q_tile - 1) / q_tile);   // < 2^32: the launcher checked
    unsigned task = 0, task_issued = 0;
    if constexpr (DYN) {
        task = first_wave_task();
        if (task >= n_tasks) return;
    }
    do {
    if constexpr (DYN) task_issued = issue_wave_task(task_counter);   // the next one, asked for under this one's work
    int group0, tile;
    if constexpr (DYN) {
        group0 = static_cast<int>((task % wave_groups) * G);
        tile = static_cast<int>(task / wave_groups);
    } else {
        group0 = __builtin_amdgcn_readfirstlane((blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6)) * G);
        tile = blockIdx.y;
        if (group0 >= n_groups) return;
    }
    int n_regroup = 0;   // wave-uniform
    // DYN: what does not depend on the task is invariant in the task loop, and the compiler would keep it — per-lane base
    // pointers, the lane's byte offsets — in VGPRs across the row loop (89 instead of 73 registers: the sixth wave per SIMD).
    // Laundered through an empty asm these values are redefined per task as far as the compiler can tell, and recomputed.
    int word_num_t = word_num;
    const uint32_t *mext_t = mext;
    int8_t *out_t = out;
    unsigned lane_t = static_cast<unsigned>(lane);
    if constexpr (DYN) asm volatile("" : "+s"(word_num_t), "+s"(mext_t), "+s"(out_t), "+v"(lane_t));
    // an odd group count leaves the last wave's second half without a group: it runs the first one's again (loads only)
    const bool has[2] = {true, G == 2 && group0 + 1 < n_groups};
    const size_t group_words = static_cast<size_t>(kChars) * word_num_t * kLanes;
    const uint32_t *g = mext_t + static_cast<size_t>(group0) * group_words;
    const uint32_t gstride[2] = {0u, has[1] ? static_cast<uint32_t>(group_words * sizeof(uint32_t)) : 0u};

    constexpr int NM = FORM == 0 ? 2 : (FORM == 1 ? 3 : 4);   // resident match-string words per class and group
    constexpr int NS = FORM == 2 ? 5 : 3;                      // state registers per group
    // FORM 0: the first two words of every class stay in registers across the tile's queries (20 VGPRs with two groups); the
    // funnel forms would need 30 / 40 for that and re-read their first words per query instead (L2 hits: the group's block
    // is 5 x NM x 256 B per query against 150 rows x 24 / 44 vector instructions) — 92 -> 62 and 123 -> 83 VGPRs
    constexpr int NF = FORM == 0 ? NM : 1;
    uint32_t first[G][kChars][NF];   // (word NM on is fetched by the row loop)
    unsigned long long base[kChars];
#pragma unroll
    for (int c = 0; c < kChars; c++) {
        base[c] = uniform_u64(reinterpret_cast<unsigned long long>(g + static_cast<size_t>(c) * word_num_t * kLanes));
        if constexpr (FORM == 0) {
#pragma unroll
            for (int gg = 0; gg < G; gg++)
#pragma unroll
                for (int w = 0; w < NM; w++) first[gg][c][w] = g[gstride[gg] / 4 + (c * word_num_t + w) * kLanes + lane_t];
        }
    }
    const int h = k;
    const unsigned long long band64 = (k + h + 1 >= 64) ? ~0ull : ((1ull << (k + h + 1)) - 1ull);
    const uint32_t band = static_cast<uint32_t>(band64);
    const uint32_t limit = static_cast<uint32_t>(h + 1);   // err > k+h+1  <=>  errors since row k > h+1

    const int q0 = tile * q_tile;
    const int q1 = (q0 + q_tile < n_queries) ? q0 + q_tile : n_queries;
    int8_t *dst = out_t + static_cast<size_t>(group0) * kLanes;      // wave-uniform: the lane joins at the stores

    for (int q = q0; q <= q1; q++) {
        // the dense pass over the listed pairs (see banded_asm_kernel): one pair per lane, from row 0
        if (n_regroup > kLanes - static_cast<int>(push_max) || (q == q1 && n_regroup > 0)) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (lane < n_regroup) {
                const uint32_t entry = regroup[lane];
                const uint32_t gg = (entry >> 6) & 1u;
                int8_t *tile_out = out_t + static_cast<size_t>(q0) * ld + static_cast<size_t>(group0 + gg) * kLanes;
                if constexpr (FORM == 0)
                    banded_finish_pair_cut(entry & ~0x40u, g + gg * group_words, content, ref_start + q0, len, word_num_t, k, static_cast<int>(cut_rows),
                                           tile_out, ld);
                else
                    banded_finish_pair<typename std::conditional<FORM == 2, uint64_t, uint32_t>::type>(
                        entry & ~0x40u, g + gg * group_words, content, ref_start + q0, len, word_num_t, k, tile_out, ld);
            }
            __builtin_amdgcn_wave_barrier();
            n_regroup = 0;
        }
        if (q == q1) break;
        uint32_t st[NS * G];
#pragma unroll
        for (int i = 0; i < NS * G; i++) st[i] = 0u;
        uint32_t M[G][kChars][NM];
        uint32_t voff[G];
#pragma unroll
        for (int gg = 0; gg < G; gg++) {
#pragma unroll
            for (int c = 0; c < kChars; c++)
#pragma unroll
                for (int w = 0; w < NM; w++) {
                    if constexpr (FORM == 0) {
                        M[gg][c][w] = first[gg][c][w];   // the window of row 0 = the first word itself, and the word behind it
                    } else {
                        unsigned lane_m = static_cast<unsigned>(lane);
                        asm volatile("" : "+v"(lane_m));      // per query: the loads are not to be hoisted out of the query loop and kept
                        M[gg][c][w] = g[gstride[gg] / 4 + (c * word_num_t + w) * kLanes + lane_m];
                    }
                }
            unsigned lane_q = static_cast<unsigned>(lane);
            if constexpr (DYN) asm volatile("" : "+v"(lane_q));
            voff[gg] = static_cast<uint32_t>(lane_q * 4 + NM * kLanes * 4) + gstride[gg];   // word NM of this lane
        }
        const unsigned long long s =
            reinterpret_cast<unsigned long long>(streams) + static_cast<unsigned long long>(q) * stream_stride_bytes;
        const int n_windows = __builtin_amdgcn_readfirstlane(stream_stride_bytes / 8 - 2);
        unsigned long long dead_mask[G];
        int left, early;
        if constexpr (FORM == 2 && G == 1)
            banded_funnel64s_rows_asm_g1(st, M, voff, base, uniform_u64(s), n_windows, band, static_cast<uint32_t>(band64 >> 32), cut_rows, limit, push_row, push_row_solid, solid_limit, push_max, dead_mask, left, early);
        else if constexpr (FORM == 2)
            banded_funnel64_rows_asm_g2(st, M, voff, base, uniform_u64(s), n_windows, band, static_cast<uint32_t>(band64 >> 32), cut_rows, limit, push_row, push_row_solid, solid_limit, push_max, dead_mask, left, early);
        else if constexpr (FORM == 1)
            banded_funnel32_rows_asm_g2(st, M, voff, base, uniform_u64(s), n_windows, band, cut_rows, limit, push_row, push_row_solid, solid_limit, push_max, dead_mask, left, early);
        else if constexpr (G == 2)
            banded_cut_rows_asm_g2(st, M, voff, base, uniform_u64(s), n_windows, band, cut_rows, limit, push_row, push_row_solid, solid_limit, push_max, dead_mask, left, early);
        else
            banded_cut_rows_asm_g1(st, M, voff, base, uniform_u64(s), n_windows, band, cut_rows, limit, push_row, push_row_solid, solid_limit, push_max, dead_mask, left, early);
        note_stream_fault(fault_word, left);
        unsigned lane_s = static_cast<unsigned>(lane);          // the lane's offset in the stores below, formed here, not kept
        if constexpr (DYN) asm volatile("" : "+v"(lane_s));
        if (early) {
            // few lanes within the limit at a late test: they wait in the regroup list, the others are rejected here
#pragma unroll
            for (int gg = 0; gg < G; gg++) {
                if (!has[gg]) continue;
                const unsigned long long alive = ~dead_mask[gg];
                if ((dead_mask[gg] >> lane) & 1ull)
                    dst[static_cast<size_t>(q) * ld + gg * kLanes + lane_s] = static_cast<int8_t>(HIP_MAX_ERROR);
                else
                    regroup[n_regroup + __popcll(alive & ((1ull << lane) - 1ull))] =
                        (static_cast<uint32_t>(q - q0) << 8) | (static_cast<uint32_t>(gg) << 6) | lane;
                n_regroup += __builtin_amdgcn_readfirstlane(static_cast<int>(__popcll(alive)));
            }
            continue;
        }
#pragma unroll
        for (int gg = 0; gg < G; gg++) {
            if (!has[gg]) continue;
            const bool dead = (dead_mask[gg] >> lane) & 1ull;
            int8_t result = static_cast<int8_t>(HIP_MAX_ERROR);
            if (dead_mask[gg] != ~0ull) {
                // :230-245 — walk the last row across the band, keep the minimum.
                unsigned long long vp = st[NS * gg], vn = st[NS * gg + (FORM == 2 ? 2 : 1)];
                if constexpr (FORM == 2) {
                    vp |= static_cast<unsigned long long>(st[NS * gg + 1]) << 32;
                    vn |= static_cast<unsigned long long>(st[NS * gg + 3]) << 32;
                }
                uint32_t err = static_cast<uint32_t>(k) + st[NS * gg + NS - 1], best = err;
                for (int i = 0; i <= h; i++) {
                    err += static_cast<uint32_t>((vp >> i) & 1ull);
                    err -= static_cast<uint32_t>((vn >> i) & 1ull);
                    best = err < best ? err : best;
                }
                if (!dead) result = static_cast<int8_t>(best);
            }
            dst[static_cast<size_t>(q) * ld + gg * kLanes + lane_s] = result;
        }
    }
    if constexpr (DYN) task = resolve_wave_task(task_issued);
    } while (DYN && task < n_tasks);
}

// ---- regrouping of sparse survivors -------------------------------------------------------------------
// The filter exists to reject: on realistic inputs a wave of 64 subjects holds at most a few pairs that
// survive, and running all 64 lanes to the last row for them is what made sparse survivors expensive
// (1 % of the pairs surviving, scattered over the subjects: 47 % of the waves ran to the end — 2.6x the time
// of random pairs).  A wave that finds 1..push_max lanes alive at a late test therefore stops and puts them
// on a list (LDS, per wave); when the list is nearly full or the wave's query tile is done, the listed pairs
// are scored one per lane, densely, from row 0 (banded_asm_kernel, top of the query loop).  Whatever gets
// onto the list is scored exactly, so where the first pass stops is a pure performance choice.
#if BGSA_AB_KERNELS
// ---- straight-line rows (k <= 15): banded_chunk_rows_asm32 ---------------------------------------------
// Same task decomposition, tests, regrouping and final walk as banded_asm_kernel; what differs is how a row gets
// its match words: the wave keeps them in LDS ([class][slot][lane] dwords, 3840 B per wave) and a row's token is the
// byte offset of its class, so the 32 rows of a chunk are straight-line code with immediate shift amounts and no
// scalar work per row (gen_rows_asm.py: gen_banded_chunk_function).  Tokens: one dword per row, 32 per chunk, packed
// per launch by pack_banded_tokens_kernel.
__global__ __launch_bounds__(256) void banded_chunk_kernel(
    const uint32_t *__restrict__ tokens, const uint32_t *__restrict__ mext, int8_t *__restrict__ out,
    long long ld, int n_groups, int word_num, int n_queries, int q_tile, int k, int token_stride_bytes,
    const char *__restrict__ content, int ref_start, int len, uint32_t push_row, uint32_t push_max)
{
    constexpr int NM = 3;
    __shared__ uint32_t s_words[kWavesPerBlock][kChars][NM][kLanes];
    __shared__ uint32_t s_regroup[kWavesPerBlock][kLanes];
    const int lane = threadIdx.x & (kLanes - 1);
    const int wave = threadIdx.x >> 6;
    const int group = __builtin_amdgcn_readfirstlane(blockIdx.x * kWavesPerBlock + wave);
    if (group >= n_groups) return;
    uint32_t *regroup = s_regroup[wave];
    int n_regroup = 0;
    const uint32_t *g = mext + static_cast<size_t>(group) * kChars * word_num * kLanes;
    const unsigned long long gbase = uniform_u64(reinterpret_cast<unsigned long long>(g));
    // LDS byte address of this lane's dword in the wave's block (LDS addresses are 32-bit offsets)
    const uint32_t lanebase = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(&s_words[wave][0][0][lane]));

    uint32_t first[kChars][NM];
#pragma unroll
    for (int c = 0; c < kChars; c++)
#pragma unroll
        for (int w = 0; w < NM; w++) first[c][w] = g[(c * word_num + w) * kLanes + lane];
    const int h = k;
    const uint32_t band = static_cast<uint32_t>((1ull << (k + h + 1)) - 1ull);
    const uint32_t limit = static_cast<uint32_t>(h + 1);
    const uint32_t last_check = static_cast<uint32_t>((len <= 64) ? len : ((len - h > 64) ? len - h : 64));

    const int q0 = blockIdx.y * q_tile;
    const int q1 = (q0 + q_tile < n_queries) ? q0 + q_tile : n_queries;
    int8_t *dst = out + static_cast<size_t>(group) * kLanes + lane;

    for (int q = q0; q <= q1; q++) {
        if (n_regroup > kLanes - static_cast<int>(push_max) || (q == q1 && n_regroup > 0)) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (lane < n_regroup)
                banded_finish_pair<uint32_t>(regroup[lane], g, content, ref_start + q0, len, word_num, k,
                                             out + static_cast<size_t>(q0) * ld + static_cast<size_t>(group) * kLanes, ld);
            __builtin_amdgcn_wave_barrier();
            n_regroup = 0;
        }
        if (q == q1) break;
        uint32_t st[3] = {0u, 0u, 0u};
#pragma unroll
        for (int c = 0; c < kChars; c++)
#pragma unroll
            for (int w = 0; w < NM; w++) s_words[wave][c][w][lane] = first[c][w];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        uint32_t voff[kChars];
#pragma unroll
        for (int c = 0; c < kChars; c++) voff[c] = static_cast<uint32_t>(((c * word_num + NM) * kLanes + lane) * 4);
        int early;
        const unsigned long long dead_mask = banded_chunk_rows_asm32(
            st, voff, lanebase, uniform_u64(reinterpret_cast<unsigned long long>(tokens)),
            __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(q) * static_cast<uint32_t>(token_stride_bytes)), gbase, band, limit,
            static_cast<uint32_t>(k), last_check, static_cast<uint32_t>(len), push_row, push_max, early);
        __builtin_amdgcn_wave_barrier();
        const bool dead = (dead_mask >> lane) & 1ull;
        if (early) {
            const unsigned long long alive = ~dead_mask;
            if (dead)
                dst[static_cast<size_t>(q) * ld] = static_cast<int8_t>(HIP_MAX_ERROR);
            else
                regroup[n_regroup + __popcll(alive & ((1ull << lane) - 1ull))] = (static_cast<uint32_t>(q - q0) << 8) | lane;
            n_regroup += __builtin_amdgcn_readfirstlane(static_cast<int>(__popcll(alive)));
            continue;
        }
        int8_t result = static_cast<int8_t>(HIP_MAX_ERROR);
        if (dead_mask != ~0ull) {
            uint32_t err = static_cast<uint32_t>(k) + st[2], best = err;
            for (int i = 0; i <= h; i++) {
                err += (st[0] >> i) & 1u;
                err -= (st[1] >> i) & 1u;
                best = err < best ? err : best;
            }
            if (!dead) result = static_cast<int8_t>(best);
        }
        dst[static_cast<size_t>(q) * ld] = result;
    }
}

// Row tokens of banded_chunk_kernel: token[q][r] = class * 768 (the class's byte offset in the wave's LDS block),
// zero beyond the query's end up to whole chunks of 32 rows.
__global__ __launch_bounds__(256) void pack_banded_tokens_kernel(const char *__restrict__ content, uint32_t *__restrict__ tokens,
                                                                 int len, int ref_start, int n_queries, int rows_padded)
{
    const long long tid = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (tid >= static_cast<long long>(n_queries) * rows_padded) return;
    const int q = static_cast<int>(tid / rows_padded), r = static_cast<int>(tid % rows_padded);
    uint32_t c = 0;
    if (r < len) {
        c = static_cast<unsigned char>(content[static_cast<size_t>(ref_start + q) * (len + 1) + r]);
        if (c > 4u) c = 0u;   // as the stream packers: out-of-alphabet bytes behave as 'A'
    }
    tokens[tid] = c * 768u;
}
#endif  // BGSA_AB_KERNELS

namespace {

// Regrouping policy (measurement knobs; any value gives the same scores): a wave puts its survivors on the
// list when a test at or after row k + BGSA_BANDED_PUSH_ROW (default 48: random pairs are dead by then)
// finds at most BGSA_BANDED_PUSH_MAX (default 8, at most 32, 0 = never) lanes within the limit.
int banded_push_max()
{
    static const int v = [] {
        const char *e = getenv("BGSA_BANDED_PUSH_MAX");
        const int x = e ? atoi(e) : 8;
        return (x >= 0 && x <= 32) ? x : 8;
    }();
    return v;
}
int banded_push_solid_offset()
{
    static const int v = [] {
        const char *e = getenv("BGSA_BANDED_PUSH_SOLID");
        const int x = e ? atoi(e) : 32;
        return x >= 0 ? x : 32;
    }();
    return v;
}
int banded_push_solid_margin()
{
    static const int v = [] {
        const char *e = getenv("BGSA_BANDED_SOLID_MARGIN");   // unset: (k + 2) / 2
        return e ? atoi(e) : -1;
    }();
    return v;
}
int banded_push_row_offset()
{
    static const int v = [] {
        const char *e = getenv("BGSA_BANDED_PUSH_ROW");
        const int x = e ? atoi(e) : 48;
        return x >= 0 ? x : 48;
    }();
    return v;
}

// 0 = generated asm, threaded row loop, sliding band (default), 3 = the same loop with the band held in place for
// k <= 11 (BGSA_BANDED_IMPL=p: measured 4-17 % slower, see rows_ir.py: banded_phase_body), 1 = compiler-scheduled C++ kernel (BGSA_BANDED_IMPL=c),
// 2 = generated asm, straight-line chunk rows for k <= 15 (BGSA_BANDED_IMPL=s).  The straight-line kernel removes the
// scalar work per row (7.85 SALU per wave-row in the threaded loop, on a scalar unit shared by the CU's four SIMDs) at
// the price of one more VALU, two LDS reads per row and a slot rotation every 32 rows; measured on 10k x 1M x 150 bp,
// k = 8, same box, threaded / straight-line: random pairs 111.6 / 117.7 ms, 1 % dense survivors 171.8 / 163.7,
// every pair surviving 384.2 / 393.5 — no better where it matters, so the threaded loop stays the default and this
// one the measured alternative.
int banded_impl()
{
    static const int impl = [] {
        const char *e = getenv("BGSA_BANDED_IMPL");
        return (e && e[0] == 'c') ? 1 : ((e && e[0] == 's') ? 2 : ((e && e[0] == 'p') ? 3 : ((e && e[0] == 'a') ? 4 : 0)));
    }();
    return impl;
}

// Dynamic task handout for the one-word-window kernels.  Until round 4 the task loop cost the two-group kernel a wave per
// SIMD (89 instead of 73 VGPRs: 290.0 -> 306.6 ms with every pair surviving) and the counter was an off-by-default knob; with the
// loop's invariants laundered out of the VGPRs (banded_cut_kernel: 75) it is the default: same box, static grid -> counter,
// 10k x 1M x 150 bp, k = 8: planted mix 85.6 -> 84.8 ms, 1 % of all pairs surviving 106.2 -> 102.1, every pair surviving
// 276.5 -> 272.9 (scripts/r04_dyn_ab.sh, profiles/r04_dyn_ab.txt).  BGSA_BANDED_DYNAMIC=0 restores the static grid.
bool banded_dynamic_tasks()
{
    static const bool on = [] {
        const char *e = getenv("BGSA_BANDED_DYNAMIC");
        return !(e && e[0] == '0') && dynamic_tasks();
    }();
    return on;
}

// Subject groups per wave of the one-word-window kernel (BGSA_BANDED_GROUPS, 1 or 2; default 2).
int banded_groups()
{
    static const int v = [] {
        const char *e = getenv("BGSA_BANDED_GROUPS");
        return (e && e[0] == '1') ? 1 : 2;
    }();
    return v;
}

#if BGSA_AB_KERNELS
int launch_chunk(const char *d_content, const uint32_t *d_peq, int8_t *d_results, int len, int64_t read_count,
                 int ref_start, int ref_end, int word_num, int k, void *d_workspace, hipStream_t stream)
{
    const int nq = ref_end - ref_start;
    const int64_t n_groups = read_count / kLanes;
    const int q_tile = pick_query_tile(nq, n_groups, len, 32);
    note_query_tile(q_tile);
    dim3 grid(static_cast<unsigned>((n_groups + kWavesPerBlock - 1) / kWavesPerBlock),
              static_cast<unsigned>((nq + q_tile - 1) / q_tile));
    if (grid.y > 65535u) {
        set_error_text("banded: too many query tiles for one launch");
        return BGSA_HIP_EUNSUPPORTED;
    }
    const int rows_padded = (len + 31) / 32 * 32;
    const long long total = static_cast<long long>(nq) * rows_padded;
    hipLaunchKernelGGL(pack_banded_tokens_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, stream,
                       d_content, static_cast<uint32_t *>(d_workspace), len, ref_start, nq, rows_padded);
    BGSA_HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(banded_chunk_kernel, grid, dim3(256), 0, stream, static_cast<const uint32_t *>(d_workspace), d_peq,
                       d_results, static_cast<long long>(read_count), static_cast<int>(n_groups), word_num, nq, q_tile, k,
                       rows_padded * 4, d_content, ref_start, len, static_cast<uint32_t>(k + banded_push_row_offset()),
                       static_cast<uint32_t>(banded_push_max()));
    BGSA_HIP_TRY(hipGetLastError());
    return BGSA_HIP_OK;
}
#endif  // BGSA_AB_KERNELS

int launch_asm(const char *d_content, const uint32_t *d_peq, int8_t *d_results, int len, int64_t read_count,
               int ref_start, int ref_end, int word_num, int k, void *d_workspace, hipStream_t stream)
{
    const int nq = ref_end - ref_start;
    const int64_t n_groups = read_count / kLanes;
    const int phase = banded_stream_phase(k), cut = banded_stream_cut(k);
    // the two-group loop around the funnel-shift rows (k >= 13): the default since round 4; BGSA_BANDED_IMPL=a keeps round 2's
    // one-group loops (banded_asm_kernel) at every k, BGSA_BANDED_GROUPS=1 at k >= 13
    // measured (scripts/r04_banded_funnel_ab.sh, r04_banded_pair_loop.sh; every pair surviving): the 32-bit rows gain 8 % from the
    // two-group loop (random pairs 14-24 %); the 64-bit pair LOSES 4.5 % with two groups, loses 1.4 % with one group on a static
    // grid and gains 2.2 % with one group on the task counter (k = 31: 660.8 -> 646.4 ms): that is its default.
    // BGSA_BANDED_PAIR_LOOP=0: round 2's loop; =2: two groups (A/B flavour of the library only)
    static const int pair_loop = [] { const char *e = getenv("BGSA_BANDED_PAIR_LOOP"); return e ? atoi(e) : 1; }();
    int form = -1, G = 1;
    if (cut > 0) { form = 0; G = banded_groups(); }
    else if (phase == 0 && banded_impl() == 0 && banded_groups() == 2 && k <= 15) { form = 1; G = 2; }
    else if (phase == 0 && banded_impl() == 0 && k > 15 && (pair_loop == 1 || pair_loop == 2)) { form = 2; G = pair_loop; }
#if !BGSA_AB_KERNELS
    if (form == 2 && G == 2) return ab_knob_refused("BGSA_BANDED_PAIR_LOOP=2");
#endif
    const int64_t n_waves = (n_groups + G - 1) / G;
    const int q_tile = pick_query_tile(nq, n_waves, static_cast<long long>(len) * G, 32);
    if (int rc = launch_pack_banded(d_content, len, k, phase, cut, ref_start, ref_end, d_workspace, stream)) return rc;
    const int stride = banded_stream_layout(len, k, phase, cut, nullptr, nullptr);
    unsigned *fault = nullptr;
    if (int rc = stream_guard(d_workspace, stride, kBandedRefill, 40, stream, &fault)) return rc;

    note_query_tile(q_tile);
    dim3 grid(static_cast<unsigned>((n_waves + kWavesPerBlock - 1) / kWavesPerBlock),
              static_cast<unsigned>((nq + q_tile - 1) / q_tile));
    if (grid.y > 65535u) {
        set_error_text("banded: too many query tiles for one launch");
        return BGSA_HIP_EUNSUPPORTED;
    }
    const uint32_t push_row = static_cast<uint32_t>(k + banded_push_row_offset());
    const uint32_t push_max = static_cast<uint32_t>(banded_push_max());
    // one-word-window kernels: already from this row on if an alive lane is a solid survivor (gen_rows_asm.py: push_or)
    const uint32_t push_row_solid = std::min(push_row, static_cast<uint32_t>(k + banded_push_solid_offset()));
    // a solid survivor: at most this many errors since row k — by default about half the limit of k + 1 (swept on
    // 10k x 1M, k = 8, scripts/r03_solid.sh: rows k + 16 ... k + 32 with margins 4 ... 6 all give 128-129 ms on the 1 %
    // mix and 88 ms on random pairs; with margin 2 random stragglers pass for survivors: 107 ms from row k + 32)
    const int margin = banded_push_solid_margin() >= 0 ? banded_push_solid_margin() : (k + 2) / 2;
    const uint32_t solid_limit = static_cast<uint32_t>(k + 1 > margin ? k + 1 - margin : 0);
    if (cut > 0 || form > 0) {
        unsigned *counter = nullptr;
        const long long blocks = static_cast<long long>(grid.x) * grid.y;
        if (banded_dynamic_tasks() && dynamic_tasks_fit(blocks * kWavesPerBlock)) {
            counter = task_counter_in(d_workspace, static_cast<size_t>(stride) * nq);
            BGSA_HIP_TRY(hipMemsetAsync(counter, 0, 8, stream));
#if BGSA_AB_KERNELS
            const int resident = (form == 2 && G == 2) ? persistent_blocks_for(banded_cut_kernel<2, true, 2>)
                               : form == 2 ? persistent_blocks_for(banded_cut_kernel<1, true, 2>)
#else
            const int resident = form == 2 ? persistent_blocks_for(banded_cut_kernel<1, true, 2>)
#endif
                               : form == 1 ? persistent_blocks_for(banded_cut_kernel<2, true, 1>)
                               : G == 2 ? persistent_blocks_for(banded_cut_kernel<2, true>) : persistent_blocks_for(banded_cut_kernel<1, true>);
            grid = dim3(static_cast<unsigned>(blocks < resident ? blocks : resident), 1u);
        }
#if BGSA_AB_KERNELS
        auto kernel = (form == 2 && G == 2) ? (counter ? banded_cut_kernel<2, true, 2> : banded_cut_kernel<2, false, 2>)
                    : form == 2 ? (counter ? banded_cut_kernel<1, true, 2> : banded_cut_kernel<1, false, 2>)
#else
        auto kernel = form == 2 ? (counter ? banded_cut_kernel<1, true, 2> : banded_cut_kernel<1, false, 2>)
#endif
                    : form == 1 ? (counter ? banded_cut_kernel<2, true, 1> : banded_cut_kernel<2, false, 1>)
                    : G == 2 ? (counter ? banded_cut_kernel<2, true> : banded_cut_kernel<2, false>)
                             : (counter ? banded_cut_kernel<1, true> : banded_cut_kernel<1, false>);
        static const unsigned lds_pad = [] { const char *e = getenv("BGSA_BANDED_LDS_PAD"); return e ? static_cast<unsigned>(atoi(e)) : 0u; }();
        hipLaunchKernelGGL(kernel, grid, dim3(256), lds_pad, stream,
                           static_cast<const unsigned char *>(d_workspace), d_peq, d_results,
                           static_cast<long long>(read_count), static_cast<int>(n_groups), word_num, nq, q_tile, k, stride,
                           fault, d_content, ref_start, len, push_row, push_row_solid, solid_limit, push_max, static_cast<uint32_t>(cut),
                           counter);
    }
#if BGSA_AB_KERNELS
    else if (phase > 0)
        hipLaunchKernelGGL((banded_asm_kernel<false, true>), grid, dim3(256), 0, stream,
                           static_cast<const unsigned char *>(d_workspace), d_peq, d_results,
                           static_cast<long long>(read_count), static_cast<int>(n_groups), word_num, nq, q_tile, k, stride,
                           fault, d_content, ref_start, len, push_row, push_max, phase);
#endif
    else if (k <= 15)
        hipLaunchKernelGGL((banded_asm_kernel<false, false>), grid, dim3(256), 0, stream,
                           static_cast<const unsigned char *>(d_workspace), d_peq, d_results,
                           static_cast<long long>(read_count), static_cast<int>(n_groups), word_num, nq, q_tile, k, stride,
                           fault, d_content, ref_start, len, push_row, push_max, 0);
    else
        hipLaunchKernelGGL((banded_asm_kernel<true, false>), grid, dim3(256), 0, stream,
                           static_cast<const unsigned char *>(d_workspace), d_peq, d_results,
                           static_cast<long long>(read_count), static_cast<int>(n_groups), word_num, nq, q_tile, k, stride,
                           fault, d_content, ref_start, len, push_row, push_max, 0);
    BGSA_HIP_TRY(hipGetLastError());
    return BGSA_HIP_OK;
}

#if BGSA_AB_KERNELS
template <typename T>
int launch_t(const char *d_content, const uint32_t *d_peq, int8_t *d_results, int len,
              int64_t read_count, int ref_start, int ref_end, int word_num, int k, hipStream_t stream)
{
    const int nq = ref_end - ref_start;
    const int64_t n_groups = read_count / kLanes;
    const int q_tile = pick_query_tile(nq, n_groups, len, 32);
    note_query_tile(q_tile);
    dim3 grid(static_cast<unsigned>((n_groups + kWavesPerBlock - 1) / kWavesPerBlock),
              static_cast<unsigned>((nq + q_tile - 1) / q_tile));
    if (grid.y > 65535u) {
        set_error_text("banded: too many query tiles for one launch");
        return BGSA_HIP_EUNSUPPORTED;
    }
    hipLaunchKernelGGL((banded_kernel<T>), grid, dim3(256), 0, stream, d_content, d_peq, d_results, len,
                       static_cast<long long>(read_count), static_cast<int>(n_groups), word_num, ref_start,
                       ref_end, q_tile, k);
    BGSA_HIP_TRY(hipGetLastError());
    return BGSA_HIP_OK;
}
#endif  // BGSA_AB_KERNELS

}  // namespace

static thread_local int g_last_k = 8;  // threshold of this thread's last banded launch

int banded_stream_phase(int k)
{
    return (banded_impl() == 3 && k >= 1 && k <= 15) ? banded_phase_rows(k) : 0;
}

int banded_stream_cut(int k)
{
    return (banded_impl() == 0 && k >= 1) ? banded_cut_rows(k) : 0;
}

const char *banded_kernel_name(int word_num)
{
    (void)word_num;
    if (banded_impl() == 1) return g_last_k <= 15 ? "banded_kernel<uint32_t>" : "banded_kernel<uint64_t>";
    if (banded_impl() == 2 && g_last_k <= 15) return "banded_chunk_kernel";
    if (banded_stream_phase(g_last_k) > 0) return "banded_asm_kernel<false, true>";
    if (banded_stream_cut(g_last_k) > 0) return banded_groups() == 2 ? "banded_cut_kernel<2>" : "banded_cut_kernel<1>";
    if (banded_impl() == 0 && banded_groups() == 2 && g_last_k <= 15) return "banded_cut_kernel<2, funnel32>";
    if (banded_impl() == 0 && g_last_k > 15) {
        static const int pair_loop = [] { const char *e = getenv("BGSA_BANDED_PAIR_LOOP"); return e ? atoi(e) : 1; }();
        if (pair_loop == 1) return "banded_cut_kernel<1, funnel64>";
        if (pair_loop == 2) return "banded_cut_kernel<2, funnel64>";
    }
    return g_last_k <= 15 ? "banded_asm_kernel<false, false>" : "banded_asm_kernel<true, false>";
}

int launch_banded(const char *d_content, const uint32_t *d_peq, int8_t *d_results, int ref_len,
                  int read_len, int64_t read_count, int ref_start, int ref_end, int word_num, int k,
                  void *d_workspace, hipStream_t stream)
{
    if (ref_end <= ref_start || read_count == 0) return BGSA_HIP_OK;
    if (ref_len != read_len) {
        set_error_text("banded: query_len must equal subject_len (the reference's band is mis-aligned otherwise)");
        return BGSA_HIP_EUNSUPPORTED;
    }
    if (k < 1 || k > 31 || 2 * k + 1 >= read_len) {
        set_error_text("banded: threshold must satisfy 1 <= k <= 31 and 2k+1 < length");
        return BGSA_HIP_EUNSUPPORTED;
    }
    g_last_k = k;
#if !BGSA_AB_KERNELS
    if (banded_impl() >= 1 && banded_impl() <= 3) return ab_knob_refused("BGSA_BANDED_IMPL=c/s/p");
    return launch_asm(d_content, d_peq, d_results, read_len, read_count, ref_start, ref_end, word_num, k, d_workspace, stream);
#else
    if (banded_impl() == 2 && k <= 15)
        return launch_chunk(d_content, d_peq, d_results, read_len, read_count, ref_start, ref_end, word_num, k,
                            d_workspace, stream);
    if (banded_impl() != 1)   // 0, 3 and 4: the threaded loops
        return launch_asm(d_content, d_peq, d_results, read_len, read_count, ref_start, ref_end, word_num, k,
                          d_workspace, stream);
    if (k <= 15)
        return launch_t<uint32_t>(d_content, d_peq, d_results, read_len, read_count, ref_start, ref_end,
                                  word_num, k, stream);
    return launch_t<uint64_t>(d_content, d_peq, d_results, read_len, read_count, ref_start, ref_end,
                              word_num, k, stream);
#endif
}

}  // namespace bgsa
