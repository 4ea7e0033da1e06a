// bgsa_common.h — shared host/device helpers of the gfx950 backend (internal, not installed).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <stdio.h>

#include "../../include/bgsa_hip.h"

namespace bgsa {

constexpr int kLanes = HIP_V_NUM;      // one subject per lane of a 64-wide wavefront
constexpr int kChars = BGSA_CHAR_NUM;  // A C G T N
constexpr int kWavesPerBlock = 4;      // 256-thread workgroups: one wave per SIMD of a CU
constexpr int kMaxWords = 32;          // longest subject kept entirely in VGPRs: 32 x 32 = 1024 bp

void set_error(const char *what, hipError_t e, const char *file, int line);
void set_error_text(const char *text);

#define BGSA_HIP_TRY(expr)                                              \
    do {                                                                \
        hipError_t e_ = (expr);                                         \
        if (e_ != hipSuccess) {                                         \
            ::bgsa::set_error(#expr, e_, __FILE__, __LINE__);           \
            return BGSA_HIP_EHIP;                                       \
        }                                                               \
    } while (0)

// Launchers implemented in the per-algorithm .hip files.  All pointers are device pointers.
int launch_myers(const char *d_content, const uint32_t *d_peq, int16_t *d_results, int ref_len,
                 int read_len, int64_t read_count, int ref_start, int ref_end, int word_num,
                 void *d_workspace, hipStream_t stream, int semi_global = 0);
const char *myers_kernel_name(int word_num, int semi_global = 0);

// Packed query stream (one per query, 8-byte windows): codes 0..4 = A C G T N row, 5 = END,
// 6 = REFILL.  Window i < n_windows-1 holds characters 7i..7i+6 and a REFILL; the last window
// holds the remaining characters, an END, and END padding; one spare all-END window follows
// (the kernels fetch one window ahead).
constexpr int kCodeEnd = 5, kCodeRefill = 6;
inline int stream_windows(int ref_len) { return ref_len / 7 + 1; }
inline size_t stream_stride(int ref_len) { return static_cast<size_t>(stream_windows(ref_len) + 1) * 8; }
// Window `i` of the plain stream of one query (row = its mapped characters); i == n_windows is the
// spare all-END window.  Shared by the packer kernel and the host-side introspection entry point.
__host__ __device__ inline unsigned long long plain_stream_window(const char *row, int ref_len, int i)
{
    const int n_windows = ref_len / 7 + 1;
    unsigned long long win = 0;
    for (int j = 0; j < 8; j++) {
        unsigned code = kCodeEnd;
        if (i < n_windows) {
            const int pos = 7 * i + j;
            if (j == 7) code = (i < n_windows - 1) ? kCodeRefill : kCodeEnd;
            else if (pos < ref_len) {
                code = static_cast<unsigned char>(row[pos]);
                if (code > 4) code = 0;  // the reference would index Peq out of range; treat as 'A'
            }
        }
        win |= static_cast<unsigned long long>(code) << (8 * j);
    }
    return win;
}
// d_zero_word: a device word the packer also sets to 0 (the scoring launch's task counter: no memset
// of its own between the two kernels), or null.
int launch_pack_queries(const char *d_content, int ref_len, int ref_start, int ref_end,
                        void *d_streams, hipStream_t stream, unsigned *d_zero_word = nullptr);

// Two-rows-per-token stream of the short-subject Myers kernels (<= 64 bp): token t covers query
// characters 2t and 2t+1 as 5*a + b (0..24), an odd last character travels alone as 25 + c; 30 = END,
// 31 = REFILL; 7 tokens + REFILL per window, one spare all-END window, as in the plain stream.
constexpr int kPairSingle = 25, kPairEnd = 30, kPairRefill = 31;
inline int pair_stream_windows(int ref_len) { return ((ref_len + 1) / 2) / 7 + 1; }
inline size_t pair_stream_stride(int ref_len) { return static_cast<size_t>(pair_stream_windows(ref_len) + 1) * 8; }
__host__ __device__ inline unsigned long long pair_stream_window(const char *row, int ref_len, int i)
{
    const int n_tokens = (ref_len + 1) / 2;
    const int n_windows = n_tokens / 7 + 1;
    auto code_of = [&](int r) -> unsigned {
        const unsigned c = static_cast<unsigned char>(row[r]);
        return c > 4 ? 0u : c;   // as plain_stream_window: out-of-alphabet bytes behave as 'A'
    };
    unsigned long long win = 0;
    for (int j = 0; j < 8; j++) {
        unsigned code = kPairEnd;
        if (i < n_windows) {
            const int t = 7 * i + j;
            if (j == 7) code = (i < n_windows - 1) ? kPairRefill : kPairEnd;
            else if (t < n_tokens)
                code = (2 * t + 1 < ref_len) ? 5u * code_of(2 * t) + code_of(2 * t + 1) : kPairSingle + code_of(2 * t);
        }
        win |= static_cast<unsigned long long>(code) << (8 * j);
    }
    return win;
}
int launch_pack_query_pairs(const char *d_content, int ref_len, int ref_start, int ref_end,
                            void *d_streams, hipStream_t stream, unsigned *d_zero_word = nullptr);

// Banded stream (rows_ir.py: banded_tokens / banded_stream_codes).  The banded row is short (12 VALU),
// so the scalar work of the threaded-code dispatch is what its loop waits for; one token therefore
// carries TWO consecutive rows whenever nothing has to happen between them:
//   0..24   two rows, classes a then b: 5*a + b        25..29  one row of class c: 25 + c
//   30 END  31 REFILL  63 EVENT, followed by an argument byte of bits: 1 = reset the error count
//   (row k), 2 = advance the match-string words (every 32 rows), 4 = test the limit on all lanes,
//   8 = latch the reject mask (the reference's last checkpoint).
// (EVENT is the last of the 64 slots the dispatch's 6-bit mask can reach, so its code may be longer than a
// slot; 32..62 are fail slots: gen_rows_asm.py.)
// A two-byte token never straddles a window: the window is closed early with a REFILL instead.
// Writes the stream when dst != nullptr (row = mapped query characters); returns its length in bytes
// including the spare window.  The same routine sizes the workspace on the host and fills it on the
// device.
constexpr int kBandedSingle = kPairSingle, kBandedEnd = kPairEnd, kBandedRefill = kPairRefill, kBandedEvent = 63;
// Upper bound of banded_stream_layout(len, k) over every k (sizes the workspace, which does not know k):
// all rows as one-row tokens, every event two bytes plus one byte lost to an early window close.
// Rows between two tests of the error limit (a wave stops as soon as all 64 lanes are past it).  The
// reference tests every 16 rows (banded/BGSA_CPU/config.h: batch_size); the errors never decrease, so
// testing more often rejects exactly the same pairs, only sooner.
constexpr int kBandedCheckRows = 8;
constexpr int kBandedLateRows = 48;   // from row k + 48 on the tests come every 16 rows
// Rows a 32-bit word can hold the band of threshold k in place (rows_ir.py: banded_phase_body): its 2k + 1 bits start
// at bit 0 and move up one bit per row; the bit above them must still exist in the phase's last row.  0: no room worth
// a phase (k > 11) — those thresholds run the sliding form.
constexpr int kBandedPhaseMin = 8;
__host__ __device__ inline int banded_phase_rows(int k)
{
    const int rows = 31 - 2 * k;
    return rows >= kBandedPhaseMin ? rows : 0;
}
int banded_stream_phase(int k);   // banded.hip: the phase length the kernel launched for threshold k uses (0: sliding form)
// Rows between two cuts of the one-word window form (rows_ir.py: banded_cut_body): the band's 2k + 1 bits, offset by up
// to rows - 1 bits, must fit one 32-bit word.  0: no room worth it (k > 12) — those thresholds keep the funnel-shift row.
__host__ __device__ inline int banded_cut_rows(int k) { return k <= 8 ? 16 : (k <= 12 ? 8 : 0); }
int banded_stream_cut(int k);     // banded.hip: the cut length the kernel launched for threshold k uses (0: no cut events)
inline size_t banded_stream_bound(int len)
{
    const size_t events = static_cast<size_t>(len) / kBandedCheckRows + static_cast<size_t>(len) / 32 +
                          static_cast<size_t>(len) / kBandedPhaseMin + 4;
    const size_t threaded = (static_cast<size_t>(len) + 3 * events + 1 + 6) / 7 * 8 + 16;
    const size_t chunk_tokens = (static_cast<size_t>(len) + 31) / 32 * 32 * 4;   // banded_chunk_kernel: one dword per row
    return threaded > chunk_tokens ? threaded : chunk_tokens;
}
// EVENT bits: 1 = scoring starts (row k), 2 = next 32 rows (the match-string words move down), 4 = test the error
// limit, 8 = latch the reject mask (the reference's last checkpoint), 16 = re-anchor the band (every `phase` rows;
// phase = 0: the sliding form, no such event), 32 = cut the next one-word window (every `cut` rows that do not also
// advance the words; cut = 0: the funnel-shift row, no such event).
__host__ __device__ inline int banded_stream_layout(int len, int k, int phase, int cut, const char *row, unsigned char *dst)
{
    const int last = (len <= 64) ? len : ((len - k > 64) ? len - k : 64);
    int pos = 0, slot = 0, pending = 0;
    auto put = [&](unsigned char b) {
        if (dst) dst[pos] = b;
        pos++;
        if (++slot == 7) {
            if (dst) dst[pos] = kBandedRefill;
            pos++;
            slot = 0;
        }
    };
    auto put_event = [&](int bits) {
        if (slot == 6) {  // one payload byte left: close the window
            if (dst) { dst[pos] = kBandedRefill; dst[pos + 1] = kBandedEnd; }
            pos += 2;
            slot = 0;
        }
        put(kBandedEvent);
        put(static_cast<unsigned char>(bits));
    };
    auto code_of = [&](int r) -> int {
        if (!row) return 0;
        const int c = static_cast<unsigned char>(row[r]);
        return c > 4 ? 0 : c;
    };
    // events due before row r starts / after `done` rows are complete
    auto before = [&](int r) {
        return (r == k ? 1 : 0) | ((r > 0 && (r & 31) == 0) ? 2 : 0) | ((phase > 0 && r > 0 && r % phase == 0) ? 16 : 0) |
               ((cut > 0 && r > 0 && r % cut == 0 && (r & 31) != 0) ? 32 : 0);
    };
    // Tests between checkpoints only decide how soon a wave may stop, never the result (the errors are monotone and
    // the reject mask is latched at `last`): none before a lane can be past the limit at all (more than k + 1 errors
    // need more than k + 1 scored rows), every kBandedCheckRows rows while random pairs are dying, every
    // 2 * kBandedCheckRows once they are dead (k + kBandedLateRows rows on) — what is still alive then mostly stays.
    auto after = [&](int done) {
        if (!(done > k && done <= last)) return 0;
        if (done == last) return 4 | 8;
        if ((done & (kBandedCheckRows - 1)) != 0 || done - k <= k + 1) return 0;
        if (done > k + kBandedLateRows && (done & (2 * kBandedCheckRows - 1)) != 0) return 0;
        return 4;
    };
    for (int r = 0; r < len;) {
        pending |= before(r);
        if (pending) { put_event(pending); pending = 0; }
        if (r + 1 < len && !after(r + 1) && !before(r + 1)) {   // nothing between rows r and r+1
            put(static_cast<unsigned char>(5 * code_of(r) + code_of(r + 1)));
            r += 2;
        } else {
            put(static_cast<unsigned char>(kBandedSingle + code_of(r)));
            r += 1;
        }
        pending |= after(r);
    }
    if (pending) put_event(pending);
    put(kBandedEnd);
    while (pos & 7) {  // pad the last window with END
        if (dst) dst[pos] = kBandedEnd;
        pos++;
    }
    for (int i = 0; i < 8; i++) {  // spare window: the loop fetches one window ahead
        if (dst) dst[pos] = kBandedEnd;
        pos++;
    }
    return pos;
}
// Stream of the column-block Myers kernel: row codes with a CARRY token (code 7, no argument)
// in front of every 32nd row (rows_ir.py: myers_blocked_simulate).  Same contract as
// banded_stream_layout.
__host__ __device__ inline int blocked_stream_layout(int len, const char *row, unsigned char *dst)
{
    int pos = 0, slot = 0;
    auto put = [&](unsigned char b) {
        if (dst) dst[pos] = b;
        pos++;
        if (++slot == 7) {
            if (dst) dst[pos] = kCodeRefill;
            pos++;
            slot = 0;
        }
    };
    for (int r = 0; r < len; r++) {
        if (r > 0 && (r & 31) == 0) put(7);
        unsigned char code = 0;
        if (row) { code = static_cast<unsigned char>(row[r]); if (code > 4) code = 0; }
        put(code);
    }
    put(kCodeEnd);
    while (pos & 7) { if (dst) dst[pos] = kCodeEnd; pos++; }
    for (int i = 0; i < 8; i++) { if (dst) dst[pos] = kCodeEnd; pos++; }
    return pos;
}
int launch_pack_blocked(const char *d_content, int len, int ref_start, int ref_end, void *d_streams,
                        hipStream_t stream);

int launch_pack_banded(const char *d_content, int len, int k, int phase, int cut, int ref_start, int ref_end, void *d_streams,
                       hipStream_t stream);

int launch_banded(const char *d_content, const uint32_t *d_peq, int8_t *d_results, int ref_len,
                  int read_len, int64_t read_count, int ref_start, int ref_end, int word_num, int k,
                  void *d_workspace, hipStream_t stream);
const char *banded_kernel_name(int word_num);

// One compiled BitPAl score set (bitpal.hip; the kernels come from gen_bitpal_sets.py).
struct BitpalSet {
    int match, mismatch, gap;
    int planes;      // bit-planes of state per 32 subject columns
    int chains;      // inter-word carry chains per row (carry words per 32 rows of a column block)
    int carry_words; // > 0: packed-carry column blocks — this many carry words per direction and ROW (rows_ir.py: make_blocked_packed)
    int max_plain;   // widest subject, in words, whose state stays in registers; beyond: column blocks
    int valu_per_word;
    int (*launch)(const char *d_content, const uint32_t *d_peq, int16_t *d_results, int ref_len, int read_len,
                  int64_t read_count, int ref_start, int ref_end, int word_num, void *d_workspace,
                  hipStream_t stream, int semi_global);
    const char *(*kernel_name)(int word_num);
};
int bitpal_set_count();
const BitpalSet *bitpal_set_at(int i);
const BitpalSet *bitpal_find_set(int match, int mismatch, int gap);  // nullptr: not compiled in
int bitpal_common_factor(int match, int mismatch, int gap);  // the generator's commonFactor (Main.java:213-238)

// Scores with the kernels of compiled set `s` (capi.hip: make_plan picks it and the result factor).
int launch_bitpal(const BitpalSet *s, const char *d_content, const uint32_t *d_peq, int16_t *d_results, int ref_len,
                  int read_len, int64_t read_count, int ref_start, int ref_end, int word_num,
                  void *d_workspace, hipStream_t stream, int semi_global);
const char *bitpal_kernel_name(const BitpalSet *s, int word_num);

// Widest subject (in words) the Myers kernels keep whole in registers; wider ones run as column blocks.
int myers_max_plain_words();
int myers_semi_max_plain_words();   // the same for semi-global scoring (resident Peq planes, then code planes up to 32 words)

// long_kernels.hip: state-in-memory kernels for subjects beyond the register-resident limits (A/B only).
// Myers beyond kMaxWords words: column blocks of the generated body, per-wave carry buffers.
// Persistent workgroups of the column-block kernels (each owns a carry buffer slice): 512 = two per CU.
// BGSA_BLOCKED_WORKGROUPS overrides it for measurements.
inline int blocked_workgroups()
{
    static const int n = [] {
        const char *e = getenv("BGSA_BLOCKED_WORKGROUPS");
        const int v = e ? atoi(e) : 512;
        return (v >= 1 && v <= 65535) ? v : 512;
    }();
    return n;
}

// The persistent workgroups of a column-block kernel all start on the same clock and would run the same
// phase of the same row in lockstep, so the two waves of a SIMD stall at the same places; a per-workgroup
// start delay of up to ~8k cycles spreads them out (+1.5 % measured on 800 and 4,000 bp).
__device__ __forceinline__ void dephase_persistent_workgroup()
{
    const int skew = static_cast<int>((blockIdx.x * 2654435761u) >> 28);  // 0..15
    for (int i = 0; i < skew; i++) __builtin_amdgcn_s_sleep(8);             // ~512 cycles each
}

// ---- queries per task --------------------------------------------------------------------------------------------------
// A task is one wave's subject group(s) against `q_tile` consecutive queries: the Peq words are loaded once per task, so a
// task needs enough rows to pay for that (row_words = subject columns x 32-bit words of one query's recurrence; 512 of
// them are >= 6,000 instructions against ~100 of set-up).  The tile is halved from q_max until the launch has
// BGSA_TASK_TARGET tasks (default 16,384 = two per wave slot of a 256-CU part).  Finer is not better: with 262,144 a
// 12-query x 1M-subject x 150 bp launch (one tile of the coarse seam) went 1.43 -> 1.46 ms on the static grid, and to
// 2.21 ms with the task counter (187,500 requests to one address in 1.4 ms); 100 queries 10.62 -> 10.79 ms static
// (profiles/r03_seam_ab.txt).  A short launch loses its 5-10 % to start-up and drain, not to task size.
inline long long task_target()
{
    static const long long n = [] {
        const char *e = getenv("BGSA_TASK_TARGET");
        const long long v = e ? atoll(e) : 0;
        return v > 0 ? v : 16384ll;
    }();
    return n;
}
inline int pick_query_tile(int nq, long long wave_tasks_per_tile, long long row_words, int q_max,
                           long long target = task_target(), long long min_words = 512)
{
    int q_min = 1;
    while (q_min < q_max && q_min * row_words < min_words) q_min <<= 1;
    int q_tile = q_max;
    while (q_tile > q_min && ((nq + q_tile - 1) / q_tile) * wave_tasks_per_tile < target) q_tile >>= 1;
    while (q_tile < q_max && (nq + q_tile - 1) / q_tile > 65535) q_tile <<= 1;   // the static grids put the tile in blockIdx.y
    return q_tile;
}

// ---- dynamic task handout for the register-resident kernels ------------------------------------------------------------
// The chip's eight XCDs get the workgroups of a static grid round robin and do not sustain the same clock (bench.py's probe
// waves: 2,216 to 2,305 MHz under the Myers kernel on one box), so with equal shares the slowest XCD finishes last.  With a
// persistent grid whose WAVES take their (subject group, query tile) tasks from a device-wide counter every XCD takes work
// at the rate it runs at: Myers 10k x 1M x 150 bp 218,300 -> 222,900 GCUPS, 64 bp 225,600 -> 232,000 (same box, round 3).
// The counter sits behind the packed streams in the workspace (bgsa_hip_workspace_bytes reserves it) and is zeroed by the
// query packer that runs in front of the launch (launch_pack_queries: d_zero_word).  BGSA_DYNAMIC_TASKS=0 restores the static grids (A/B).
inline bool dynamic_tasks()
{
    static const bool on = [] {
        const char *e = getenv("BGSA_DYNAMIC_TASKS");
        return !(e && e[0] == '0');
    }();
    return on;
}
// Workgroups of a persistent grid: every CU's wave slots (8 waves per SIMD = 8 workgroups of four waves per CU).
inline int persistent_blocks()
{
    static const int n = [] {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        return cus * 8;
    }();
    return n;
}
// ... of ONE kernel: as many workgroups as the device really keeps resident for it (hipOccupancyMaxActiveBlocksPerMultiprocessor:
// a kernel at two waves per SIMD holds two workgroups per CU, not eight).  Workgroups beyond that would start only after the
// resident waves have drained the counter, each run its one statically assigned first task and leave: a static tail, the
// imbalance the counter is there to remove.  Per device and kernel; cached.
template <typename Kernel>
inline int persistent_blocks_for(Kernel kernel, size_t dynamic_lds = 0)
{
    struct Entry { int dev; const void *fn; size_t lds; int blocks; };
    static thread_local Entry seen[16] = {};      // a thread alternating between a few kernels asks the runtime once per kernel
    static thread_local unsigned next = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return persistent_blocks();
    const void *fn = reinterpret_cast<const void *>(kernel);
    for (const Entry &e : seen)
        if (e.fn == fn && e.dev == dev && e.lds == dynamic_lds && e.blocks > 0) return e.blocks;
    Entry &last = seen[next++ % 16];
    int cus = 256, per_cu = 0;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, dynamic_lds) != hipSuccess || per_cu < 1) {
        (void)hipGetLastError();
        per_cu = 8;
    }
    if (const char *e = getenv("BGSA_PERSISTENT_PER_CU")) {      // measurement knob: workgroups per CU of the persistent grids
        const int v = atoi(e);
        if (v >= 1 && v <= 8) per_cu = v;
    }
    last = Entry{dev, fn, dynamic_lds, cus * (per_cu < 8 ? per_cu : 8)};
    return last.blocks;
}
// The counter of a launch whose streams take stream_bytes of the workspace.
inline unsigned *task_counter_in(void *d_workspace, size_t stream_bytes)
{
    return reinterpret_cast<unsigned *>(static_cast<unsigned char *>(d_workspace) + ((stream_bytes + 255) & ~static_cast<size_t>(255)));
}
// Whether a launch of this many tasks may use the counter (32-bit task numbers).
// ... and only for launches long enough to gain from it: the counter costs a short launch a fixed 0.05-0.4 ms (150 bp x 1M
// subjects, kernel ms static / dynamic: 12 queries 1.41 / 1.54, 25: 2.80 / 2.97, 100: 10.73 / 10.95, 400: 42.90 / 42.15 —
// the crossover is between 8 and 25 tasks per wave slot), which is what made the coarse seam (12-query launches) slower.
// Default floor: 16 tasks per wave of the persistent grid; BGSA_DYNAMIC_MIN_TASKS=<n> overrides (tests use 1).
inline long long dynamic_min_tasks()
{
    static const long long n = [] {
        const char *e = getenv("BGSA_DYNAMIC_MIN_TASKS");
        const long long v = e ? atoll(e) : 0;
        return v > 0 ? v : 16ll * persistent_blocks() * kWavesPerBlock;
    }();
    return n;
}
inline bool dynamic_tasks_fit(long long n_tasks) { return n_tasks >= dynamic_min_tasks() && n_tasks < 0xffffffffll; }
// Queries per task and the handout of one launch.  With the counter a finer tile pays (100 / 200 / 400 queries x 1M x
// 150 bp: 10.30 / 20.44 / 41.05 ms static, 10.21 / 20.30 / 40.68 with the counter and 262,144 tasks as the target), as long
// as the requests stay rare: one address takes some tens of requests per microsecond from the whole chip (187,500 in a
// 1.4 ms launch cost 0.8 ms), so a counter task is at least 3,000 row-words (~0.2 ms for a wave among eight on its SIMD).
// BGSA_DYNAMIC_TASK_TARGET / BGSA_DYNAMIC_TASK_WORDS override the two numbers (tests, A/B).
struct TaskPlan {
    int q_tile;
    bool dynamic;
};
// Most queries per task of a counter launch.  A task loads its group's Peq block once and walks its queries, so the launch's HBM traffic is
// 2 B per score + block bytes / tile per pair: tiles of 32 (Myers) and 16 (BitPAl) — rounds 1-4 — read 1.7 x and 2.8 x SURVEY's
// algorithmic bytes (its model assumes the reference's 100-query blocks), 128 read 0.93 x.  At < 1 % of HBM peak neither costs time; the
// tile is still halved towards the task target for launches that need the tasks.  BGSA_QUERY_TILE_MAX (power of two, 8 .. 256) is the knob.
inline int query_tile_max()
{
    static const int v = [] {
        const char *e = getenv("BGSA_QUERY_TILE_MAX");
        const int t = e ? atoi(e) : 128;
        return (t >= 8 && t <= 256 && (t & (t - 1)) == 0) ? t : 128;
    }();
    return v;
}
inline TaskPlan plan_tasks(int nq, long long wave_tasks_per_tile, long long row_words, int q_max, bool counter_kernel, int q_max_counter = 0)
{
    static const long long target = [] { const char *e = getenv("BGSA_DYNAMIC_TASK_TARGET"); const long long v = e ? atoll(e) : 0; return v > 0 ? v : 262144ll; }();
    static const long long words = [] { const char *e = getenv("BGSA_DYNAMIC_TASK_WORDS"); const long long v = e ? atoll(e) : 0; return v > 0 ? v : 3000ll; }();
    if (counter_kernel && dynamic_tasks()) {
        const int q = pick_query_tile(nq, wave_tasks_per_tile, row_words, q_max_counter > 0 ? q_max_counter : q_max, target, words);
        if (dynamic_tasks_fit(((nq + q - 1) / q) * wave_tasks_per_tile)) return {q, true};
    }
    return {pick_query_tile(nq, wave_tasks_per_tile, row_words, q_max), false};
}
constexpr size_t kTaskCounterBytes = 512;   // what plan_workspace_bytes adds for it (alignment included)

// Tasks of a column-block kernel are handed out from a device-wide counter (zeroed by the launcher, it
// sits behind the carry buffers in the workspace): the XCDs do not sustain exactly the same clock, and with
// a static round-robin the slowest one finished last (average occupancy 96 %; 800 bp: 239 -> 227 ms).
__device__ __forceinline__ long long next_blocked_task(unsigned long long *counter)
{
    __shared__ unsigned long long s_task;
    if (threadIdx.x == 0) s_task = atomicAdd(counter, 1ull);
    __syncthreads();
    const unsigned long long task = s_task;
    __syncthreads();   // everyone has read it before thread 0 fetches the next one
    return static_cast<long long>(task);
}

// Queries per task of a column-block kernel: 2, or 1 while that still leaves fewer than 64 tasks per
// workgroup (the Peq planes are loaded per query and block either way, so a smaller tile costs nothing
// and shortens the tail of small problems).
inline int blocked_q_tile(int n_queries, int64_t n_groups)
{
    const int64_t tasks = ((n_groups + kWavesPerBlock - 1) / kWavesPerBlock) * ((n_queries + 1) / 2);
    return tasks < 64 * static_cast<int64_t>(blocked_workgroups()) ? 1 : 2;
}

inline size_t blocked_carry_bytes(int ref_len, int n_chains)
{
    return static_cast<size_t>((ref_len + 31) / 32) * n_chains * kLanes * sizeof(uint32_t) * kWavesPerBlock * blocked_workgroups();
}
size_t long_state_bytes(int algo, int word_num);
// Two flavours of the library are built from these sources (Makefile):
//   libbgsa_hip.so     the kernels that are defaults, and the BitPAl score sets of BITPAL_SETS;
//   libbgsa_hip_ab.so  (-DBGSA_AB_KERNELS=1, `make ab`) additionally every measured-and-not-adopted alternative that a
//                      measurement knob can select — the compiler-scheduled kernels (BGSA_*_IMPL=c, long_kernels.hip), the
//                      banded loops with the band held in place / straight-line rows (BGSA_BANDED_IMPL=p / s), the Myers
//                      code-plane kernels below 29 words (BGSA_MYERS_PEQ_MAX_WORDS) and code-plane column blocks
//                      (BGSA_MYERS_BLOCK_FORM=planes) — and the score sets of BITPAL_SETS_AB.
// A knob that asks the default flavour for a kernel it does not carry fails loudly (ab_knob_refused), it is never ignored.
#ifndef BGSA_AB_KERNELS
#define BGSA_AB_KERNELS 0
#endif
// `what` names the knob as the caller set it; returns BGSA_HIP_EUNSUPPORTED with the error text set.
int ab_knob_refused(const char *what);

int launch_long(int algo, const char *d_content, const uint32_t *d_peq, int16_t *d_results, int ref_len,
                int read_len, int64_t read_count, int ref_start, int ref_end, int word_num, void *d_state,
                hipStream_t stream);

// Queries per Peq / Mext load of this thread's last scoring launch (bgsa_hip_last_query_tile: bench.py models the
// launch's HBM traffic from it — tiles x block bytes + scores — where no PMC pass is at hand, e.g. per rank at N > 1).
void note_query_tile(int q_tile);

int launch_preprocess(int algo, const char *d_rows, int64_t avail_bytes, int len,
                      int64_t read_count, int word_num, int k, uint32_t *d_peq, hipStream_t stream);
int launch_map_queries(char *d_content, int64_t bytes, hipStream_t stream);
int launch_scale_scores(int16_t *d_scores, int64_t count, int factor, hipStream_t stream);

// ---- stream faults ------------------------------------------------------------------------------
// Every generated row loop hands back what is left of its window budget (gen_rows_asm.py: S_LEFT): >= 0
// after a well-formed stream, -1 if the budget ran out before an END token, -2 if a byte that is no
// stream code was dispatched.  Either means the wave did not score its query — the kernel then raises a
// bit in the device's sticky fault word instead of storing a wrong score silently.  The word lives in
// device memory (one per device, allocated and zeroed at the first launch) and is read back by
// bgsa_hip_stream_faults(); the host-buffer seams check it after every call.
unsigned *device_fault_word();   // capi.hip; nullptr + error text if the allocation failed
// One-shot fault injection for the tests of the above (bgsa_hip_debug_inject_stream_fault): the next
// scoring launch of this process overwrites the first packed stream with REFILL codes (kind 1) or with a
// byte that is no code (kind 2) between the packer and the scoring kernel.
// Called by every launcher between its packer and its scoring kernel: fetches the fault word and applies
// a pending injection.  refill_code = the REFILL token of this stream format; bad_code = a byte value
// its dispatch sends to a fail slot, or -1 if the format has none (then kind 2 degrades to kind 1).
int stream_guard(void *d_streams, int stride_bytes, int refill_code, int bad_code, hipStream_t stream,
                 unsigned **fault_word);

__device__ __forceinline__ void note_stream_fault(unsigned *fault_word, int left)
{
    if (left < 0) {  // wave-uniform: `left` comes back in an SGPR
        if ((threadIdx.x & (kLanes - 1)) == 0)
            atomicOr(fault_word, left == -1 ? static_cast<unsigned>(BGSA_HIP_FAULT_BUDGET) : static_cast<unsigned>(BGSA_HIP_FAULT_CODE));
    }
}

// ---- device helpers ---------------------------------------------------------------------------

// The next task of this WAVE from a device-wide counter (wave-uniform result).  32-bit task numbers: the launchers keep the
// static grid for the (hypothetical) launch with 2^32 tasks or more, and all of the index arithmetic stays unsigned — no
// sign-extended 64-bit scalar near an asm block (scripts/check_asm_kernels.py looks for exactly that).
__device__ __forceinline__ unsigned first_wave_task();
__device__ __forceinline__ unsigned issue_wave_task(unsigned *counter);
__device__ __forceinline__ unsigned resolve_wave_task(unsigned issued);

// A wave-uniform byte string read through the scalar data cache, four characters per fetch.
// Query rows have stride len+1 (reference cal_cpu.c:78), so a row may start at any byte
// alignment: the fetch is the aligned dword pair that covers the next four bytes.  Everything
// stays in SGPRs as long as the row pointer is wave-uniform.
struct UniformBytes {
    const uint32_t *words;  // row start rounded down to a dword
    uint32_t shift;         // 8 * (row start & 3)
    uint32_t window;        // the characters fetched but not yet consumed, next one in bits 7:0

    __device__ __forceinline__ explicit UniformBytes(const char *row)
    {
        uintptr_t a = reinterpret_cast<uintptr_t>(row);
        words = reinterpret_cast<const uint32_t *>(a & ~static_cast<uintptr_t>(3));
        shift = static_cast<uint32_t>(a & 3) * 8;
        window = 0;
    }
    // Call when (r & 3) == 0; `remaining` = characters of the row not yet fetched (>= 1).
    __device__ __forceinline__ void refill(int r, int remaining)
    {
        const uint32_t lo = words[r >> 2];
        uint32_t hi = 0;
        const int need_bits = 8 * (remaining < 4 ? remaining : 4);
        if (shift + need_bits > 32) hi = words[(r >> 2) + 1];  // only when the bytes straddle
        window = static_cast<uint32_t>(((static_cast<uint64_t>(hi) << 32) | lo) >> shift);
    }
    __device__ __forceinline__ uint32_t next()
    {
        uint32_t c = window & 0xffu;
        window >>= 8;
        return c;
    }
};

// A 64-bit value the compiler cannot prove wave-uniform, forced into an SGPR pair.
__device__ __forceinline__ unsigned long long uniform_u64(unsigned long long v)
{
    const unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(v));
    const unsigned hi = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(v >> 32));
    return (static_cast<unsigned long long>(hi) << 32) | lo;
}

// A wave's FIRST task is its own index in the persistent grid — no atomic: eight thousand waves asking one address at the
// same instant take their turns at the L2 (measured through the coarse host seam, whose calls are eight small launches:
// 11.8 ms per call with static grids, 12.5 with a counter fetch at the head of every wave's life).  The following tasks come
// from the counter (it counts from 0: task = grid waves + value), and each is asked for BEFORE the current one is processed
// (issue_wave_task ... resolve_wave_task), so the atomic's round trip runs under the task's work and the one request per
// wave that finds nothing left is spread over the last tasks instead of arriving in a burst when the waves finish.
__device__ __forceinline__ unsigned first_wave_task()
{
    return __builtin_amdgcn_readfirstlane(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));
}
__device__ __forceinline__ unsigned issue_wave_task(unsigned *counter)
{
    unsigned t = 0;
    if ((threadIdx.x & (kLanes - 1)) == 0) t = atomicAdd(counter, 1u);
    return t;   // lane 0 holds it; resolve_wave_task() makes it wave-uniform when it is needed
}
__device__ __forceinline__ unsigned resolve_wave_task(unsigned issued)
{
    return __builtin_amdgcn_readfirstlane(issued) + gridDim.x * kWavesPerBlock;
}

}  // namespace bgsa
