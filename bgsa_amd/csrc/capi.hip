// capi.hip — the C ABI declared in include/bgsa_hip.h.
//
// Two layers:
//   * the BGSA backend surface on HOST buffers (hip_handle_reads / align_hip /
//     hip_cal_align_score + the globals every reference backend defines), so the library can be
//     linked where original/BGSA_<ARCH>/{global.c,align_core.c,cal_<arch>.c} are linked;
//   * the device-resident layer (bgsa_hip_*_dev) the pipeline driver and bench use.
// No CPU fallback anywhere: without a GPU every compute entry point fails loudly.
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <string>
#include <thread>

#include "bgsa_common.h"

namespace bgsa {

static thread_local std::string g_last_error;

void set_error(const char *what, hipError_t e, const char *file, int line)
{
    char buf[512];
    snprintf(buf, sizeof buf, "%s failed: %s (%s:%d)", what, hipGetErrorString(e), file, line);
    g_last_error = buf;
}
void set_error_text(const char *text) { g_last_error = text; }

void host_handle_reads(int algo, const char *rows, int64_t avail, int len, uint32_t *result_reads,
                       int word_num, int64_t read_count, int k, int threads);

static int g_algo = BGSA_ALGO_MYERS;
static int g_alignment = BGSA_ALIGN_GLOBAL;

// Grow-only device workspace behind the host-buffer entry points.
struct HostPathWorkspace {
    void *d_content = nullptr, *d_peq = nullptr, *d_results = nullptr, *d_scratch = nullptr;
    size_t cap_content = 0, cap_peq = 0, cap_results = 0, cap_scratch = 0;
    int reserve(void **p, size_t *cap, size_t need)
    {
        if (need <= *cap) return BGSA_HIP_OK;
        if (*p) BGSA_HIP_TRY(hipFree(*p));
        *p = nullptr;
        *cap = 0;
        BGSA_HIP_TRY(hipMalloc(p, need));
        *cap = need;
        return BGSA_HIP_OK;
    }
};
static HostPathWorkspace g_ws;

[[noreturn]] static void die(const char *where)
{
    // Reference convention: print and exit(1) (original/BGSA_CPU/file.c:13-16).
    printf("Error - %s: %s\n", where, g_last_error.c_str());
    exit(1);
}

static size_t result_elem_size(int algo) { return algo == BGSA_ALGO_BANDED ? 1 : 2; }

}  // namespace bgsa

using namespace bgsa;

extern "C" {

// ---- globals of the reference's backend surface ------------------------------------------------
int match_score = 0;       // reference original/BGSA_CPU/align_core.c:13-17
int mismatch_score = -1;
int gap_score = -1;
int dvdh_len = 16;
int full_bits = 1;         // all 32 bits of a word carry data on this backend
int threshold = HIP_BANDED_WORD_SIZE / 2 - 1;  // banded/BGSA_CPU/main.c:43
int cpu_threads = 0;       // 0 = hardware concurrency
uint32_t mapping_table[128] __attribute__((aligned(64)));

void init_mapping_table(void)
{
    // reference original/BGSA_CPU/global.c:9-15 (table is zero-initialised: everything else -> 0)
    mapping_table[(int)'A'] = 0;
    mapping_table[(int)'C'] = 1;
    mapping_table[(int)'G'] = 2;
    mapping_table[(int)'T'] = 3;
    mapping_table[(int)'N'] = 4;
}

void *malloc_mem(uint64_t size)
{
    void *p = nullptr;
    if (posix_memalign(&p, 64, size ? size : 64) != 0) return nullptr;
    return p;
}
void free_mem(void *mem) { free(mem); }

int bgsa_hip_select_algorithm(int algo)
{
    switch (algo) {
    case BGSA_ALGO_MYERS:
    case BGSA_ALGO_BANDED:
        match_score = 0; mismatch_score = -1; gap_score = -1;
        full_bits = 1;
        break;
    case BGSA_ALGO_BITPAL:
        match_score = 2; mismatch_score = -3; gap_score = -5;  // original/BGSA_AVX2/align_core.c:13-15
        full_bits = 1;
        break;
    default:
        set_error_text("unknown algorithm");
        return BGSA_HIP_EINVAL;
    }
    dvdh_len = 16;
    g_algo = algo;
    return BGSA_HIP_OK;
}
int bgsa_hip_current_algorithm(void) { return g_algo; }

int bgsa_hip_select_alignment(int mode)
{
    if (mode != BGSA_ALIGN_GLOBAL && mode != BGSA_ALIGN_SEMIGLOBAL) {
        set_error_text("select_alignment: unknown mode");
        return BGSA_HIP_EINVAL;
    }
    g_alignment = mode;
    return BGSA_HIP_OK;
}
int bgsa_hip_current_alignment(void) { return g_alignment; }

int bgsa_hip_select_scores(int match, int mismatch, int gap)
{
    if (match == 0 && mismatch == 1 && gap == 1) {  // the generator's `-m 1`: Myers, result = +distance
        if (int rc = bgsa_hip_select_algorithm(BGSA_ALGO_MYERS)) return rc;
        mismatch_score = 1; gap_score = 1;
        return BGSA_HIP_OK;
    }
    const int old_m = match_score, old_x = mismatch_score, old_g = gap_score;
    match_score = match; mismatch_score = mismatch; gap_score = gap;
    const bool ok = match > mismatch && mismatch >= 2 * gap && gap < 0 && bitpal_current_set() != nullptr;
    match_score = old_m; mismatch_score = old_x; gap_score = old_g;
    if (!ok) {
        if (!(match > mismatch && mismatch >= 2 * gap && gap < 0))
            set_error_text("select_scores: BitPAl needs match > mismatch >= 2*gap and gap < 0");
        return BGSA_HIP_EUNSUPPORTED;
    }
    if (int rc = bgsa_hip_select_algorithm(BGSA_ALGO_BITPAL)) return rc;
    match_score = match; mismatch_score = mismatch; gap_score = gap;
    return BGSA_HIP_OK;
}
int bgsa_hip_score_set_count(void) { return bitpal_set_count(); }
int bgsa_hip_score_set(int index, int *match, int *mismatch, int *gap, int *valu_per_word)
{
    const BitpalSet *s = bitpal_set_at(index);
    if (!s) {
        set_error_text("score_set: index out of range");
        return BGSA_HIP_EINVAL;
    }
    if (match) *match = s->match;
    if (mismatch) *mismatch = s->mismatch;
    if (gap) *gap = s->gap;
    if (valu_per_word) *valu_per_word = s->valu_per_word;
    return BGSA_HIP_OK;
}

int bgsa_hip_word_num(int algo, int query_len, int subject_len, int k)
{
    switch (algo) {
    case BGSA_ALGO_MYERS: return (subject_len + 31) / 32;   // cal_cpu.c:252-253, full_bits
    case BGSA_ALGO_BITPAL: return (subject_len + 31) / 32;  // same layout as Myers
    case BGSA_ALGO_BANDED: {
        (void)query_len;                                     // equal lengths only (banded.hip)
        (void)k;
        return (subject_len + 31) / 32 + 3;                  // + zero words: 64-bit window of the last row, prefetch
    }
    default: return -1;
    }
}

size_t bgsa_hip_group_words(int algo, int word_num, int k)
{
    (void)algo;
    (void)k;
    return static_cast<size_t>(BGSA_CHAR_NUM) * word_num * HIP_V_NUM;
}

int bgsa_hip_release_workspace(void)
{
    void **slots[] = {&g_ws.d_content, &g_ws.d_peq, &g_ws.d_results, &g_ws.d_scratch};
    size_t *caps[] = {&g_ws.cap_content, &g_ws.cap_peq, &g_ws.cap_results, &g_ws.cap_scratch};
    for (int i = 0; i < 4; i++) {
        if (*slots[i]) BGSA_HIP_TRY(hipFree(*slots[i]));
        *slots[i] = nullptr;
        *caps[i] = 0;
    }
    return BGSA_HIP_OK;
}

// ---- device-resident layer ---------------------------------------------------------------------

const char *bgsa_hip_last_error(void) { return g_last_error.c_str(); }

int bgsa_hip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
int bgsa_hip_set_device(int device)
{
    BGSA_HIP_TRY(hipSetDevice(device));
    return BGSA_HIP_OK;
}
int bgsa_hip_malloc(void **dptr, size_t bytes)
{
    if (!dptr) return BGSA_HIP_EINVAL;
    BGSA_HIP_TRY(hipMalloc(dptr, bytes ? bytes : 1));
    return BGSA_HIP_OK;
}
int bgsa_hip_free(void *dptr)
{
    BGSA_HIP_TRY(hipFree(dptr));
    return BGSA_HIP_OK;
}
int bgsa_hip_malloc_host(void **hptr, size_t bytes)
{
    if (!hptr) return BGSA_HIP_EINVAL;
    BGSA_HIP_TRY(hipHostMalloc(hptr, bytes ? bytes : 1, hipHostMallocDefault));
    return BGSA_HIP_OK;
}
int bgsa_hip_free_host(void *hptr)
{
    BGSA_HIP_TRY(hipHostFree(hptr));
    return BGSA_HIP_OK;
}
int bgsa_hip_memcpy_h2d(void *dst, const void *src, size_t bytes, void *stream)
{
    BGSA_HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, static_cast<hipStream_t>(stream)));
    return BGSA_HIP_OK;
}
int bgsa_hip_memcpy_d2h(void *dst, const void *src, size_t bytes, void *stream)
{
    BGSA_HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream)));
    return BGSA_HIP_OK;
}
int bgsa_hip_memset(void *dst, int value, size_t bytes, void *stream)
{
    BGSA_HIP_TRY(hipMemsetAsync(dst, value, bytes, static_cast<hipStream_t>(stream)));
    return BGSA_HIP_OK;
}
int bgsa_hip_stream_create(void **stream)
{
    if (!stream) return BGSA_HIP_EINVAL;
    hipStream_t s;
    BGSA_HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = s;
    return BGSA_HIP_OK;
}
int bgsa_hip_stream_destroy(void *stream)
{
    BGSA_HIP_TRY(hipStreamDestroy(static_cast<hipStream_t>(stream)));
    return BGSA_HIP_OK;
}
int bgsa_hip_stream_synchronize(void *stream)
{
    BGSA_HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return BGSA_HIP_OK;
}

int bgsa_hip_handle_reads_dev(int algo, const char *d_rows, int64_t avail_bytes, int len,
                              int64_t read_count, int word_num, int k, hip_read_t *d_peq,
                              void *stream)
{
    if (!d_rows || !d_peq || len <= 0 || read_count < 0 || word_num <= 0 ||
        (read_count % HIP_V_NUM) != 0) {
        set_error_text("handle_reads_dev: bad argument (read_count must be a multiple of 64)");
        return BGSA_HIP_EINVAL;
    }
    return launch_preprocess(algo, d_rows, avail_bytes, len, read_count, word_num, k, d_peq,
                             static_cast<hipStream_t>(stream));
}

int bgsa_hip_map_queries_dev(char *d_content, int64_t bytes, void *stream)
{
    if (!d_content || bytes < 0) {
        set_error_text("map_queries_dev: bad argument");
        return BGSA_HIP_EINVAL;
    }
    return launch_map_queries(d_content, bytes, static_cast<hipStream_t>(stream));
}

size_t bgsa_hip_workspace_bytes(int algo, int ref_len, int read_len, int n_queries)
{
    if (ref_len <= 0 || n_queries <= 0) return 0;
    if (read_len > 0 && beyond_registers(algo, (read_len + 31) / 32)) {  // column blocks: streams + carry buffers
        const int chains = algo == BGSA_ALGO_BITPAL ? bitpal_current_set()->chains : 3;  // non-null: beyond_registers() saw it
        const size_t blocked = static_cast<size_t>(blocked_stream_layout(ref_len, nullptr, nullptr)) * n_queries + 256 +
                               blocked_carry_bytes(ref_len, chains) + 256;   // + the task counter
        // the A/B state-in-memory kernels (BGSA_MYERS_IMPL=c / BGSA_BITPAL_IMPL=c) keep the DP state here
        const char *ab = getenv(algo == BGSA_ALGO_BITPAL ? "BGSA_BITPAL_IMPL" : "BGSA_MYERS_IMPL");
        const size_t in_memory = (ab && ab[0] == 'c') ? long_state_bytes(algo, (read_len + 31) / 32) : 0;
        return blocked > in_memory ? blocked : in_memory;
    }
    if (algo == BGSA_ALGO_BANDED)  // the stream length depends on k, which this query does not know
        return banded_stream_bound(ref_len) * n_queries;
    return stream_stride(ref_len) * static_cast<size_t>(n_queries);
}

int bgsa_hip_cal_align_score_dev(int algo, const char *d_content, const hip_read_t *d_peq,
                                 void *d_results, int ref_len, int read_len, int64_t read_count,
                                 int ref_start, int ref_end, int word_num, int k,
                                 void *d_workspace, size_t workspace_bytes, void *stream)
{
    if (ref_start >= 0 && ref_end >= ref_start && read_count >= 0 && (ref_end == ref_start || read_count == 0))
        return BGSA_HIP_OK;  // an empty query window or an empty bucket: nothing to score
    if (!d_content || !d_peq || !d_results || ref_len <= 0 || read_len <= 0 || read_count < 0 ||
        (read_count % HIP_V_NUM) != 0 || ref_start < 0 || ref_end < ref_start || word_num <= 0) {
        set_error_text("cal_align_score_dev: bad argument (read_count must be a multiple of 64)");
        return BGSA_HIP_EINVAL;
    }
    if (g_alignment == BGSA_ALIGN_SEMIGLOBAL && algo == BGSA_ALGO_BANDED) {
        set_error_text("cal_align_score_dev: semi-global alignment is not defined for the banded filter");
        return BGSA_HIP_EUNSUPPORTED;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t need = bgsa_hip_workspace_bytes(algo, ref_len, read_len, ref_end - ref_start);
    if (d_workspace) {
        if (workspace_bytes < need) {
            set_error_text("cal_align_score_dev: workspace smaller than bgsa_hip_workspace_bytes()");
            return BGSA_HIP_EINVAL;
        }
    } else {
        if (g_ws.reserve(&g_ws.d_scratch, &g_ws.cap_scratch, need)) return BGSA_HIP_EHIP;
        d_workspace = g_ws.d_scratch;
    }
    switch (algo) {
    case BGSA_ALGO_MYERS:
        if (word_num != (read_len + 31) / 32) {
            set_error_text("cal_align_score_dev: word_num does not match read_len for Myers");
            return BGSA_HIP_EINVAL;
        }
    {
        // weights (0, 1, 1) — the generator's `-m 1` — report +distance; anything else (0/-1/-1, or the ints
        // of a BitPAl selection still in place while `algo` asks for Myers) the reference's -distance
        const bool positive = match_score == 0 && mismatch_score == 1 && gap_score == 1;
        if (int rc = launch_myers(d_content, d_peq, static_cast<int16_t *>(d_results), ref_len, read_len,
                                  read_count, ref_start, ref_end, word_num, d_workspace, s,
                                  g_alignment == BGSA_ALIGN_SEMIGLOBAL))
            return rc;
        return launch_scale_scores(static_cast<int16_t *>(d_results), static_cast<int64_t>(ref_end - ref_start) * read_count,
                                   positive ? -1 : 1, s);
    }
    case BGSA_ALGO_BANDED:
        return launch_banded(d_content, d_peq, static_cast<int8_t *>(d_results), ref_len, read_len,
                             read_count, ref_start, ref_end, word_num, k, d_workspace, s);
    case BGSA_ALGO_BITPAL:
        if (word_num != (read_len + 31) / 32) {
            set_error_text("cal_align_score_dev: word_num does not match read_len for BitPAl");
            return BGSA_HIP_EINVAL;
        }
        return launch_bitpal(d_content, d_peq, static_cast<int16_t *>(d_results), ref_len,
                             read_len, read_count, ref_start, ref_end, word_num, d_workspace, s,
                             g_alignment == BGSA_ALIGN_SEMIGLOBAL);
    default:
        set_error_text("cal_align_score_dev: unknown algorithm");
        return BGSA_HIP_EINVAL;
    }
}

int bgsa_hip_query_stream(int algo, const char *mapped_row, int ref_len, int k, unsigned char *dst, int cap)
{
    if (!mapped_row || ref_len <= 0) return BGSA_HIP_EINVAL;
    if (algo == BGSA_ALGO_BANDED) {
        const int n = banded_stream_layout(ref_len, k, nullptr, nullptr);
        if (dst && cap >= n) banded_stream_layout(ref_len, k, mapped_row, dst);
        return n;
    }
    if (algo == BGSA_ALGO_MYERS && k == -2) {  // the two-rows-per-token stream (subjects <= 64 bp)
        const int n = static_cast<int>(pair_stream_stride(ref_len));
        if (dst && cap >= n)
            for (int i = 0; i <= pair_stream_windows(ref_len); i++) {
                const unsigned long long w = pair_stream_window(mapped_row, ref_len, i);
                memcpy(dst + 8 * i, &w, 8);
            }
        return n;
    }
    if (algo == BGSA_ALGO_MYERS && k < 0) {  // k = -1: the column-block stream (subjects > 1024 bp)
        const int n = blocked_stream_layout(ref_len, nullptr, nullptr);
        if (dst && cap >= n) blocked_stream_layout(ref_len, mapped_row, dst);
        return n;
    }
    const int n = static_cast<int>(stream_stride(ref_len));
    if (dst && cap >= n)
        for (int i = 0; i <= stream_windows(ref_len); i++) {
            const unsigned long long w = plain_stream_window(mapped_row, ref_len, i);
            memcpy(dst + 8 * i, &w, 8);
        }
    return n;
}

const char *bgsa_hip_kernel_name(int algo, int word_num)
{
    switch (algo) {
    case BGSA_ALGO_MYERS: return myers_kernel_name(word_num);
    case BGSA_ALGO_BANDED: return banded_kernel_name(word_num);
    case BGSA_ALGO_BITPAL: return bitpal_kernel_name(word_num);
    default: return "";
    }
}

// ---- BGSA backend surface on host buffers --------------------------------------------------------

void hip_handle_reads(seq_t *read_seq, hip_read_t *result_reads, int word_num, int64_t read_start,
                      int64_t read_count)
{
    if (!read_seq || !read_seq->content || !result_reads || (read_count % HIP_V_NUM) != 0) {
        set_error_text("hip_handle_reads: bad argument (read_count must be a multiple of 64)");
        die("hip_handle_reads");
    }
    const int len = read_seq->len;
    const int64_t off = read_start * static_cast<int64_t>(len + 1);
    int threads = cpu_threads > 0 ? cpu_threads : static_cast<int>(std::thread::hardware_concurrency());
    host_handle_reads(g_algo, read_seq->content + off, read_seq->size - off, len, result_reads,
                      word_num, read_count, threshold, threads);
}

void hip_cal_align_score(char *content, hip_read_t *preprocess_reads, hip_write_t *align_results,
                         int ref_len, int ref_count, int read_len, int read_count, int ref_start,
                         int ref_end, int word_num, int chunk_read_num, hip_data_t *dvdh_bit_mem)
{
    (void)chunk_read_num;  // CPU cache-blocking knob (cal_cpu.c:266); the GPU grid tiles itself
    (void)dvdh_bit_mem;    // per-thread scratch of the CPU kernels; state lives in VGPRs here
    if (ref_end <= ref_start || read_count <= 0) return;
    // The reference calls align_<arch> from its OpenMP workers (cal_cpu.c:63-84): the staging buffers
    // below are shared, so concurrent callers take turns.
    static std::mutex seam;
    std::lock_guard<std::mutex> turn(seam);
    const size_t content_bytes = static_cast<size_t>(ref_count) * (ref_len + 1);
    const size_t peq_bytes = bgsa_hip_group_words(g_algo, word_num, threshold) * sizeof(hip_read_t) *
                             (static_cast<size_t>(read_count) / HIP_V_NUM);
    const size_t res_bytes = static_cast<size_t>(ref_end - ref_start) * read_count * result_elem_size(g_algo);
    if (g_ws.reserve(&g_ws.d_content, &g_ws.cap_content, content_bytes + 8) ||
        g_ws.reserve(&g_ws.d_peq, &g_ws.cap_peq, peq_bytes) ||
        g_ws.reserve(&g_ws.d_results, &g_ws.cap_results, res_bytes))
        die("hip_cal_align_score");
    if (hipMemcpy(g_ws.d_content, content, content_bytes, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(g_ws.d_peq, preprocess_reads, peq_bytes, hipMemcpyHostToDevice) != hipSuccess) {
        set_error_text("hipMemcpy H2D failed");
        die("hip_cal_align_score");
    }
    if (bgsa_hip_cal_align_score_dev(g_algo, static_cast<const char *>(g_ws.d_content),
                                     static_cast<const hip_read_t *>(g_ws.d_peq), g_ws.d_results,
                                     ref_len, read_len, read_count, ref_start, ref_end, word_num,
                                     threshold, nullptr, 0, nullptr) != BGSA_HIP_OK)
        die("hip_cal_align_score");
    hipError_t e = hipMemcpy(align_results, g_ws.d_results, res_bytes, hipMemcpyDeviceToHost);
    if (e != hipSuccess) {
        set_error("hipMemcpy D2H", e, __FILE__, __LINE__);
        die("hip_cal_align_score");
    }
}

void align_hip(char *ref, hip_read_t *read, int ref_len, int read_len, int word_num,
               int chunk_read_num, int result_index, hip_write_t *results, hip_data_t *dvdh_bit_mem)
{
    // One query against chunk_read_num groups: the coarse call with a 1-row query buffer and the
    // chunk as the whole bucket; results land at results[result_index * HIP_V_NUM ...]
    // (reference original/BGSA_CPU/align_core.c:138-145).
    const size_t esz = result_elem_size(g_algo);
    char *dst = reinterpret_cast<char *>(results) + static_cast<size_t>(result_index) * HIP_V_NUM * esz;
    hip_cal_align_score(ref, read, reinterpret_cast<hip_write_t *>(dst), ref_len, 1, read_len,
                        chunk_read_num * HIP_V_NUM, 0, 1, word_num, chunk_read_num, dvdh_bit_mem);
}

}  // extern "C"
