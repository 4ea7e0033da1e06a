// capi.hip — the C ABI declared in include/bgsa_hip.h.
//
// Two layers:
//   * the BGSA backend surface on HOST buffers (hip_handle_reads / align_hip /
//     hip_cal_align_score + the globals every reference backend defines), so the library can be
//     linked where original/BGSA_<ARCH>/{global.c,align_core.c,cal_<arch>.c} are linked;
//   * the device-resident layer (bgsa_hip_*_dev / *_ex) the pipeline driver and bench use.
// No CPU fallback anywhere: without a GPU every compute entry point fails loudly.
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <utility>
#include <vector>

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

static thread_local int g_last_q_tile = 0;
void note_query_tile(int q_tile) { g_last_q_tile = q_tile; }

void host_handle_reads(int algo, const char *rows, int64_t avail, int len, uint32_t *result_reads,
                       int word_num, int64_t read_count, int k, int threads);

static int g_algo = BGSA_ALGO_MYERS;
static int g_alignment = BGSA_ALIGN_GLOBAL;

[[noreturn]] static void die(const char *where)
{
    // Reference convention: print and exit(1) (original/BGSA_CPU/file.c:13-16).
    printf("Error - %s: %s\n", where, g_last_error.c_str());
    exit(1);
}

static size_t result_elem_size(int algo) { return algo == BGSA_ALGO_BANDED ? 1 : 2; }

// ---- per-device sticky fault word (bgsa_common.h "stream faults") -----------------------------------
static std::mutex g_fault_mu;
static std::map<int, unsigned *> g_fault_words;
static int g_inject_fault = 0;

unsigned *device_fault_word()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        set_error_text("fault word: no current device");
        return nullptr;
    }
    std::lock_guard<std::mutex> lock(g_fault_mu);
    auto it = g_fault_words.find(dev);
    if (it != g_fault_words.end()) return it->second;
    unsigned *p = nullptr;
    if (hipMalloc(reinterpret_cast<void **>(&p), 64) != hipSuccess || hipMemset(p, 0, 64) != hipSuccess) {
        set_error_text("fault word: device allocation failed");
        return nullptr;
    }
    g_fault_words[dev] = p;
    return p;
}

int take_injected_stream_fault()
{
    std::lock_guard<std::mutex> lock(g_fault_mu);
    const int kind = g_inject_fault;
    g_inject_fault = 0;
    return kind;
}

__global__ void corrupt_stream_kernel(unsigned char *stream, int bytes, unsigned char value)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < bytes) stream[i] = value;
}

int stream_guard(void *d_streams, int stride_bytes, int refill_code, int bad_code, hipStream_t stream,
                 unsigned **fault_word)
{
    *fault_word = device_fault_word();
    if (!*fault_word) return BGSA_HIP_EHIP;
    const int kind = take_injected_stream_fault();
    if (kind) {  // tests only: the first stream of this launch becomes all-REFILL, or all-<no code>
        const int value = (kind == 2 && bad_code >= 0) ? bad_code : refill_code;
        hipLaunchKernelGGL(corrupt_stream_kernel, dim3((stride_bytes + 63) / 64), dim3(64), 0, stream,
                           static_cast<unsigned char *>(d_streams), stride_bytes, static_cast<unsigned char>(value));
        BGSA_HIP_TRY(hipGetLastError());
    }
    return BGSA_HIP_OK;
}

// ---- library-owned scratch for callers that pass no workspace ---------------------------------------
// One grow-only buffer per (device, stream): launches on one stream are ordered, so the pack -> score
// sequence of consecutive calls cannot overlap on it; calls that use it hold `g_scratch_mu` for the whole
// launch sequence (two host threads sharing a stream would otherwise interleave their packers), and a
// buffer is only freed after its stream has drained.
struct Scratch {
    void *p = nullptr;
    size_t cap = 0;
};
static std::mutex g_scratch_mu;
static std::map<std::pair<int, hipStream_t>, Scratch> g_scratch;

static int scratch_reserve(hipStream_t stream, size_t need, void **out)  // g_scratch_mu held
{
    int dev = 0;
    BGSA_HIP_TRY(hipGetDevice(&dev));
    Scratch &e = g_scratch[{dev, stream}];
    if (need > e.cap) {
        if (e.p) {
            BGSA_HIP_TRY(hipStreamSynchronize(stream));  // an earlier launch may still be reading it
            BGSA_HIP_TRY(hipFree(e.p));
        }
        e.p = nullptr;
        e.cap = 0;
        BGSA_HIP_TRY(hipMalloc(&e.p, need));
        e.cap = need;
    }
    *out = e.p;
    return BGSA_HIP_OK;
}

// ---- what a set of scoring parameters runs as ---------------------------------------------------------
struct Plan {
    int kernel;               // BGSA_ALGO_* of the kernel family that runs
    const BitpalSet *set;     // BitPAl kernels of this compiled set (kernel == BGSA_ALGO_BITPAL)
    int factor;               // results are multiplied by this
    int semi;
};

// The generator's score handling (Main.java:213-271) plus two normalisations of its domain:
//   * a mismatch scoring below two gaps is never taken (insert + delete costs 2*gap), so any
//     mismatch < 2*gap is the mismatch = 2*gap instance;
//   * scores with a common factor f run as the reduced set and the result is multiplied by f;
//   * the reduced set 0/-1/-1 in global mode is minus the edit distance: it runs on the Myers body
//     (8 VALU per word against 17 for the BitPAl body of that set) — the reference's `isEdit`
//     specialisation (Main.java:270-271) taken to its conclusion.
static int make_plan(const bgsa_hip_params_t &p, Plan *plan)
{
    plan->set = nullptr;
    plan->factor = 1;
    plan->semi = p.alignment == BGSA_ALIGN_SEMIGLOBAL;
    plan->kernel = p.algo;
    switch (p.algo) {
    case BGSA_ALGO_MYERS:
        // weights (0, 1, 1) — the generator's `-m 1` — report +distance; anything else (0/-1/-1, or the ints
        // of a BitPAl selection still in place while `algo` asks for Myers) the reference's -distance
        plan->factor = (p.match == 0 && p.mismatch == 1 && p.gap == 1) ? -1 : 1;
        return BGSA_HIP_OK;
    case BGSA_ALGO_BANDED:
        if (plan->semi) {
            set_error_text("cal_align_score: semi-global alignment is not defined for the banded filter");
            return BGSA_HIP_EUNSUPPORTED;
        }
        return BGSA_HIP_OK;
    case BGSA_ALGO_BITPAL: {
        int m = p.match, x = p.mismatch, g = p.gap;
        if (!(g < 0 && m > x)) {
            set_error_text("bitpal: scores need match > mismatch and gap < 0");
            return BGSA_HIP_EUNSUPPORTED;
        }
        if (x < 2 * g) x = 2 * g;
        const int f = bitpal_common_factor(m, x, g);
        m /= f; x /= f; g /= f;
        plan->factor = f;
        if (m == 0 && x == -1 && g == -1 && !plan->semi) {
            plan->kernel = BGSA_ALGO_MYERS;
            return BGSA_HIP_OK;
        }
        plan->set = bitpal_find_set(m, x, g);
        if (!plan->set) {
            char msg[256];
            snprintf(msg, sizeof msg,
                     "bitpal: no kernels compiled for match %d / mismatch %d / gap %d (= %d/%d/%d reduced by the common "
                     "factor %d; rebuild with BITPAL_SETS, see bgsa_hip_score_set())", m, x, g, p.match, p.mismatch, p.gap, f);
            set_error_text(msg);
            return BGSA_HIP_EUNSUPPORTED;
        }
        return BGSA_HIP_OK;
    }
    default:
        set_error_text("unknown algorithm");
        return BGSA_HIP_EINVAL;
    }
}

static bool plan_beyond_registers(const Plan &plan, int word_num)
{
    if (plan.kernel == BGSA_ALGO_BITPAL) return word_num > plan.set->max_plain;
    if (plan.kernel == BGSA_ALGO_MYERS && plan.semi) return word_num > myers_semi_max_plain_words();
    return plan.kernel == BGSA_ALGO_MYERS && word_num > myers_max_plain_words();
}

static size_t plan_workspace_bytes(const Plan &plan, int ref_len, int read_len, int n_queries)
{
    if (ref_len <= 0 || n_queries <= 0) return 0;
    const int algo = plan.kernel;
    if (read_len > 0 && plan_beyond_registers(plan, (read_len + 31) / 32)) {  // column blocks: streams + carry buffers
        const int chains = algo == BGSA_ALGO_BITPAL ? plan.set->chains : 3;
        // packed-carry blocks (score sets with many chains): carry_words words per direction and query ROW, one row spare
        const size_t carries = (algo == BGSA_ALGO_BITPAL && plan.set->carry_words > 0)
                                   ? static_cast<size_t>(ref_len + 1) * plan.set->carry_words * kLanes * sizeof(uint32_t) * kWavesPerBlock * blocked_workgroups()
                                   : blocked_carry_bytes(ref_len, chains);
        const size_t blocked = static_cast<size_t>(blocked_stream_layout(ref_len, nullptr, nullptr)) * n_queries + 256 +
                               carries + 256;   // + the task counter
        // the A/B state-in-memory kernels (BGSA_MYERS_IMPL=c / BGSA_BITPAL_IMPL=c) keep the DP state here
#if BGSA_AB_KERNELS
        const char *ab = getenv(algo == BGSA_ALGO_BITPAL ? "BGSA_BITPAL_IMPL" : "BGSA_MYERS_IMPL");
        const size_t in_memory = (ab && ab[0] == 'c') ? long_state_bytes(algo, (read_len + 31) / 32) : 0;
        return blocked > in_memory ? blocked : in_memory;
#else
        return blocked;
#endif
    }
    if (algo == BGSA_ALGO_BANDED)  // the stream length depends on k: sized for the worst k
        return banded_stream_bound(ref_len) * n_queries + kTaskCounterBytes;
    return stream_stride(ref_len) * static_cast<size_t>(n_queries) + kTaskCounterBytes;   // bgsa_common.h "dynamic task handout"
}

int ab_knob_refused(const char *what)
{
    std::string msg = std::string(what) + " selects a kernel that only the A/B flavour of the library carries: build it with `make -C "
                      "bgsa_amd/csrc ab` and point BGSA_HIP_LIB at libbgsa_hip_ab.so";
    set_error_text(msg.c_str());
    return BGSA_HIP_EUNSUPPORTED;
}

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

// malloc_mem / free_mem (reference global.c:17-23: _mm_malloc(size, 64)).  Every large buffer of the
// reference's pipeline comes from here (cal_cpu.c:206-267: row buffers, Peq A/B, result A/B), so handing
// out page-locked memory makes all of the host seams' copies full-rate DMA without the host code knowing.
// Small blocks, and everything when no GPU is visible, come from the C heap.
static std::mutex g_pinned_mu;
static std::map<void *, size_t> g_pinned;
static std::map<void *, size_t> g_heap_blocks;   // the small blocks: only their extent is of interest (align_hip's read-ahead)
static const size_t kPinThreshold = 1u << 20;
static void forget_host_range(const void *p, size_t bytes);   // a freed buffer cannot stay a resident bucket

void *malloc_mem(uint64_t size)
{
    if (size >= kPinThreshold) {
        int n = 0;
        void *p = nullptr;
        if (hipGetDeviceCount(&n) == hipSuccess && n > 0 &&
            hipHostMalloc(&p, size, hipHostMallocPortable) == hipSuccess && p) {
            std::lock_guard<std::mutex> lock(g_pinned_mu);
            g_pinned[p] = size;
            return p;
        }
        (void)hipGetLastError();
    }
    void *p = nullptr;
    if (posix_memalign(&p, 64, size ? size : 64) != 0) return nullptr;
    {
        std::lock_guard<std::mutex> lock(g_pinned_mu);
        g_heap_blocks[p] = size;
    }
    return p;
}
// Bytes from p to the end of the malloc_mem() block that holds it (0: p is not inside a block of ours).
static size_t block_bytes_after(const void *p)
{
    std::lock_guard<std::mutex> lock(g_pinned_mu);
    for (const std::map<void *, size_t> *m : {&g_pinned, &g_heap_blocks}) {
        auto it = m->upper_bound(const_cast<void *>(p));
        if (it == m->begin()) continue;
        --it;
        const unsigned char *base = static_cast<const unsigned char *>(it->first), *q = static_cast<const unsigned char *>(p);
        if (q < base + it->second) return static_cast<size_t>(base + it->second - q);
    }
    return 0;
}

void free_mem(void *mem)
{
    if (!mem) return;
    size_t pinned_bytes = 0;
    {
        std::lock_guard<std::mutex> lock(g_pinned_mu);
        auto it = g_pinned.find(mem);
        if (it != g_pinned.end()) {
            pinned_bytes = it->second;
            g_pinned.erase(it);
        }
        g_heap_blocks.erase(mem);
    }
    forget_host_range(mem, pinned_bytes ? pinned_bytes : 1);
    if (pinned_bytes)
        (void)hipHostFree(mem);
    else
        free(mem);
}

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

int bgsa_hip_current_params(bgsa_hip_params_t *out)
{
    if (!out) return BGSA_HIP_EINVAL;
    out->algo = g_algo;
    out->alignment = g_alignment;
    out->match = match_score;
    out->mismatch = mismatch_score;
    out->gap = gap_score;
    out->k = threshold;
    return BGSA_HIP_OK;
}

int bgsa_hip_select_scores(int match, int mismatch, int gap)
{
    if (match == 0 && mismatch == 1 && gap == 1) {  // the generator's `-m 1`: Myers, result = +distance
        if (int rc = bgsa_hip_select_algorithm(BGSA_ALGO_MYERS)) return rc;
        mismatch_score = 1; gap_score = 1;
        return BGSA_HIP_OK;
    }
    bgsa_hip_params_t p = {BGSA_ALGO_BITPAL, BGSA_ALIGN_GLOBAL, match, mismatch, gap, 0};
    Plan plan;
    if (int rc = make_plan(p, &plan)) return rc == BGSA_HIP_EINVAL ? BGSA_HIP_EUNSUPPORTED : rc;
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
int bgsa_hip_mem_info(size_t *free_bytes, size_t *total_bytes)
{
    size_t f = 0, t = 0;
    BGSA_HIP_TRY(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
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
    BGSA_HIP_TRY(hipHostMalloc(hptr, bytes ? bytes : 1, hipHostMallocPortable));
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
    {   // its library-owned scratch goes with it
        std::lock_guard<std::mutex> lock(g_scratch_mu);
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess) {
            auto it = g_scratch.find({dev, static_cast<hipStream_t>(stream)});
            if (it != g_scratch.end()) {
                (void)hipStreamSynchronize(static_cast<hipStream_t>(stream));
                if (it->second.p) (void)hipFree(it->second.p);
                g_scratch.erase(it);
            }
        }
    }
    BGSA_HIP_TRY(hipStreamDestroy(static_cast<hipStream_t>(stream)));
    return BGSA_HIP_OK;
}
int bgsa_hip_stream_synchronize(void *stream)
{
    BGSA_HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return BGSA_HIP_OK;
}
int bgsa_hip_event_create(void **event)
{
    if (!event) return BGSA_HIP_EINVAL;
    hipEvent_t e;
    BGSA_HIP_TRY(hipEventCreate(&e));
    *event = e;
    return BGSA_HIP_OK;
}
int bgsa_hip_event_destroy(void *event)
{
    BGSA_HIP_TRY(hipEventDestroy(static_cast<hipEvent_t>(event)));
    return BGSA_HIP_OK;
}
int bgsa_hip_event_record(void *event, void *stream)
{
    BGSA_HIP_TRY(hipEventRecord(static_cast<hipEvent_t>(event), static_cast<hipStream_t>(stream)));
    return BGSA_HIP_OK;
}
int bgsa_hip_event_synchronize(void *event)
{
    BGSA_HIP_TRY(hipEventSynchronize(static_cast<hipEvent_t>(event)));
    return BGSA_HIP_OK;
}
int bgsa_hip_event_elapsed_ms(void *start, void *stop, float *ms)
{
    if (!ms) return BGSA_HIP_EINVAL;
    BGSA_HIP_TRY(hipEventElapsedTime(ms, static_cast<hipEvent_t>(start), static_cast<hipEvent_t>(stop)));
    return BGSA_HIP_OK;
}
int bgsa_hip_stream_wait_event(void *stream, void *event)
{
    BGSA_HIP_TRY(hipStreamWaitEvent(static_cast<hipStream_t>(stream), static_cast<hipEvent_t>(event), 0));
    return BGSA_HIP_OK;
}

int bgsa_hip_stream_faults(int clear)
{
    int dev = 0;
    BGSA_HIP_TRY(hipGetDevice(&dev));
    unsigned *word = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_fault_mu);
        auto it = g_fault_words.find(dev);
        if (it == g_fault_words.end()) return 0;  // nothing was ever launched on this device
        word = it->second;
    }
    unsigned value = 0;
    BGSA_HIP_TRY(hipMemcpy(&value, word, sizeof value, hipMemcpyDeviceToHost));
    if (value && clear) BGSA_HIP_TRY(hipMemset(word, 0, sizeof value));
    if (value) {
        char msg[200];
        snprintf(msg, sizeof msg, "stream fault on device %d: flags 0x%x (%s%s) — at least one query was not scored", dev, value,
                 (value & BGSA_HIP_FAULT_BUDGET) ? "window budget exhausted before END " : "",
                 (value & BGSA_HIP_FAULT_CODE) ? "byte that is no stream code dispatched" : "");
        set_error_text(msg);
    }
    return static_cast<int>(value);
}

int bgsa_hip_debug_inject_stream_fault(int kind)
{
    if (kind < 0 || kind > 2) return BGSA_HIP_EINVAL;
    std::lock_guard<std::mutex> lock(g_fault_mu);
    g_inject_fault = kind;
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
    if (word_num != bgsa_hip_word_num(algo, len, len, k)) {
        set_error_text("handle_reads_dev: word_num is not bgsa_hip_word_num() for this algorithm and length");
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

size_t bgsa_hip_workspace_bytes_ex(const bgsa_hip_params_t *params, int ref_len, int read_len, int n_queries)
{
    Plan plan;
    if (!params || make_plan(*params, &plan)) return 0;
    return plan_workspace_bytes(plan, ref_len, read_len, n_queries);
}

size_t bgsa_hip_workspace_bytes(int algo, int ref_len, int read_len, int n_queries)
{
    bgsa_hip_params_t p;
    bgsa_hip_current_params(&p);
    p.algo = algo;
    return bgsa_hip_workspace_bytes_ex(&p, ref_len, read_len, n_queries);
}

int bgsa_hip_cal_align_score_ex(const bgsa_hip_params_t *params, const char *d_content, const hip_read_t *d_peq,
                                void *d_results, int ref_len, int read_len, int64_t read_count,
                                int ref_start, int ref_end, int word_num,
                                void *d_workspace, size_t workspace_bytes, void *stream)
{
    if (!params) {
        set_error_text("cal_align_score: params is NULL");
        return BGSA_HIP_EINVAL;
    }
    if (ref_start >= 0 && ref_end >= ref_start && read_count >= 0 && (ref_end == ref_start || read_count == 0))
        return BGSA_HIP_OK;  // an empty query window or an empty bucket: nothing to score
    if (!d_content || !d_peq || !d_results || ref_len <= 0 || read_len <= 0 || read_count < 0 ||
        (read_count % HIP_V_NUM) != 0 || ref_start < 0 || ref_end < ref_start || word_num <= 0) {
        set_error_text("cal_align_score_dev: bad argument (read_count must be a multiple of 64)");
        return BGSA_HIP_EINVAL;
    }
    Plan plan;
    if (int rc = make_plan(*params, &plan)) return rc;
    // the kernels index the Peq / Mext blocks with the caller's word_num: it must be the layout's
    if (word_num != bgsa_hip_word_num(params->algo, ref_len, read_len, params->k)) {
        set_error_text(params->algo == BGSA_ALGO_BANDED
                           ? "cal_align_score_dev: word_num is not bgsa_hip_word_num(BGSA_ALGO_BANDED, ...) — the Mext layout "
                             "has ceil(len/32)+3 words, not the reference's banded formula (cal_cpu.c:253-254)"
                           : "cal_align_score_dev: word_num does not match read_len");
        return BGSA_HIP_EINVAL;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t need = plan_workspace_bytes(plan, ref_len, read_len, ref_end - ref_start);
    std::unique_lock<std::mutex> own_scratch(g_scratch_mu, std::defer_lock);
    if (d_workspace) {
        if (workspace_bytes < need) {
            set_error_text("cal_align_score_dev: workspace smaller than bgsa_hip_workspace_bytes()");
            return BGSA_HIP_EINVAL;
        }
    } else {
        own_scratch.lock();
        if (int rc = scratch_reserve(s, need, &d_workspace)) return rc;
    }
    const int64_t n_scores = static_cast<int64_t>(ref_end - ref_start) * read_count;
    switch (plan.kernel) {
    case BGSA_ALGO_MYERS:
        if (int rc = launch_myers(d_content, d_peq, static_cast<int16_t *>(d_results), ref_len, read_len,
                                  read_count, ref_start, ref_end, word_num, d_workspace, s, plan.semi))
            return rc;
        return launch_scale_scores(static_cast<int16_t *>(d_results), n_scores, plan.factor, s);
    case BGSA_ALGO_BANDED:
        return launch_banded(d_content, d_peq, static_cast<int8_t *>(d_results), ref_len, read_len,
                             read_count, ref_start, ref_end, word_num, params->k, d_workspace, s);
    case BGSA_ALGO_BITPAL:
        if (int rc = launch_bitpal(plan.set, d_content, d_peq, static_cast<int16_t *>(d_results), ref_len,
                                   read_len, read_count, ref_start, ref_end, word_num, d_workspace, s, plan.semi))
            return rc;
        return launch_scale_scores(static_cast<int16_t *>(d_results), n_scores, plan.factor, s);
    default:
        set_error_text("cal_align_score_dev: unknown algorithm");
        return BGSA_HIP_EINVAL;
    }
}

int bgsa_hip_cal_align_score_dev(int algo, const char *d_content, const hip_read_t *d_peq,
                                 void *d_results, int ref_len, int read_len, int64_t read_count,
                                 int ref_start, int ref_end, int word_num, int k,
                                 void *d_workspace, size_t workspace_bytes, void *stream)
{
    bgsa_hip_params_t p;  // the process-global selection, read once, here
    bgsa_hip_current_params(&p);
    p.algo = algo;
    p.k = k;
    return bgsa_hip_cal_align_score_ex(&p, d_content, d_peq, d_results, ref_len, read_len, read_count, ref_start,
                                       ref_end, word_num, d_workspace, workspace_bytes, stream);
}

int bgsa_hip_query_stream(int algo, const char *mapped_row, int ref_len, int k, unsigned char *dst, int cap)
{
    if (!mapped_row || ref_len <= 0) return BGSA_HIP_EINVAL;
    if (algo == BGSA_ALGO_BANDED) {
        const int phase = banded_stream_phase(k), cut = banded_stream_cut(k);
        const int n = banded_stream_layout(ref_len, k, phase, cut, nullptr, nullptr);
        if (dst && cap >= n) banded_stream_layout(ref_len, k, phase, cut, mapped_row, dst);
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

int bgsa_hip_last_query_tile(void) { return g_last_q_tile; }

const char *bgsa_hip_kernel_name(int algo, int word_num)
{
    bgsa_hip_params_t p;
    bgsa_hip_current_params(&p);
    p.algo = algo;
    Plan plan;
    if (make_plan(p, &plan)) return algo == BGSA_ALGO_BITPAL ? "bitpal: score set not compiled" : "";
    switch (plan.kernel) {
    case BGSA_ALGO_MYERS: return myers_kernel_name(word_num, plan.semi);
    case BGSA_ALGO_BANDED: return banded_kernel_name(word_num);
    case BGSA_ALGO_BITPAL: return bitpal_kernel_name(plan.set, word_num);
    default: return "";
    }
}

// ---- BGSA backend surface on host buffers --------------------------------------------------------
//
// The seams take host buffers and return when the results are in host memory, like the reference's.
// Behind them (one instance per process, guarded by `g_seam`):
//   * device mirrors of the query buffer, the Peq bucket and the result tile, grow-only;
//   * RESIDENT BUCKETS: hip_handle_reads remembers the host range it filled; the first scoring call on
//     that range uploads it and later calls reuse the device copy until hip_handle_reads (or
//     bgsa_hip_bucket_release) touches the range again — what the KNC backend does with `nocopy ... RETAIN`
//     (BGSA_KNC/cal_mic.c:348-356).  The reference's loop scores 100 queries per call against the same
//     bucket (cal_cpu.c:363-401), so the 100 MB of Peq cross PCIe once per bucket instead of once per
//     call.  A host that builds Peq words itself registers them with bgsa_hip_bucket_resident();
//     bgsa_hip_set_auto_resident(0) restores upload-on-every-call;
//   * the query buffer is re-uploaded only when its bytes changed (memcmp against a host copy: 1.5 MB);
//   * copies are asynchronous on a private stream; buffers from malloc_mem() are page-locked.

struct ResidentRange {
    const unsigned char *host = nullptr;   // host address of the Peq words
    size_t bytes = 0;                      // host bytes
    int w_host = 0, w_dev = 0;             // 32-bit words per class and lane: host layout, device layout
    void *dev = nullptr;                   // device mirror (nullptr: not uploaded yet)
    int device = -1;
    bool uploaded = false;
    uint64_t gen = 0;                      // identity of this content: a rewritten range gets a new one
    uint64_t fp = 0;                       // range_fingerprint() of the host bytes the device copy was made from
    // those host bytes themselves, where the policy keeps them (wants_shadow): ranges up to kStrictAutoBytes by default, every
    // range in strict mode.  Immutable once made and shared with the threads' LastRow, so the lock-free path of align_hip
    // compares against it without the seam lock and without touching anything but the chunk it was handed.
    std::shared_ptr<const std::vector<unsigned char>> shadow;
};

// A registered range that the caller rewrites by other means than hip_handle_reads (a memcpy of a saved bucket, a host
// that builds Peq itself and forgets bgsa_hip_bucket_resident) breaks the contract of include/bgsa_hip.h — a resident range
// changes only through hip_handle_reads / bgsa_hip_bucket_resident — and the library still tries not to score the stale
// device copy (SURVEY 8(b) "Ownership": the callee keeps no state between calls; the KNC precedent,
// BGSA_KNC/cal_mic.c:348-356, justifies residency, not silence).  Two checks, by range size:
//   * EXACT, ranges up to kStrictAutoBytes (8 MiB; every range under BGSA_HIP_STRICT_RESIDENT=1): the library keeps the host
//     bytes it uploaded and every scoring call compares the bytes it is about to use against them (memcmp: < 1 ms for a
//     whole 8 MiB bucket, per call of the coarse seam; a chunk per call of the fine one);
//   * BEST EFFORT above that: a fingerprint of kFingerprintLines cache lines — the first, the last and a golden-ratio (Weyl)
//     sequence of positions in between, a few hundred nanoseconds once the lines are in the calling core's cache.  A rewrite
//     that changes none of the sampled lines is NOT seen: that is what the contract, or strict mode, is for.
// A range found changed is uploaded again and its cached rows dropped.  The sample positions must not be an arithmetic
// progression: a stride of range / n lines is, for n - 1 groups, the group size itself, and every sample then lands in the
// same plane and word of its group — the last word of the class-N plane, all zeros in every bucket (found by
// scripts/soak_seams.py: 33 groups, 34 samples).
static constexpr size_t kFingerprintLines = 66;
static uint64_t range_fingerprint(const unsigned char *p, size_t bytes)
{
    uint64_t h = 0x9E3779B97F4A7C15ull ^ bytes;
    const size_t lines = bytes / 64;
    auto mix = [&](uint64_t w) { h = (h ^ w) * 0x100000001B3ull; h ^= h >> 29; };
    if (lines < 2) {
        for (size_t i = 0; i < bytes; i++) mix(p[i]);
        return h;
    }
    const size_t n = std::min(lines, kFingerprintLines);
    uint64_t weyl = 0;
    for (size_t j = 0; j < n; j++) {
        // quasi-random over the range: about a fifth of the samples fall into the plane of a class no read contains (all
        // zeros in every bucket), the others land in different planes, words and lanes of different groups
        weyl += 0x9E3779B97F4A7C15ull;
        const size_t line = j == 0 ? 0 : (j == 1 ? lines - 1 : static_cast<size_t>((static_cast<unsigned __int128>(weyl) * lines) >> 64));
        uint64_t w[8];
        memcpy(w, p + line * 64, 64);
        for (uint64_t x : w) mix(x);
    }
    return h;
}
static constexpr size_t kStrictAutoBytes = 8u << 20;
static std::atomic<int> g_strict_resident{-2};   // 1: every range keeps its host copy; 0: ranges <= kStrictAutoBytes do (default); -1: none; -2: not decided yet
static int strict_mode()
{
    int v = g_strict_resident.load(std::memory_order_relaxed);
    if (v < -1) {
        const char *e = getenv("BGSA_HIP_STRICT_RESIDENT");   // "1": always, "-1": never (the fingerprint alone, a measurement knob), else the default
        v = (e && e[0] == '1') ? 1 : ((e && e[0] == '-') ? -1 : 0);
        g_strict_resident.store(v, std::memory_order_relaxed);
    }
    return v;
}
static bool wants_shadow(size_t bytes)
{
    const int m = strict_mode();
    return m == 1 || (m == 0 && bytes <= kStrictAutoBytes);
}

// 32-bit words per (class, lane) of a HOST Peq buffer whose caller passed `word_num`, or -1.  Myers and
// BitPAl: the library's own layout only.  Banded: the library's Mext layout (ceil(len/32)+3 words of
// uint32), or the size the reference's banded host computes — word_num = (len - h + 63)/64 + 1 words of
// its 64-bit cpu_read_t (banded/BGSA_CPU/cal_cpu.c:253-254, config.h:26) = twice as many 32-bit words.
// That buffer always holds the len + k + 1 bits of the offset match string (k <= 31), but not the zero
// words behind it that the kernels' 64-bit windows and prefetch rely on, so the seams re-pitch it to the
// device layout on upload.
static int host_words32(int algo, int len, int k, int word_num)
{
    const int own = bgsa_hip_word_num(algo, len, len, k);
    if (word_num == own) return own;
    if (algo == BGSA_ALGO_BANDED && word_num == (len - k + 63) / 64 + 1) return 2 * word_num;
    return -1;
}

// Host Peq blocks [group][class][w_host][lane] -> device blocks [group][class][w_dev][lane].
static int upload_peq(void *dev, const void *host, size_t groups, int w_host, int w_dev, hipStream_t s)
{
    const size_t rows = groups * BGSA_CHAR_NUM;
    if (w_host == w_dev) {
        BGSA_HIP_TRY(hipMemcpyAsync(dev, host, rows * w_dev * 256, hipMemcpyHostToDevice, s));
        return BGSA_HIP_OK;
    }
    BGSA_HIP_TRY(hipMemsetAsync(dev, 0, rows * w_dev * 256, s));
    const int w = w_host < w_dev ? w_host : w_dev;
    BGSA_HIP_TRY(hipMemcpy2DAsync(dev, static_cast<size_t>(w_dev) * 256, host, static_cast<size_t>(w_host) * 256,
                                  static_cast<size_t>(w) * 256, rows, hipMemcpyHostToDevice, s));
    return BGSA_HIP_OK;
}

// align_hip's row cache.  The reference's grid calls align_<arch> once per (query, chunk of ~27 groups)
// (cal_cpu.c:63-84): 57,900 calls per 100-query block of a 1M-subject bucket, each a launch + two copies if
// taken literally.  With the bucket resident, the first call for a query scores it against the WHOLE bucket in
// one launch and keeps the row on the host; the other 578 calls for that query are a memcpy of their chunk.
// A row is identified by the bucket's content id, the query's bytes and the scoring parameters.
// Where a cached row lives: a slot of one page-locked arena (the device-to-host copy lands in it directly — staging
// a row and copying it once more cost more than scoring it), or, when the arena has no slot of that size, the heap.
// One launch of align_hip's row cache: its rows are in the cache as soon as the launch is issued, usable once `done`
// (recorded behind the last copy down) has been waited for.  A batch issued ahead of the calls runs on the GPU while
// the host threads are still copying chunks out of the previous one.
struct RowBuf;
struct RowBatch {
    hipEvent_t done = nullptr;
    bool waited = false;
    bool chained = false;                 // the batch behind this one has been issued (or cannot be)
    bool full = false;                    // as many rows as a launch takes: the calls are walking the buffer
    const char *next = nullptr;           // the query row behind the last one of this batch
    std::string queries;                  // the rows as uploaded (the caller's buffer may change under a launch in flight)
    std::vector<std::pair<RowBuf *, size_t>> staged;   // rows without an arena slot: copied after the wait, from `region`
    int region = 0;
    size_t row_size = 0;                  // bytes per row of THIS launch (another resident bucket's rows may differ)
    ~RowBatch() { if (done) (void)hipEventDestroy(done); }
};
struct RowBuf {
    unsigned char *p = nullptr;
    size_t size = 0;
    int slot = -1;            // >= 0: arena slot, given back when the last holder lets go; -1: heap
    std::shared_ptr<RowBatch> pending;   // the launch that fills it
};
struct RowArena {
    std::mutex mu;            // the free list: rows are released by whichever thread drops the last reference
    unsigned char *base = nullptr;
    size_t slot_bytes = 0, n_slots = 0;
    std::vector<int> free_slots;
    bool tried = false;
};
static RowArena g_arena;
static constexpr size_t kRowArenaBytes = size_t(1) << 30;

static void row_release(RowBuf *b)
{
    if (b->pending && !b->pending->waited) {   // dropped with its launch still in flight (eviction, bucket rewritten):
        if (b->pending->done) (void)hipEventSynchronize(b->pending->done);   // never hand a slot back under a copy in flight
        auto &st = b->pending->staged;         // and the launch must not copy into this buffer later (g_seam is held here:
        for (size_t j = 0; j < st.size();)     // only served rows, whose `pending` is gone, are released by other threads)
            if (st[j].first == b) st.erase(st.begin() + j); else j++;
    }
    b->pending.reset();
    if (b->slot >= 0) {
        std::lock_guard<std::mutex> lock(g_arena.mu);
        g_arena.free_slots.push_back(b->slot);
    } else {
        delete[] b->p;
    }
    delete b;
}

// A buffer for one row of `size` bytes; *pinned says whether a device copy may land in it directly.
static std::shared_ptr<RowBuf> row_take(size_t size, bool *pinned)   // g_seam held
{
    RowBuf *b = new RowBuf;
    b->size = size;
    {
        std::lock_guard<std::mutex> lock(g_arena.mu);
        if (!g_arena.tried) {      // carved once, for the row size of the first bucket seen
            g_arena.tried = true;
            const size_t slot = (size + 4095) & ~size_t(4095);
            size_t bytes = std::min(kRowArenaBytes, std::max(size_t(64) << 20, 512 * slot));   // small buckets: a small arena
            if (const char *e = getenv("BGSA_HIP_ROW_ARENA")) {   // "0": no arena — every row takes the staged path (tests)
                if (e[0] == '0') bytes = 0;
            }
            while (bytes >= 4 * slot && bytes >= (size_t(64) << 20)) {
                void *m = nullptr;
                if (hipHostMalloc(&m, bytes, hipHostMallocPortable) == hipSuccess && m) {
                    g_arena.base = static_cast<unsigned char *>(m);
                    g_arena.slot_bytes = slot;
                    g_arena.n_slots = bytes / slot;
                    for (int i = static_cast<int>(bytes / slot) - 1; i >= 0; i--) g_arena.free_slots.push_back(i);
                    break;
                }
                (void)hipGetLastError();
                bytes >>= 1;
            }
        }
        if (g_arena.base && size <= g_arena.slot_bytes && !g_arena.free_slots.empty()) {
            b->slot = g_arena.free_slots.back();
            g_arena.free_slots.pop_back();
            b->p = g_arena.base + static_cast<size_t>(b->slot) * g_arena.slot_bytes;
        }
    }
    if (b->slot < 0) b->p = new unsigned char[size ? size : 1];
    *pinned = b->slot >= 0;
    return std::shared_ptr<RowBuf>(b, row_release);
}

// Rows of `size` bytes do not fit the arena's slots (it was carved for a smaller bucket): true if the arena is idle —
// the caller has dropped its cached rows and no thread holds one — and has been given up, so that the next row_take()
// carves it anew for this size.
static bool row_arena_recarve(size_t size)   // g_seam held
{
    std::lock_guard<std::mutex> lock(g_arena.mu);
    if (!g_arena.base || size <= g_arena.slot_bytes || g_arena.free_slots.size() != g_arena.n_slots) return false;
    (void)hipHostFree(g_arena.base);
    g_arena.base = nullptr;
    g_arena.slot_bytes = g_arena.n_slots = 0;
    g_arena.free_slots.clear();
    g_arena.tried = false;
    return true;
}
// Free slots of the arena for rows of `size` bytes (-1: the arena does not serve that size, or does not exist yet).
static long row_arena_free(size_t size)
{
    std::lock_guard<std::mutex> lock(g_arena.mu);
    if (!g_arena.base || size > g_arena.slot_bytes) return -1;
    return static_cast<long>(g_arena.free_slots.size());
}
// Slots of the arena that are not on its free list: held by cached rows or by some thread's last row.
static size_t row_arena_slots_out()
{
    std::lock_guard<std::mutex> lock(g_arena.mu);
    return g_arena.n_slots - g_arena.free_slots.size();
}
// Slots the arena has for rows of `size` bytes: the arena that EXISTS once it is carved (it may have been halved when
// hipHostMalloc refused, or carved for another bucket's rows), the carving rule before that.
static size_t row_arena_slots_for(size_t size)
{
    const size_t slot = (size + 4095) & ~size_t(4095);
    std::lock_guard<std::mutex> lock(g_arena.mu);
    if (g_arena.base && g_arena.slot_bytes && size <= g_arena.slot_bytes) return g_arena.n_slots;
    // no arena (yet, or none could be had, or carved for smaller rows): such rows take the heap / staged path, which
    // the cache's byte budget bounds; the carving rule stands in for the count
    return std::min(kRowArenaBytes, std::max(size_t(64) << 20, 512 * slot)) / std::max<size_t>(slot, 1);
}
static bool row_arena_too_small(size_t size)
{
    std::lock_guard<std::mutex> lock(g_arena.mu);
    return g_arena.base && size > g_arena.slot_bytes;
}

struct CachedRow {
    uint64_t range_gen = 0;
    bgsa_hip_params_t params{};
    int read_len = 0;
    std::string query;                                         // the mapped query row, ref_len bytes
    std::shared_ptr<RowBuf> scores;                            // [subjects of the range] x element size
    uint64_t stamp = 0;
};

struct HostSeam {
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;   // hip_cal_align_score: copies tile t out while tile t+1 is scored
    static constexpr int kCopyTiles = 8;   // at most; seam_tiles() of them are used
    hipEvent_t tile_done[kCopyTiles] = {};
    int device = -1;
    void *d_content = nullptr, *d_peq = nullptr, *d_results = nullptr, *d_rowq = nullptr;
    size_t cap_content = 0, cap_peq = 0, cap_results = 0, cap_rowq = 0;
    std::vector<CachedRow> rows;
    void *h_stage = nullptr;           // page-locked landing buffer of a row (copies to pageable memory are far slower)
    size_t cap_stage = 0;
    uint64_t next_gen = 1, clock = 0, row_hits = 0, row_misses = 0, rows_ahead = 0;
    const unsigned char *last_miss = nullptr;   // where align_hip's previous row miss was (read-ahead heuristic)
    size_t last_miss_stride = 0;
    static constexpr int kRowRegions = 3;       // device buffers of the row launches in flight (one miss + its successor + one spare)
    void *d_rowq_ring[kRowRegions] = {}, *d_row_results[kRowRegions] = {};
    size_t cap_rowq_ring[kRowRegions] = {}, cap_row_results[kRowRegions] = {};
    std::shared_ptr<RowBatch> region_owner[kRowRegions];
    unsigned *h_fault = nullptr;                // page-locked: the device's fault word as of each region's last launch
    uint64_t batch_seq = 0, prefetch_launches = 0;
    size_t row_bytes = 0;
    static constexpr size_t kRowCacheBytes = 1u << 30;
    std::vector<unsigned char> content_copy;   // what d_content holds
    std::vector<ResidentRange> ranges;
    bool auto_resident = true;
    uint64_t peq_uploads = 0, peq_upload_bytes = 0, calls = 0, stale_ranges = 0;
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
static std::mutex g_seam;
static HostSeam g_host;
// bumped whenever a resident range is dropped or replaced: lock-free validity test of a thread's last row
static std::atomic<uint64_t> g_range_epoch{1};
static std::atomic<uint64_t> g_row_fast_hits{0};   // align_hip calls served from the calling thread's last row

static void drop_overlapping(const unsigned char *lo, size_t bytes)  // g_seam held
{
    for (size_t i = 0; i < g_host.ranges.size();) {
        ResidentRange &r = g_host.ranges[i];
        if (lo < r.host + r.bytes && r.host < lo + bytes) {
            g_range_epoch.fetch_add(1, std::memory_order_release);
            if (r.dev) (void)hipFree(r.dev);
            for (size_t j = 0; j < g_host.rows.size();) {      // its cached rows go with it
                if (g_host.rows[j].range_gen == r.gen) {
                    g_host.row_bytes -= g_host.rows[j].scores->size;
                    g_host.rows.erase(g_host.rows.begin() + j);
                } else {
                    j++;
                }
            }
            g_host.ranges.erase(g_host.ranges.begin() + i);
        } else {
            i++;
        }
    }
}

static void forget_host_range(const void *p, size_t bytes)
{
    std::lock_guard<std::mutex> turn(g_seam);
    if (!g_host.ranges.empty()) drop_overlapping(static_cast<const unsigned char *>(p), bytes);
}

// Query tiles per hip_cal_align_score call (BGSA_HIP_SEAM_TILES, 1..8; 1 = score the block, then copy it).
static int seam_tiles()
{
    static const int n = [] {
        const char *e = getenv("BGSA_HIP_SEAM_TILES");
        const int v = e ? atoi(e) : HostSeam::kCopyTiles;
        return (v >= 1 && v <= HostSeam::kCopyTiles) ? v : HostSeam::kCopyTiles;
    }();
    return n;
}

// Query rows align_hip scores per launch when the calls walk a query buffer (BGSA_HIP_ROW_AHEAD, 1..256; 1 = none).
static int seam_row_ahead()
{
    static const int n = [] {
        const char *e = getenv("BGSA_HIP_ROW_AHEAD");
        // 10k x 1M x 150 bp through the reference's own pipeline (oracle/_ref/original_hip/aligner -N 16; scripts/r03_rowahead.sh,
        // profiles/r03_rowahead.txt), its `cal GCUPS` at 32 / 50 / 64 / 100 / 128 rows: 121-140k / 133k / 150-158k / 230k / 94k.
        // 100 is the reference's REF_BUCKET_COUNT (config.h): a launch is then exactly one of its query blocks, the
        // speculative successor launch the next block, scored while the reference's writer thread still writes this one —
        // its cal timer only sees the copy-out.  Rows that straddle blocks (64, 128) leave launches half used.
        const int v = e ? atoi(e) : 100;
        return (v >= 1 && v <= 256) ? v : 100;
    }();
    return n;
}

static bool row_batch_finish(const std::shared_ptr<RowBatch> &bt);   // below: wait + staged rows down

static int seam_stream()  // g_seam held
{
    int dev = 0;
    BGSA_HIP_TRY(hipGetDevice(&dev));
    if (g_host.stream && g_host.device == dev) return BGSA_HIP_OK;
    if (g_host.stream) {   // the caller moved to another device: start over there
        (void)hipStreamSynchronize(g_host.stream);
        (void)hipStreamDestroy(g_host.stream);
        g_host.stream = nullptr;
        if (g_host.copy_stream) {
            (void)hipStreamSynchronize(g_host.copy_stream);
            (void)hipStreamDestroy(g_host.copy_stream);
            g_host.copy_stream = nullptr;
        }
        for (hipEvent_t &e : g_host.tile_done) {
            if (e) (void)hipEventDestroy(e);
            e = nullptr;
        }
        for (void **p : {&g_host.d_content, &g_host.d_peq, &g_host.d_results, &g_host.d_rowq}) {
            if (*p) (void)hipFree(*p);
            *p = nullptr;
        }
        // launches still in flight on the old device: bring their rows down — the staged ones too, which sit in the
        // region buffers freed below — while that device is current
        (void)hipSetDevice(g_host.device);
        for (int i = 0; i < HostSeam::kRowRegions; i++) (void)row_batch_finish(g_host.region_owner[i]);
        (void)hipSetDevice(dev);
        for (int i = 0; i < HostSeam::kRowRegions; i++) {
            g_host.region_owner[i].reset();
            for (void **p : {&g_host.d_rowq_ring[i], &g_host.d_row_results[i]}) {
                if (*p) (void)hipFree(*p);
                *p = nullptr;
            }
            g_host.cap_rowq_ring[i] = g_host.cap_row_results[i] = 0;
        }
        g_host.cap_content = g_host.cap_peq = g_host.cap_results = g_host.cap_rowq = 0;
        g_host.content_copy.clear();
        for (ResidentRange &r : g_host.ranges) {
            if (r.dev) (void)hipFree(r.dev);
            r.dev = nullptr;
            r.uploaded = false;
        }
    }
    BGSA_HIP_TRY(hipStreamCreateWithFlags(&g_host.stream, hipStreamNonBlocking));
    BGSA_HIP_TRY(hipStreamCreateWithFlags(&g_host.copy_stream, hipStreamNonBlocking));
    for (hipEvent_t &e : g_host.tile_done) BGSA_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    g_host.device = dev;
    return BGSA_HIP_OK;
}

// The resident range that holds `bytes` at `peq_host` in this layout, uploaded; nullptr if there is none.
static ResidentRange *resident_range(const unsigned char *peq_host, size_t bytes, int w_host, int w_dev, hipStream_t s)  // g_seam held
{
    const size_t host_group_bytes = static_cast<size_t>(BGSA_CHAR_NUM) * w_host * 256;
    const size_t dev_group_bytes = static_cast<size_t>(BGSA_CHAR_NUM) * w_dev * 256;
    for (ResidentRange &r : g_host.ranges) {
        if (peq_host < r.host || peq_host + bytes > r.host + r.bytes || r.w_host != w_host || r.w_dev != w_dev ||
            (peq_host - r.host) % host_group_bytes != 0)
            continue;
        if (r.uploaded && r.device == g_host.device) {
            // the bytes the device copy was made from must still be there (see range_fingerprint above); strict mode
            // compares every byte of the part this call is about to use
            bool stale = range_fingerprint(r.host, r.bytes) != r.fp;
            if (!stale && r.shadow)
                stale = memcmp(r.shadow->data() + (peq_host - r.host), peq_host, bytes) != 0;
            if (stale) {
                g_host.stale_ranges++;
                g_range_epoch.fetch_add(1, std::memory_order_release);
                for (size_t j = 0; j < g_host.rows.size();) {      // rows scored from the old content go with it
                    if (g_host.rows[j].range_gen == r.gen) {
                        g_host.row_bytes -= g_host.rows[j].scores->size;
                        g_host.rows.erase(g_host.rows.begin() + j);
                    } else {
                        j++;
                    }
                }
                r.gen = g_host.next_gen++;
                r.uploaded = false;
            }
        }
        if (!r.uploaded || r.device != g_host.device) {
            const size_t r_groups = r.bytes / host_group_bytes;
            if (r.dev && r.device != g_host.device) { (void)hipFree(r.dev); r.dev = nullptr; }
            if (!r.dev && hipMalloc(&r.dev, r_groups * dev_group_bytes) != hipSuccess) {
                set_error_text("hipMalloc (resident bucket) failed");
                die("resident bucket");
            }
            if (upload_peq(r.dev, r.host, r_groups, w_host, w_dev, s)) die("resident bucket");
            r.fp = range_fingerprint(r.host, r.bytes);
            if (wants_shadow(r.bytes)) {
                // the copy above may still be reading a pageable source through the runtime's staging: the shadow must hold
                // exactly what travels, so let it finish first (once per upload of a bucket; small ranges by default)
                if (hipStreamSynchronize(s) != hipSuccess) die("resident bucket");
                r.shadow = std::make_shared<const std::vector<unsigned char>>(r.host, r.host + r.bytes);
                r.fp = range_fingerprint(r.host, r.bytes);
            } else {
                r.shadow.reset();
            }
            r.device = g_host.device;
            r.uploaded = true;
            g_host.peq_uploads++;
            g_host.peq_upload_bytes += r.bytes;
        }
        return &r;
    }
    return nullptr;
}

int bgsa_hip_set_auto_resident(int on)
{
    std::lock_guard<std::mutex> turn(g_seam);
    g_host.auto_resident = on != 0;
    return BGSA_HIP_OK;
}

int bgsa_hip_set_strict_resident(int on)
{
    std::lock_guard<std::mutex> turn(g_seam);
    g_strict_resident.store(on > 0 ? 1 : (on < 0 ? -1 : 0), std::memory_order_relaxed);
    g_range_epoch.fetch_add(1, std::memory_order_release);   // no thread keeps serving from a row checked the other way
    // Every range is uploaded again on its next use, under the new policy — and as NEW content: the host bytes may have
    // changed since the upload in a way the old policy could not see, so the rows cached from the old device copy go too.
    for (ResidentRange &r : g_host.ranges) {
        if (r.uploaded) {
            for (size_t j = 0; j < g_host.rows.size();) {
                if (g_host.rows[j].range_gen == r.gen) {
                    g_host.row_bytes -= g_host.rows[j].scores->size;
                    g_host.rows.erase(g_host.rows.begin() + j);
                } else {
                    j++;
                }
            }
            r.gen = g_host.next_gen++;
            r.uploaded = false;
        }
        r.shadow.reset();
    }
    return BGSA_HIP_OK;
}

int bgsa_hip_stale_ranges(uint64_t *count)
{
    std::lock_guard<std::mutex> turn(g_seam);
    if (count) *count = g_host.stale_ranges;
    return BGSA_HIP_OK;
}

int bgsa_hip_bucket_resident(const hip_read_t *host_peq, size_t bytes, int word_num)
{
    if (!host_peq || bytes == 0 || word_num <= 0 || bytes % (static_cast<size_t>(BGSA_CHAR_NUM) * word_num * 256) != 0) {
        set_error_text("bucket_resident: bad argument (bytes must be whole groups of 5 x word_num x 64 words)");
        return BGSA_HIP_EINVAL;
    }
    std::lock_guard<std::mutex> turn(g_seam);
    const unsigned char *lo = reinterpret_cast<const unsigned char *>(host_peq);
    drop_overlapping(lo, bytes);
    ResidentRange r;
    r.host = lo;
    r.bytes = bytes;
    r.w_host = r.w_dev = word_num;   // the library's own layout
    r.gen = g_host.next_gen++;
    g_host.ranges.push_back(r);
    return BGSA_HIP_OK;
}

int bgsa_hip_bucket_release(const hip_read_t *host_peq)
{
    std::lock_guard<std::mutex> turn(g_seam);
    if (g_host.stream) (void)hipStreamSynchronize(g_host.stream);
    if (!host_peq) {
        drop_overlapping(nullptr, ~static_cast<size_t>(0));
        return BGSA_HIP_OK;
    }
    drop_overlapping(reinterpret_cast<const unsigned char *>(host_peq), 1);
    return BGSA_HIP_OK;
}

int bgsa_hip_seam_stats(uint64_t *calls, uint64_t *peq_uploads, uint64_t *peq_upload_bytes)
{
    std::lock_guard<std::mutex> turn(g_seam);
    if (calls) *calls = g_host.calls;
    if (peq_uploads) *peq_uploads = g_host.peq_uploads;
    if (peq_upload_bytes) *peq_upload_bytes = g_host.peq_upload_bytes;
    return BGSA_HIP_OK;
}

// BGSA_HIP_SEAM_STATS=1: print the counters of the host seams when the process ends (what the unmodified
// reference pipeline did with the library: calls, bucket uploads, row-cache hits and misses, time in them).
static std::atomic<uint64_t> g_row_ns{0}, g_lock_wait_ns{0}, g_issue_ns{0}, g_event_ns{0}, g_scan_ns{0}, g_fault_ns{0};
static void print_seam_stats()
{
    fprintf(stderr, "[bgsa_hip] seam calls %llu, Peq uploads %llu (%.1f MB), align_hip row launches %llu on a miss + %llu ahead of the calls (%llu rows read ahead), waited %.3f s for rows, "
                    "served from a row: %llu under the lock + %llu from the caller's last row; waited for the seam lock %.3f s; issuing %.3f s, in hipEventSynchronize %.3f s, scanning %.3f s, fault check %.3f s\n",
            (unsigned long long)g_host.calls, (unsigned long long)g_host.peq_uploads, g_host.peq_upload_bytes / 1e6,
            (unsigned long long)g_host.row_misses, (unsigned long long)g_host.prefetch_launches, (unsigned long long)g_host.rows_ahead, g_row_ns.load() / 1e9, (unsigned long long)g_host.row_hits,
            (unsigned long long)g_row_fast_hits.load(), g_lock_wait_ns.load() / 1e9, g_issue_ns.load() / 1e9, g_event_ns.load() / 1e9, g_scan_ns.load() / 1e9, g_fault_ns.load() / 1e9);
}
static void seam_stats_at_exit()
{
    static std::once_flag once;
    std::call_once(once, [] {
        const char *e = getenv("BGSA_HIP_SEAM_STATS");
        if (e && e[0] == '1') atexit(print_seam_stats);
    });
}
static inline uint64_t now_ns()
{
    return static_cast<uint64_t>(std::chrono::duration_cast<std::chrono::nanoseconds>(
        std::chrono::steady_clock::now().time_since_epoch()).count());
}

// Waits for one row launch and brings down the rows that have no arena slot (through the page-locked staging buffer,
// sized for the largest row seen).  Uses the batch's OWN row size and region: the batch may belong to another resident
// bucket than the call that happens to wait for it.  g_seam held.  false: a HIP call failed.
static bool row_batch_finish(const std::shared_ptr<RowBatch> &bt)
{
    if (!bt || bt->waited) return true;
    const uint64_t t_ev = now_ns();
    bool ok = !bt->done || hipEventSynchronize(bt->done) == hipSuccess;
    g_event_ns.fetch_add(now_ns() - t_ev, std::memory_order_relaxed);
    if (ok && !bt->staged.empty() && g_host.cap_stage < bt->row_size) {
        if (g_host.h_stage) (void)hipHostFree(g_host.h_stage);
        g_host.h_stage = nullptr;
        g_host.cap_stage = 0;
        ok = hipHostMalloc(&g_host.h_stage, bt->row_size, hipHostMallocPortable) == hipSuccess;
        if (ok) g_host.cap_stage = bt->row_size;
    }
    for (size_t j = 0; ok && j < bt->staged.size(); j++) {
        ok = g_host.d_row_results[bt->region] &&
             hipMemcpyAsync(g_host.h_stage, static_cast<unsigned char *>(g_host.d_row_results[bt->region]) + bt->staged[j].second,
                            bt->row_size, hipMemcpyDeviceToHost, g_host.stream) == hipSuccess &&
             hipStreamSynchronize(g_host.stream) == hipSuccess;
        if (ok) memcpy(bt->staged[j].first->p, g_host.h_stage, bt->row_size);
    }
    bt->staged.clear();
    bt->waited = true;
    return ok;
}

int bgsa_hip_row_cache_stats(uint64_t *hits, uint64_t *misses)
{
    std::lock_guard<std::mutex> turn(g_seam);
    if (hits) *hits = g_host.row_hits + g_row_fast_hits.load(std::memory_order_relaxed);
    if (misses) *misses = g_host.row_misses;
    return BGSA_HIP_OK;
}

int bgsa_hip_release_workspace(void)
{
    {
        std::lock_guard<std::mutex> turn(g_seam);
        if (g_host.stream) (void)hipStreamSynchronize(g_host.stream);
        for (void **p : {&g_host.d_content, &g_host.d_peq, &g_host.d_results, &g_host.d_rowq}) {
            if (*p) BGSA_HIP_TRY(hipFree(*p));
            *p = nullptr;
        }
        g_host.cap_content = g_host.cap_peq = g_host.cap_results = g_host.cap_rowq = 0;
        g_host.content_copy.clear();
        drop_overlapping(nullptr, ~static_cast<size_t>(0));
    }
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    for (auto &kv : g_scratch) {
        if (!kv.second.p) continue;
        (void)hipStreamSynchronize(kv.first.second);
        BGSA_HIP_TRY(hipFree(kv.second.p));
    }
    g_scratch.clear();
    return BGSA_HIP_OK;
}

void hip_handle_reads(seq_t *read_seq, hip_read_t *result_reads, int word_num, int64_t read_start,
                      int64_t read_count)
{
    if (!read_seq || !read_seq->content || !result_reads || (read_count % HIP_V_NUM) != 0) {
        set_error_text("hip_handle_reads: bad argument (read_count must be a multiple of 64)");
        die("hip_handle_reads");
    }
    const int len = read_seq->len;
    const int algo = g_algo, k = threshold;
    const int w_host = host_words32(algo, len, k, word_num);
    if (w_host < 0) {
        set_error_text("hip_handle_reads: word_num matches neither bgsa_hip_word_num() nor, for the banded filter, "
                       "the reference's (len - h + 63)/64 + 1 (banded/BGSA_CPU/cal_cpu.c:253-254)");
        die("hip_handle_reads");
    }
    const int64_t off = read_start * static_cast<int64_t>(len + 1);
    int threads = cpu_threads > 0 ? cpu_threads : static_cast<int>(std::thread::hardware_concurrency());
    const size_t bytes = static_cast<size_t>(BGSA_CHAR_NUM) * w_host * 256 * (static_cast<size_t>(read_count) / HIP_V_NUM);
    {   // the range is about to change: forget any device copy of it
        std::lock_guard<std::mutex> turn(g_seam);
        drop_overlapping(reinterpret_cast<const unsigned char *>(result_reads), bytes ? bytes : 1);
    }
    host_handle_reads(algo, read_seq->content + off, read_seq->size - off, len, result_reads,
                      w_host, read_count, k, threads);
    std::lock_guard<std::mutex> turn(g_seam);
    if (g_host.auto_resident && bytes) {
        ResidentRange r;
        r.host = reinterpret_cast<const unsigned char *>(result_reads);
        r.bytes = bytes;
        r.w_host = w_host;
        r.w_dev = bgsa_hip_word_num(algo, len, len, k);
        r.gen = g_host.next_gen++;
        g_host.ranges.push_back(r);
    }
}

void hip_cal_align_score(char *content, hip_read_t *preprocess_reads, hip_write_t *align_results,
                         int ref_len, int ref_count, int read_len, int read_count, int ref_start,
                         int ref_end, int word_num, int chunk_read_num, hip_data_t *dvdh_bit_mem)
{
    (void)chunk_read_num;  // CPU cache-blocking knob (cal_cpu.c:266); the GPU grid tiles itself
    (void)dvdh_bit_mem;    // per-thread scratch of the CPU kernels; state lives in VGPRs here
    if (ref_end <= ref_start || read_count <= 0) return;
    // The reference calls align_<arch> from its OpenMP workers (cal_cpu.c:63-84): the device mirrors
    // below are shared, so concurrent callers take turns.
    std::lock_guard<std::mutex> turn(g_seam);
    bgsa_hip_params_t params;   // the process-global selection, read once per call, under the lock
    bgsa_hip_current_params(&params);
    g_host.calls++;
    if (seam_stream()) die("hip_cal_align_score");
    hipStream_t s = g_host.stream;
    const size_t content_bytes = static_cast<size_t>(ref_count) * (ref_len + 1);
    const int w_host = host_words32(params.algo, read_len, params.k, word_num);
    if (w_host < 0) {
        set_error_text("hip_cal_align_score: word_num matches neither bgsa_hip_word_num() nor, for the banded filter, "
                       "the reference's (len - h + 63)/64 + 1 (banded/BGSA_CPU/cal_cpu.c:253-254)");
        die("hip_cal_align_score");
    }
    const int w_dev = bgsa_hip_word_num(params.algo, ref_len, read_len, params.k);
    const size_t groups = static_cast<size_t>(read_count) / HIP_V_NUM;
    const size_t host_group_bytes = static_cast<size_t>(BGSA_CHAR_NUM) * w_host * 256;
    const size_t dev_group_bytes = static_cast<size_t>(BGSA_CHAR_NUM) * w_dev * 256;
    const size_t peq_bytes = host_group_bytes * groups;
    const size_t res_bytes = static_cast<size_t>(ref_end - ref_start) * read_count * result_elem_size(params.algo);

    // queries: upload when the bytes differ from what the device holds
    if (g_host.reserve(&g_host.d_content, &g_host.cap_content, content_bytes + 8)) die("hip_cal_align_score");
    if (g_host.content_copy.size() != content_bytes || memcmp(g_host.content_copy.data(), content, content_bytes) != 0) {
        g_host.content_copy.assign(reinterpret_cast<unsigned char *>(content), reinterpret_cast<unsigned char *>(content) + content_bytes);
        if (hipMemcpyAsync(g_host.d_content, g_host.content_copy.data(), content_bytes, hipMemcpyHostToDevice, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess) {   // content_copy is pageable: finish before it can change
            set_error_text("hipMemcpy H2D (queries) failed");
            die("hip_cal_align_score");
        }
    }

    // Peq: a resident range, or the per-call mirror
    const unsigned char *peq_host = reinterpret_cast<const unsigned char *>(preprocess_reads);
    const hip_read_t *d_peq = nullptr;
    if (ResidentRange *r = resident_range(peq_host, peq_bytes, w_host, w_dev, s))
        d_peq = reinterpret_cast<const hip_read_t *>(static_cast<unsigned char *>(r->dev) +
                                                     (peq_host - r->host) / host_group_bytes * dev_group_bytes);
    if (!d_peq) {
        if (g_host.reserve(&g_host.d_peq, &g_host.cap_peq, groups * dev_group_bytes)) die("hip_cal_align_score");
        if (upload_peq(g_host.d_peq, preprocess_reads, groups, w_host, w_dev, s)) die("hip_cal_align_score");
        g_host.peq_uploads++;
        g_host.peq_upload_bytes += peq_bytes;
        d_peq = static_cast<const hip_read_t *>(g_host.d_peq);
    }
    if (g_host.reserve(&g_host.d_results, &g_host.cap_results, res_bytes)) die("hip_cal_align_score");
    // The block is scored in up to eight query tiles; tile t travels to the host on a second stream while tile t+1 is
    // scored (the copy-out of a reference-sized Myers block is 3.9 ms against 10.4 ms of kernel: 15.9 -> 11.9 ms per call
    // with page-locked result buffers; tiles 1 / 2 / 4 / 8: 15.9 / 13.8 / 12.4 / 11.9 ms).
    // The contract stays: every score is in align_results when the call returns.
    const int nq = ref_end - ref_start;
    const size_t row_bytes = static_cast<size_t>(read_count) * result_elem_size(params.algo);
    const int tiles = res_bytes < (size_t(8) << 20) ? 1 : std::max(1, std::min(seam_tiles(), nq / 8));
    hipError_t e = hipSuccess;
    for (int t = 0; t < tiles && e == hipSuccess; t++) {
        const int qs = ref_start + static_cast<int>(static_cast<long long>(nq) * t / tiles);
        const int qe = ref_start + static_cast<int>(static_cast<long long>(nq) * (t + 1) / tiles);
        const size_t off = static_cast<size_t>(qs - ref_start) * row_bytes, bytes = static_cast<size_t>(qe - qs) * row_bytes;
        if (bgsa_hip_cal_align_score_ex(&params, static_cast<const char *>(g_host.d_content), d_peq,
                                        static_cast<unsigned char *>(g_host.d_results) + off, ref_len, read_len, read_count,
                                        qs, qe, w_dev, nullptr, 0, s) != BGSA_HIP_OK)
            die("hip_cal_align_score");
        e = hipEventRecord(g_host.tile_done[t], s);
        if (e == hipSuccess) e = hipStreamWaitEvent(g_host.copy_stream, g_host.tile_done[t], 0);
        if (e == hipSuccess)
            e = hipMemcpyAsync(reinterpret_cast<unsigned char *>(align_results) + off,
                               static_cast<unsigned char *>(g_host.d_results) + off, bytes, hipMemcpyDeviceToHost, g_host.copy_stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(g_host.copy_stream);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) {
        set_error("hipMemcpy D2H", e, __FILE__, __LINE__);
        die("hip_cal_align_score");
    }
    if (bgsa_hip_stream_faults(1) != 0) die("hip_cal_align_score");  // text set by stream_faults
}

void align_hip(char *ref, hip_read_t *read, int ref_len, int read_len, int word_num,
               int chunk_read_num, int result_index, hip_write_t *results, hip_data_t *dvdh_bit_mem)
{
    // One query against chunk_read_num groups; results land at results[result_index * HIP_V_NUM ...]
    // (reference original/BGSA_CPU/align_core.c:138-145).  With the chunk inside a resident bucket the query is
    // scored against the whole bucket once and every call copies its chunk out of that row (CachedRow above).
    // A worker of the reference's OpenMP grid asks for the same query chunk after chunk: its last row is kept
    // per thread and reused without the seam lock while nothing about it can have changed — same bucket
    // contents (range epoch), same range, same parameters, same query bytes.
    struct LastRow {
        uint64_t epoch = 0;
        const unsigned char *range_host = nullptr;
        size_t range_bytes = 0, group_bytes = 0;
        bgsa_hip_params_t params{};
        int read_len = 0, word_num = 0;
        std::string query;
        std::shared_ptr<RowBuf> scores;
        uint64_t range_fp = 0;     // fingerprint of the bucket's host bytes the row was scored from
        std::shared_ptr<const std::vector<unsigned char>> shadow;   // those bytes themselves where the policy keeps them (ResidentRange::shadow)
    };
    static thread_local LastRow last;
    if (chunk_read_num > 0 && ref && read && results && ref_len > 0) {
        bgsa_hip_params_t params;
        bgsa_hip_current_params(&params);
        {
            const unsigned char *peq_host = reinterpret_cast<const unsigned char *>(read);
            if (last.scores && last.epoch == g_range_epoch.load(std::memory_order_acquire) && last.read_len == read_len &&
                last.word_num == word_num && last.query.size() == static_cast<size_t>(ref_len) &&
                peq_host >= last.range_host && peq_host + last.group_bytes * chunk_read_num <= last.range_host + last.range_bytes &&
                (peq_host - last.range_host) % last.group_bytes == 0 && memcmp(&last.params, &params, sizeof params) == 0 &&
                memcmp(last.query.data(), ref, ref_len) == 0 &&
                // the bucket's bytes are still the ones this row was scored from (a rewrite the library was not told about
                // sends the call down the locked path, which uploads the range again).  Where the library kept those bytes the
                // comparison is exact and touches only the chunk this call was handed; otherwise the fingerprint reads the
                // registered range, which the contract keeps allocated while any thread is inside align_hip on it — and the
                // epoch is read again behind it: a release that raced with the read sends the call down the locked path
                (last.shadow ? memcmp(last.shadow->data() + (peq_host - last.range_host), peq_host, last.group_bytes * chunk_read_num) == 0
                             : (range_fingerprint(last.range_host, last.range_bytes) == last.range_fp &&
                                last.epoch == g_range_epoch.load(std::memory_order_acquire)))) {
                const size_t esz = result_elem_size(params.algo);
                memcpy(reinterpret_cast<char *>(results) + static_cast<size_t>(result_index) * HIP_V_NUM * esz,
                       last.scores->p + (peq_host - last.range_host) / last.group_bytes * HIP_V_NUM * esz,
                       static_cast<size_t>(chunk_read_num) * HIP_V_NUM * esz);
                g_row_fast_hits.fetch_add(1, std::memory_order_relaxed);
                return;
            }
        }
        seam_stats_at_exit();
        const uint64_t t_lock = now_ns();
        std::unique_lock<std::mutex> turn(g_seam);
        g_lock_wait_ns.fetch_add(now_ns() - t_lock, std::memory_order_relaxed);
        const size_t esz = result_elem_size(params.algo);
        const int w_host = host_words32(params.algo, read_len, params.k, word_num);
        if (w_host > 0 && g_host.auto_resident && seam_stream() == BGSA_HIP_OK) {
            const int w_dev = bgsa_hip_word_num(params.algo, ref_len, read_len, params.k);
            const size_t host_group_bytes = static_cast<size_t>(BGSA_CHAR_NUM) * w_host * 256;
            const unsigned char *peq_host = reinterpret_cast<const unsigned char *>(read);
            ResidentRange *r = resident_range(peq_host, host_group_bytes * chunk_read_num, w_host, w_dev, g_host.stream);
            if (r) {
                g_host.calls++;
                const size_t r_groups = r->bytes / host_group_bytes, first_group = (peq_host - r->host) / host_group_bytes;
                const size_t n_sub = r_groups * HIP_V_NUM, row_size = n_sub * esz, stride = static_cast<size_t>(ref_len) + 1;
                // rows per launch: the launch the host threads read, the one being scored behind it and the rows of the launches
                // before them that are still in some thread's hands must all have their slot in the row arena (512 slots of
                // the bucket's row size, fewer for rows over 2 MB): with a quarter of the slots per launch the arena was exactly
                // full, every new launch evicted rows that had not been read yet, and the reference's pipeline reported 94k GCUPS
                // where 100 rows gave 230-290k (1M subjects, 128 rows: 821 launches on a miss instead of 5;
                // profiles/r03_rowahead.txt).  A fifth of the slots at most.
                const size_t arena_slots = row_arena_slots_for(row_size);
                const int ahead = static_cast<int>(std::max<size_t>(1, std::min<size_t>(static_cast<size_t>(seam_row_ahead()), arena_slots / 5)));
                auto find_row = [&](const char *qrow_bytes) -> CachedRow * {
                    for (CachedRow &c : g_host.rows)
                        if (c.range_gen == r->gen && c.read_len == read_len && c.query.size() == static_cast<size_t>(ref_len) &&
                            memcmp(&c.params, &params, sizeof params) == 0 && memcmp(c.query.data(), qrow_bytes, ref_len) == 0)
                            return &c;
                    return nullptr;
                };
                // Rows a launch starting at `first` may take: consecutive rows of the malloc_mem() block that holds it,
                // each closed by its '\n', none of them scored already.
                auto run_length = [&](const char *first, int limit) {
                    const uint64_t t_scan = now_ns();
                    struct Done { uint64_t t; ~Done() { g_scan_ns.fetch_add(now_ns() - t, std::memory_order_relaxed); } } done_{t_scan};
                    const unsigned char *u = reinterpret_cast<const unsigned char *>(first);
                    const size_t extent = block_bytes_after(first);
                    int n = 0;
                    while (n < limit && static_cast<size_t>(n + 1) * stride <= extent && u[static_cast<size_t>(n) * stride + ref_len] == '\n' &&
                           !find_row(first + static_cast<size_t>(n) * stride))
                        n++;
                    return n;
                };
                auto wait_batch = [&](const std::shared_ptr<RowBatch> &bt) {
                    if (!bt || bt->waited) return;
                    if (!row_batch_finish(bt)) {
                        if (g_last_error.empty()) set_error_text("align_hip: scoring the query rows failed");
                        die("align_hip");
                    }
                    const uint64_t t_f = now_ns();
                    if (g_host.h_fault && g_host.h_fault[bt->region] != 0 && bgsa_hip_stream_faults(1) != 0) die("align_hip");
                    g_fault_ns.fetch_add(now_ns() - t_f, std::memory_order_relaxed);
                };
                // One launch: n_rows query rows starting at `first` against the whole bucket; kernel on the seam's stream,
                // the rows down on its copy stream straight into their arena slots.  Returns with everything queued.
                auto issue = [&](const char *first, int n_rows, bool rows_in_block) -> std::shared_ptr<RowBatch> {
                    const uint64_t t_issue = now_ns();
                    auto bt = std::make_shared<RowBatch>();
                    bt->region = static_cast<int>(g_host.batch_seq++ % HostSeam::kRowRegions);
                    bt->row_size = row_size;
                    const int reg = bt->region;
                    wait_batch(g_host.region_owner[reg]);     // the launch that used these device buffers before
                    bt->queries.reserve(static_cast<size_t>(n_rows) * stride);
                    for (int j = 0; j < n_rows; j++) {        // a lone row may not be followed by its terminator in the caller's memory
                        bt->queries.append(first + static_cast<size_t>(j) * stride, ref_len);
                        bt->queries.push_back('\n');
                    }
                    bt->full = n_rows == ahead;
                    bt->next = rows_in_block ? first + static_cast<size_t>(n_rows) * stride : nullptr;
                    // Room for the new rows: rows that have been served go before rows nobody has read yet, the least recently used
                    // first among each.  (By age alone the rows read ahead were the oldest: with the reference's host files built for
                    // 250-query blocks — threads spread over two and a half launches, each first touch chaining one more — the walk
                    // ran 300 rows ahead of rows that were then evicted unread: 500 launches on a miss, 90k GCUPS; 8 and 137k now.)
                    auto victim = [&]() {
                        size_t best = 0;
                        for (size_t j = 1; j < g_host.rows.size(); j++) {
                            const bool unread_j = static_cast<bool>(g_host.rows[j].scores->pending), unread_b = static_cast<bool>(g_host.rows[best].scores->pending);
                            if (unread_j != unread_b ? unread_b : g_host.rows[j].stamp < g_host.rows[best].stamp) best = j;
                        }
                        return best;
                    };
                    while (!g_host.rows.empty() && g_host.row_bytes + static_cast<size_t>(n_rows) * row_size > HostSeam::kRowCacheBytes) {
                        const size_t oldest = victim();
                        g_host.row_bytes -= g_host.rows[oldest].scores->size;
                        g_host.rows.erase(g_host.rows.begin() + oldest);
                    }
                    // ... and the cache never holds more rows than the arena has slots: a row without a slot has to be
                    // staged and copied, under the lock, while the next launch waits
                    for (long free_slots = row_arena_free(row_size); free_slots >= 0 && free_slots < n_rows && !g_host.rows.empty();
                         free_slots = row_arena_free(row_size)) {
                        const size_t oldest = victim();
                        g_host.row_bytes -= g_host.rows[oldest].scores->size;
                        g_host.rows.erase(g_host.rows.begin() + oldest);
                    }
                    if (row_arena_too_small(row_size)) {
                        // A bigger bucket than the arena was carved for.  The arena can be carved anew only with every slot
                        // back: the cache's own slot rows can be dropped, a slot that another host thread still holds in its
                        // LastRow (an idle OpenMP worker keeps its last row across buckets) cannot.  So count first and touch
                        // nothing unless the recarve will succeed; until then rows of this size take the heap / staged path.
                        // Rows on the heap — among them the rows of a launch of THIS bucket still in flight, when this is the
                        // read-ahead launch chained behind it — are never dropped here.
                        size_t droppable = 0;
                        bool own_in_cache = false;
                        for (const CachedRow &c : g_host.rows) {
                            if (c.scores->slot < 0) continue;
                            long users = c.scores.use_count();
                            if (last.scores == c.scores) { users--; own_in_cache = true; }
                            if (users == 1) droppable++;
                        }
                        if (last.scores && last.scores->slot >= 0 && !own_in_cache && last.scores.use_count() == 1) droppable++;
                        if (row_arena_slots_out() == droppable) {
                            for (size_t j = 0; j < g_host.rows.size();) {
                                if (g_host.rows[j].scores->slot >= 0) {
                                    g_host.row_bytes -= g_host.rows[j].scores->size;
                                    g_host.rows.erase(g_host.rows.begin() + j);
                                } else {
                                    j++;
                                }
                            }
                            if (last.scores && last.scores->slot >= 0) last.scores.reset();
                            (void)row_arena_recarve(row_size);
                        }
                    }
                    hipStream_t s = g_host.stream;
                    bool ok = g_host.reserve(&g_host.d_rowq_ring[reg], &g_host.cap_rowq_ring[reg], bt->queries.size() + 16) == BGSA_HIP_OK &&
                              g_host.reserve(&g_host.d_row_results[reg], &g_host.cap_row_results[reg], static_cast<size_t>(n_rows) * row_size) == BGSA_HIP_OK &&
                              hipEventCreateWithFlags(&bt->done, hipEventDisableTiming) == hipSuccess &&
                              hipMemcpyAsync(g_host.d_rowq_ring[reg], bt->queries.data(), bt->queries.size(), hipMemcpyHostToDevice, s) == hipSuccess &&
                              bgsa_hip_cal_align_score_ex(&params, static_cast<const char *>(g_host.d_rowq_ring[reg]),
                                                          static_cast<const hip_read_t *>(r->dev), g_host.d_row_results[reg], ref_len, read_len,
                                                          static_cast<int64_t>(n_sub), 0, n_rows, w_dev, nullptr, 0, s) == BGSA_HIP_OK &&
                              hipEventRecord(g_host.tile_done[0], s) == hipSuccess &&
                              hipStreamWaitEvent(g_host.copy_stream, g_host.tile_done[0], 0) == hipSuccess;
                    std::vector<std::shared_ptr<RowBuf>> bufs(n_rows);
                    for (int j = 0; ok && j < n_rows; j++) {
                        bool pinned = false;
                        bufs[j] = row_take(row_size, &pinned);
                        bufs[j]->pending = bt;
                        if (pinned) {
                            ok = hipMemcpyAsync(bufs[j]->p, static_cast<unsigned char *>(g_host.d_row_results[reg]) + static_cast<size_t>(j) * row_size,
                                                row_size, hipMemcpyDeviceToHost, g_host.copy_stream) == hipSuccess;
                        } else {
                            bt->staged.emplace_back(bufs[j].get(), static_cast<size_t>(j) * row_size);   // row_batch_finish brings it down
                        }
                    }
                    // the device's fault word travels behind the rows (reading it with a blocking copy would wait for the
                    // launch already queued behind this one, and the overlap would be gone)
                    if (ok && !g_host.h_fault)
                        ok = hipHostMalloc(reinterpret_cast<void **>(&g_host.h_fault), 64, hipHostMallocPortable) == hipSuccess;
                    unsigned *d_fault = ok ? device_fault_word() : nullptr;
                    ok = ok && d_fault &&
                         hipMemcpyAsync(&g_host.h_fault[reg], d_fault, sizeof(unsigned), hipMemcpyDeviceToHost, g_host.copy_stream) == hipSuccess;
                    ok = ok && hipEventRecord(bt->done, g_host.copy_stream) == hipSuccess;
                    if (!ok) {
                        if (g_last_error.empty()) set_error_text("align_hip: launching the query rows failed");
                        die("align_hip");
                    }
                    g_host.region_owner[reg] = bt;
                    for (int j = 0; j < n_rows; j++) {
                        CachedRow c;
                        c.range_gen = r->gen;
                        c.params = params;
                        c.read_len = read_len;
                        // the key is the bytes that were UPLOADED AND SCORED (the snapshot), never a second read of the caller's
                        // buffer: a reader thread may have refilled the rows ahead of the one asked for in between
                        c.query.assign(bt->queries.data() + static_cast<size_t>(j) * stride, static_cast<size_t>(ref_len));
                        c.scores = bufs[j];
                        c.stamp = ++g_host.clock;
                        g_host.row_bytes += row_size;
                        g_host.rows.push_back(std::move(c));
                    }
                    g_issue_ns.fetch_add(now_ns() - t_issue, std::memory_order_relaxed);
                    return bt;
                };
                // The calls are walking the buffer: keep one launch ahead of them.
                auto chain = [&](const std::shared_ptr<RowBatch> &bt) {
                    if (!bt || bt->chained) return;
                    bt->chained = true;
                    if (!bt->full || !bt->next || ahead < 2) return;
                    const int n = run_length(bt->next, ahead);
                    if (n > 0) {
                        issue(bt->next, n, true);
                        g_host.prefetch_launches++;
                        g_host.rows_ahead += static_cast<uint64_t>(n);
                    }
                };
                std::shared_ptr<RowBuf> row;
                if (CachedRow *c = find_row(ref)) {
                    c->stamp = ++g_host.clock;
                    row = c->scores;
                    g_host.row_hits++;
                    if (row->pending) {
                        std::shared_ptr<RowBatch> bt = row->pending;
                        const uint64_t t_row = now_ns();
                        chain(bt);           // may grow g_host.rows: `c` is not used below
                        wait_batch(bt);
                        g_row_ns.fetch_add(now_ns() - t_row, std::memory_order_relaxed);
                    }
                }
                if (!row) {
                    g_host.row_misses++;
                    const uint64_t t_row = now_ns();
                    // Read-ahead: the reference's grid walks the query buffer row after row (cal_cpu.c:63-84), so when this
                    // miss sits near the previous one inside a malloc_mem() block, the rows behind it are scored in the same
                    // launch and the launch behind that one is issued at once.  A row is only ever served for a query with
                    // the same bytes, so a wrong guess costs time, never results.
                    const unsigned char *uref = reinterpret_cast<const unsigned char *>(ref);
                    const bool near_last = g_host.last_miss && g_host.last_miss_stride == stride &&
                                           (uref > g_host.last_miss ? static_cast<size_t>(uref - g_host.last_miss)
                                                                    : static_cast<size_t>(g_host.last_miss - uref)) <= 64 * stride;
                    int n_rows = near_last ? run_length(ref, ahead) : 0;
                    const bool in_block = n_rows >= 1;
                    if (n_rows < 1) n_rows = 1;
                    g_host.last_miss = uref;
                    g_host.last_miss_stride = stride;
                    std::shared_ptr<RowBatch> bt = issue(ref, n_rows, in_block);
                    g_host.rows_ahead += static_cast<uint64_t>(n_rows - 1);
                    // hold the served row's buffer BEFORE the launch behind this one is issued: that issue may evict cache
                    // entries — with an arena smaller than assumed, or many threads holding slots, even unread rows of the
                    // batch just issued — and the buffer (and the copy landing in it) lives as long as this reference
                    CachedRow *c = find_row(ref);
                    if (!c) {
                        set_error_text("align_hip: the row just issued is not in the cache");
                        die("align_hip");
                    }
                    row = c->scores;
                    chain(bt);
                    wait_batch(bt);
                    g_row_ns.fetch_add(now_ns() - t_row, std::memory_order_relaxed);
                }
                row->pending.reset();   // served: nothing in flight behind this buffer any more
                last.epoch = g_range_epoch.load(std::memory_order_acquire);
                last.range_host = r->host;
                last.range_bytes = r->bytes;
                last.group_bytes = host_group_bytes;
                last.params = params;
                last.read_len = read_len;
                last.word_num = word_num;
                last.query.assign(ref, ref + ref_len);
                last.scores = row;
                last.range_fp = r->fp;
                last.shadow = r->shadow;
                turn.unlock();   // the copy needs no lock: the row is shared, immutable
                memcpy(reinterpret_cast<char *>(results) + static_cast<size_t>(result_index) * HIP_V_NUM * esz,
                       row->p + first_group * HIP_V_NUM * esz, static_cast<size_t>(chunk_read_num) * HIP_V_NUM * esz);
                return;
            }
        }
    }
    // One query against chunk_read_num groups: the coarse call with a 1-row query buffer and the
    // chunk as the whole bucket; results land at results[result_index * HIP_V_NUM ...]
    // (reference original/BGSA_CPU/align_core.c:138-145).
    const size_t esz = result_elem_size(g_algo);
    char *dst = reinterpret_cast<char *>(results) + static_cast<size_t>(result_index) * HIP_V_NUM * esz;
    hip_cal_align_score(ref, read, reinterpret_cast<hip_write_t *>(dst), ref_len, 1, read_len,
                        chunk_read_num * HIP_V_NUM, 0, 1, word_num, chunk_read_num, dvdh_bit_mem);
}

}  // extern "C"
