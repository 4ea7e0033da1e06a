/*
 * bgsa_hip.h — C ABI of the MI355X (gfx950) backend for BGSA's all-pairs bit-parallel global
 * alignment hot path.  Everything here is `extern "C"`, plain pointers and sizes.
 *
 * The library is a drop-in "ISA backend" in the sense of the reference's original/BGSA_<ARCH>
 * directories: it exports the same three seams every backend exports —
 *
 *     <arch>_handle_reads      (reference original/BGSA_CPU/global.h:24,  global.c:25-70)
 *     align_<arch>             (reference original/BGSA_CPU/align_core.h:8, align_core.c:19-148)
 *     <arch>_cal_align_score   (reference original/BGSA_CPU/cal.h:48,      cal_cpu.c:43-85)
 *
 * with <arch> = hip, plus the five scoring ints every align_core.c defines
 * (align_core.c:13-17) and the alphabet map (global.c:7-15).  Those entry points take HOST
 * buffers, exactly like the reference's.  Underneath them sits a device-resident layer
 * (bgsa_hip_*_dev) that a pipeline driver uses to keep subjects and scores in HBM between
 * calls, the way the reference's KNC backend keeps them on the coprocessor
 * (original/BGSA_KNC/cal_mic.c:86-154, 348-356).
 *
 * Error convention: the BGSA-surface functions are `void` and print + exit(1) on failure, like
 * the reference (file.c:13-16).  The bgsa_hip_* functions return 0 on success or a negative
 * BGSA_HIP_E* code and never exit.
 */
#ifndef BGSA_HIP_H
#define BGSA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- config.h-style constants (reference original/BGSA_CPU/config.h:6-27) ------------------ */
#define BGSA_CHAR_NUM 5            /* CHAR_NUM: A C G T N planes */
#define HIP_V_NUM 64               /* subjects per group = lanes of one wavefront */
#define HIP_WORD_SIZE 32           /* bits per bit-vector word; all 32 carry data (full_bits = 1) */
#define HIP_BANDED_WORD_SIZE 64    /* widest banded window (k <= 31), as in the reference */
#define HIP_MAX_ERROR 127          /* MAX_ERROR, banded/BGSA_CPU/config.h:19 */
typedef uint32_t hip_read_t;       /* element of the preprocessed Peq blocks */
typedef uint32_t hip_data_t;       /* element of the (unused) dvdh_bit_mem scratch */
typedef int16_t hip_write_t;       /* common_write_t of original/ (Myers, BitPAl) */
typedef int8_t hip_banded_write_t; /* common_write_t of banded/ */

/* seq_t, reference original/BGSA_CPU/global.h:9-16 (same field order and types). */
#ifndef BGSA_HAVE_SEQ_T
#define BGSA_HAVE_SEQ_T
typedef struct _seq_t {
    int len;
    int64_t size;
    int64_t count;
    int extra_size;
    int extra_count;
    char *content;
} seq_t;
#endif

/* ---- algorithm selection -------------------------------------------------------------------
 * The reference bakes the algorithm into the generated align_core.c; one backend directory =
 * one algorithm.  This library carries all three and switches at run time. */
enum {
    BGSA_ALGO_MYERS = 0,  /* unit-cost global edit distance, result = -distance (int16) */
    BGSA_ALGO_BANDED = 1, /* banded Myers filter, result = band distance or 127 (int8) */
    BGSA_ALGO_BITPAL = 2, /* BitPAl packed, match 2 / mismatch -3 / gap -5 unless bgsa_hip_select_scores() (int16) */
};

/* The five ints every align_core.c exports (reference original/BGSA_CPU/align_core.c:13-17);
 * bgsa_hip_select_algorithm() keeps them consistent with the selected algorithm. */
extern int match_score;
extern int mismatch_score;
extern int gap_score;
extern int dvdh_len;
extern int full_bits;
/* -k of banded/BGSA_CPU (banded/BGSA_CPU/global.h:43, main.c:43,62-64). */
extern int threshold;
/* Host threads used by hip_handle_reads (reference global `cpu_threads`, main.c:16,38). */
extern int cpu_threads;

/* Alphabet map (reference original/BGSA_CPU/global.c:7-15). */
extern uint32_t mapping_table[128];
void init_mapping_table(void);

/* 64-byte aligned host allocation (reference global.c:17-23). */
void *malloc_mem(uint64_t size);
void free_mem(void *mem);

int bgsa_hip_select_algorithm(int algo);
int bgsa_hip_current_algorithm(void);

/* BitPAl with other integer scores (match > mismatch, gap < 0).  The reference emits one
 * align_core.c per score set with its generator (`java -jar generator.jar -M -I -G`, README.md:26-82,
 * generator/.../BitPAlGenerator.java) and is rebuilt for it; this library is built with a list of
 * sets (`make -C bgsa_amd/csrc BITPAL_SETS="2,-3,-5 1,-3,-2 ..."`, gen_bitpal_sets.py) and picks the
 * kernels of the set that match_score / mismatch_score / gap_score name at call time.
 * bgsa_hip_select_scores() = select BGSA_ALGO_BITPAL and set the three ints, or
 * BGSA_HIP_EUNSUPPORTED (ints unchanged) if that set was not compiled in; scoring with ints that
 * name no compiled set fails the same way.  bgsa_hip_score_set() enumerates the compiled sets
 * (index 0 = the reference's committed 2/-3/-5); valu_per_word = VALU instructions per (row, 32
 * columns) of that set's kernel.  Any out-pointer may be NULL.
 * Like the generator (commonFactor, Main.java:213-267), scores with a common factor f run on the set
 * (M/f, I/f, G/f) and the result is multiplied by f: 4/-6/-10 needs only 2/-3/-5 compiled in.
 * A mismatch below 2*gap can never be taken (two gaps are cheaper), so such a set runs as its
 * mismatch = 2*gap instance: 1/-9/-2 needs 1/-4/-2.  A set that reduces to 0/-1/-1 — minus the edit
 * distance times f, the generator's isEdit case (Main.java:270-271) — runs, in global mode, on the Myers
 * kernels (10 instructions per word against 22) and needs no compiled BitPAl set at all.
 * bgsa_hip_select_scores(0, 1, 1) is the generator's `-m 1`: BGSA_ALGO_MYERS reporting +distance
 * instead of -distance (the same ints written directly while Myers is selected do the same). */
int bgsa_hip_select_scores(int match, int mismatch, int gap);
int bgsa_hip_score_set_count(void);
int bgsa_hip_score_set(int index, int *match, int *mismatch, int *gap, int *valu_per_word);

/* Global (default) or semi-global scoring — the generator's `-s` option (Configuration.isSemiGlobal).
 * The two generators orient it differently, and so does this library:
 *   BGSA_ALGO_BITPAL (BitPAlGenerator.java:2201-2218 first row, :78-116 last-row maximum): the QUERY is
 *     aligned end to end, subject overhangs before and after it are free; result = max over the last DP
 *     row.  Any compiled score set, any length.
 *   BGSA_ALGO_MYERS (MyersGenerator.java:56-223 genSemiGlobal): the SUBJECT is aligned end to end inside
 *     the query (D[0][y] = 0, result = -min over y of D[slen][y]); any length (generated-asm kernels:
 *     resident Peq planes up to 800 bp, code planes up to 1024 bp, column blocks beyond).
 *   BGSA_ALGO_BANDED: not defined, BGSA_HIP_EUNSUPPORTED.
 * Process-global like the score ints. */
enum { BGSA_ALIGN_GLOBAL = 0, BGSA_ALIGN_SEMIGLOBAL = 1 };
int bgsa_hip_select_alignment(int mode);
int bgsa_hip_current_alignment(void);

/* Every parameter the scoring call reads, as one value.  The reference keeps them in globals of the
 * backend (the five ints, `threshold`); bgsa_hip_cal_align_score_dev() and the host seams snapshot those
 * globals once per call, and callers that must not share process-wide state (two pipelines with
 * different scores in one process) pass their own through the *_ex entry points instead. */
typedef struct bgsa_hip_params {
    int algo;                 /* BGSA_ALGO_* */
    int alignment;            /* BGSA_ALIGN_* */
    int match, mismatch, gap; /* BitPAl scores; Myers: (0,1,1) = +distance, anything else -distance */
    int k;                    /* banded threshold */
} bgsa_hip_params_t;
int bgsa_hip_current_params(bgsa_hip_params_t *out);   /* the process-global selection, as of now */

/* word_num of this backend's layouts: Myers / BitPAl ceil(subject_len / 32) (cal_cpu.c:252-253 with
 * full_bits); banded ceil(subject_len / 32) + 3 words of the offset match string (the host seams also
 * accept the reference's own banded value, banded/BGSA_CPU/cal_cpu.c:253-254 — see hip_handle_reads). */
int bgsa_hip_word_num(int algo, int query_len, int subject_len, int k);
/* hip_read_t elements per group of HIP_V_NUM subjects = BGSA_CHAR_NUM * word_num * HIP_V_NUM. */
size_t bgsa_hip_group_words(int algo, int word_num, int k);

/* ---- BGSA backend surface (host buffers) ---------------------------------------------------
 * align_hip / hip_cal_align_score share one set of device mirrors inside the library and take turns on
 * it: they may be called from several host threads, as the reference's OpenMP loop calls align_<arch>
 * (cal_cpu.c:63-84), but the calls are serialised; each call reads the global selection (algorithm,
 * scores, alignment mode, threshold) once, under that lock.  bgsa_hip_release_workspace() frees the
 * mirrors.
 *
 * Resident buckets.  hip_handle_reads() remembers the host range it filled; the first scoring call on
 * that range uploads it, later calls on it (the reference scores 100 queries per call against the same
 * bucket, cal_cpu.c:363-401) reuse the device copy until hip_handle_reads() writes the range again —
 * the KNC backend's `nocopy ... RETAIN` (BGSA_KNC/cal_mic.c:348-356).  The query buffer is uploaded
 * only when its bytes changed.  A host that fills Peq words by other means must either call
 * bgsa_hip_bucket_resident() after every change or switch the mechanism off with
 * bgsa_hip_set_auto_resident(0) (upload on every call, the stateless contract of the CPU backends).
 * malloc_mem() hands out page-locked memory for large blocks, so every buffer of the reference's
 * pipeline (cal_cpu.c:206-267) moves at full PCIe rate. */
int bgsa_hip_set_auto_resident(int on);
/* THE CONTRACT of a resident range: between the hip_handle_reads() / bgsa_hip_bucket_resident() call that registered it
 * and its release, its host bytes change ONLY through those two calls, and the memory stays allocated while any thread is
 * inside a scoring call on it (the reference's pipeline does both: cal_cpu.c:363-401 fills a bucket with
 * cpu_handle_reads and frees it after its compute threads have joined).  The host range is the truth; the device copy
 * is a cache of it.
 * What the library does for a caller that breaks the contract (a memcpy of a saved bucket over the registered buffer, a
 * host that patches Peq words in place):
 *   - ranges up to 8 MiB (every range in strict mode): EXACT.  The library keeps the host bytes it uploaded and every
 *     scoring call — hip_cal_align_score, and align_hip on its locked and its lock-free path — compares the bytes it is
 *     about to use against them (memcmp, < 1 ms for a whole 8 MiB bucket); any difference uploads the range again and
 *     drops the rows cached from the old content;
 *   - larger ranges: BEST EFFORT.  Every scoring call fingerprints the range (66 cache lines: first, last and a
 *     golden-ratio sequence of positions between them) and uploads it again when the fingerprint differs from the one
 *     taken at upload.  A rewrite that leaves all sampled lines unchanged (a few groups patched in place) is not seen.
 * bgsa_hip_set_strict_resident(1) / BGSA_HIP_STRICT_RESIDENT=1: the exact check for every range (costs a host copy of
 * each bucket and a memcmp per call); (0): the default above; (-1) / BGSA_HIP_STRICT_RESIDENT=-1: the fingerprint alone
 * (measurement).  Switching the mode re-uploads every range on its next use and drops the cached rows.
 * bgsa_hip_stale_ranges() counts the re-uploads either check caused.  (SURVEY 8(b) "Ownership";
 * BGSA_KNC/cal_mic.c:348-356.) */
int bgsa_hip_set_strict_resident(int on);
int bgsa_hip_stale_ranges(uint64_t *count);
/* host_peq[0 .. bytes) holds whole groups in the library's own layout with word_num words. */
int bgsa_hip_bucket_resident(const hip_read_t *host_peq, size_t bytes, int word_num);
int bgsa_hip_bucket_release(const hip_read_t *host_peq);   /* NULL: all of them */
/* Counters of the seams since process start (any pointer may be NULL). */
int bgsa_hip_seam_stats(uint64_t *calls, uint64_t *peq_uploads, uint64_t *peq_upload_bytes);
/* align_hip's row cache: the reference's grid calls align_<arch> once per (query, chunk of ~27 groups)
 * (cal_cpu.c:63-84).  With the chunk inside a resident bucket, the first call for a query scores it
 * against the whole bucket in one launch and keeps that row (page-locked host memory, at most 1 GiB of
 * rows, least recently used first out); the other calls for the same query, bucket and parameters copy
 * their chunk out of it.  When the misses walk a malloc_mem() query buffer row after row, the rows behind
 * the requested one that no launch has scored yet join its launch and the next launch is issued ahead of the
 * calls (BGSA_HIP_ROW_AHEAD, default 32 rows per launch, 1 = off); a row is only served for a query with the
 * same bytes.  hip_cal_align_score scores its
 * block in BGSA_HIP_SEAM_TILES (default 8) query tiles and copies tile t out while tile t+1 runs.
 * hits / misses (= launches) since process start. */
int bgsa_hip_row_cache_stats(uint64_t *hits, uint64_t *misses);

/* ASCII rows -> Peq blocks, layout [group][char 0..4][word][lane 0..63]
 * (replaces cpu_handle_reads, reference original/BGSA_CPU/global.c:25-70; for BGSA_ALGO_BANDED
 * the words hold the match string offset by threshold+1 bits, which is what the windows of
 * banded/BGSA_CPU/global.c:25-84 + align_core.c:35-62 slide over).  result_reads must be zeroed by the caller (cal_cpu.c:273)
 * and read_count must be a multiple of HIP_V_NUM (file.c:84-112 pads with all-'N' reads).
 * word_num = bgsa_hip_word_num(); for BGSA_ALGO_BANDED also the reference's own value,
 * (len - h + 63)/64 + 1 words of its 64-bit cpu_read_t (banded/BGSA_CPU/cal_cpu.c:253-254): that
 * buffer holds the same bit string without the trailing zero words, and hip_cal_align_score / align_hip
 * called with the same word_num re-pitch it on upload — so banded/BGSA_CPU's host files work unchanged. */
void hip_handle_reads(seq_t *read_seq, hip_read_t *result_reads, int word_num,
                      int64_t read_start, int64_t read_count);

/* One mapped query against chunk_read_num consecutive groups
 * (replaces align_cpu, reference original/BGSA_CPU/align_core.c:19-148).
 * results[(result_index + k) * HIP_V_NUM + lane].  dvdh_bit_mem is accepted and ignored. */
void align_hip(char *ref, hip_read_t *read, int ref_len, int read_len, int word_num,
               int chunk_read_num, int result_index, hip_write_t *results,
               hip_data_t *dvdh_bit_mem);

/* All queries [ref_start, ref_end) x all read_count subjects of the bucket, row-major
 * [ref][read] results (replaces cpu_cal_align_score, reference cal_cpu.c:43-85).
 * For BGSA_ALGO_BANDED align_results is really hip_banded_write_t*. */
void hip_cal_align_score(char *content, hip_read_t *preprocess_reads, hip_write_t *align_results,
                         int ref_len, int ref_count, int read_len, int read_count, int ref_start,
                         int ref_end, int word_num, int chunk_read_num, hip_data_t *dvdh_bit_mem);

int bgsa_hip_release_workspace(void);

/* ---- device-resident layer ----------------------------------------------------------------- */

#define BGSA_HIP_OK 0
#define BGSA_HIP_EINVAL (-1)      /* bad argument (null pointer, negative size, misaligned count) */
#define BGSA_HIP_EUNSUPPORTED (-2) /* length / threshold outside what the kernels cover */
#define BGSA_HIP_EHIP (-3)        /* a HIP runtime call failed; text via bgsa_hip_last_error() */

const char *bgsa_hip_last_error(void);
int bgsa_hip_device_count(void);
int bgsa_hip_set_device(int device);

/* Plain device memory helpers so C hosts need no HIP headers. */
int bgsa_hip_malloc(void **dptr, size_t bytes);
/* free and total memory of the current device (hipMemGetInfo); either pointer may be NULL */
int bgsa_hip_mem_info(size_t *free_bytes, size_t *total_bytes);
int bgsa_hip_free(void *dptr);
/* Page-locked host memory (full-rate asynchronous copies for the pipeline driver). */
int bgsa_hip_malloc_host(void **hptr, size_t bytes);
int bgsa_hip_free_host(void *hptr);
int bgsa_hip_memcpy_h2d(void *dst, const void *src, size_t bytes, void *stream);
int bgsa_hip_memcpy_d2h(void *dst, const void *src, size_t bytes, void *stream);
int bgsa_hip_memset(void *dst, int value, size_t bytes, void *stream);
int bgsa_hip_stream_create(void **stream);   /* on the current device */
int bgsa_hip_stream_destroy(void *stream);
int bgsa_hip_stream_synchronize(void *stream);
/* Events: per-device kernel times for the dynamic balancing of the command line (BGSA_KNC/global.c:120-168). */
int bgsa_hip_event_create(void **event);
int bgsa_hip_event_destroy(void *event);
int bgsa_hip_event_record(void *event, void *stream);
int bgsa_hip_event_synchronize(void *event);
int bgsa_hip_event_elapsed_ms(void *start, void *stop, float *ms);
int bgsa_hip_stream_wait_event(void *stream, void *event);

/* Subject preprocess ON the GPU: d_rows = read_count rows of (len+1) ASCII bytes in device
 * memory -> d_peq in the layout above.  read_count must be a multiple of HIP_V_NUM.
 * `avail_bytes` = readable bytes of d_rows (banded over-read guard).  Overwrites d_peq. */
int bgsa_hip_handle_reads_dev(int algo, const char *d_rows, int64_t avail_bytes, int len,
                              int64_t read_count, int word_num, int k, hip_read_t *d_peq,
                              void *stream);

/* Query bytes -> 0..4 in place on the device ('\n' kept), get_ref_from_file's map
 * (reference original/BGSA_CPU/file.c:134-139). */
int bgsa_hip_map_queries_dev(char *d_content, int64_t bytes, void *stream);

/* Scratch the hot path needs for n_queries queries of ref_len characters against subjects of
 * read_len characters: the queries re-packed into 8-byte-aligned code streams the kernels fetch
 * through the scalar cache, plus, for subjects too long for the register-resident kernels (Myers
 * > 1024 bp, BitPAl beyond the plain widths of the selected score set, > 352 bp for 2/-3/-5), the
 * per-wave carry words of the column-block kernels, plus the task counter of the launches whose waves
 * take their (subject group, query tile) tasks from it.  Depends on the selected score set.  A workspace
 * belongs to ONE launch at a time: launches that may overlap (two streams) need one each. */
size_t bgsa_hip_workspace_bytes(int algo, int ref_len, int read_len, int n_queries);

/* The hot path.  d_content = mapped query rows, stride ref_len+1 (reference cal_cpu.c:78);
 * d_peq = Peq blocks of read_count subjects; d_results = [ref_end-ref_start][read_count]
 * (int16, or int8 for banded).  d_workspace = caller-owned device scratch of at least
 * bgsa_hip_workspace_bytes(algo, ref_len, read_len, ref_end-ref_start) bytes, or NULL to let the library
 * keep a grow-only scratch of its own per (device, stream) (allocates on first use: not graph-capture
 * safe; calls that use it are serialised among themselves).  word_num must be bgsa_hip_word_num().
 * Reads the process-global selection (scores, alignment mode) once, at entry.
 * All pointers are device pointers; asynchronous on `stream`. */
int bgsa_hip_cal_align_score_dev(int algo, const char *d_content, const hip_read_t *d_peq,
                                 void *d_results, int ref_len, int read_len, int64_t read_count,
                                 int ref_start, int ref_end, int word_num, int k,
                                 void *d_workspace, size_t workspace_bytes, void *stream);

/* The same with the parameters passed explicitly instead of read from the process globals. */
size_t bgsa_hip_workspace_bytes_ex(const bgsa_hip_params_t *params, int ref_len, int read_len, int n_queries);
int bgsa_hip_cal_align_score_ex(const bgsa_hip_params_t *params, const char *d_content, const hip_read_t *d_peq,
                                void *d_results, int ref_len, int read_len, int64_t read_count,
                                int ref_start, int ref_end, int word_num,
                                void *d_workspace, size_t workspace_bytes, void *stream);

/* Stream faults.  The kernels walk each query as a packed code stream (below) under a window budget; a
 * wave whose stream ends without an END token, or holds a byte that is no token, leaves its loop and
 * raises a bit in a sticky per-device word instead of storing a score.  A well-formed stream cannot do
 * either: a set bit means the stream bytes were damaged between the packer and the row loop, and the
 * scores of that call are not to be trusted.  bgsa_hip_stream_faults() returns the word of the current
 * device (0 = clean; call after synchronising the streams that scored) and optionally clears it; the
 * host-buffer seams check it after every call and fail loudly.
 * bgsa_hip_debug_inject_stream_fault(kind): tests only — the next scoring launch of this process has
 * its first stream overwritten with REFILL tokens (1) or with a byte that is no token (2). */
#define BGSA_HIP_FAULT_BUDGET 1
#define BGSA_HIP_FAULT_CODE 2
int bgsa_hip_stream_faults(int clear);
int bgsa_hip_debug_inject_stream_fault(int kind);

/* The sustained shader clock while other launches run (bench.py's `clock` object): n_probes one-wave workgroups
 * (8 cover the 8 XCDs) are started on a stream of the library's own and sleep, reading the shader-clock
 * and the constant reference-clock counters, until bgsa_hip_clock_probe_stop() or until max_ms have passed — they
 * never outlive that bound.  caller_stream = the stream whose kernels are to be observed: start() picks a stream of its own
 * on which the probes are SEEN to run beside that stream (HIP shares hardware queues between streams; probes on the caller's
 * queue would hold its kernels back), or fails with BGSA_HIP_EUNSUPPORTED.  stop() returns per probe the clock in MHz over its lifetime and the XCD it ran on, and
 * the longest probe lifetime in seconds.  Measurement only: nothing in the scoring path depends on it. */
int bgsa_hip_clock_probe_start(int n_probes, unsigned max_ms, void *caller_stream);
int bgsa_hip_clock_probe_stop(double *mhz, int *xcc, int cap, int *n_out, double *seconds);

/* Introspection (host only, no GPU): the packed code stream the kernels walk for one mapped query
 * row, 8-byte windows of 7 tokens + REFILL.  Myers / BitPAl: codes 0..4 = row of that character
 * class, 5 = END, 6 = REFILL; for BGSA_ALGO_MYERS, k = -1 selects the stream of the column-block
 * kernel (subjects > 1024 bp: a CARRY token, code 7, in front of every 32nd row) and k = -2 the
 * two-rows-per-token stream of the <= 64 bp kernels (codes as for banded below, without EVENT).
 * BGSA_ALGO_BANDED (threshold k): 0..24 = two rows of classes a, b as 5*a + b, 25..29 = one row,
 * 30 = END, 31 = REFILL, 63 = EVENT + argument byte (1 reset the error count, 2 advance the match
 * words, 4 test the limit, 8 latch the reject mask, 16 re-anchor the band (BGSA_BANDED_IMPL=p only), 32 cut the next
 * one-word window (the default kernels for k <= 12)).
 * Writes at most `cap` bytes to dst (may be NULL) and returns the stream length in bytes. */
int bgsa_hip_query_stream(int algo, const char *mapped_row, int ref_len, int k, unsigned char *dst, int cap);

/* Queries a wave scores per load of its subject block in this thread's last scoring launch (0: none yet).  The launch's
 * HBM traffic follows from it: ceil(queries / tile) x block bytes + the scores (bench.py: traffic_model). */
int bgsa_hip_last_query_tile(void);

/* Name of the kernel the previous call would launch for these shapes (for profiles/bench). */
const char *bgsa_hip_kernel_name(int algo, int word_num);

#ifdef __cplusplus
}
#endif
#endif /* BGSA_HIP_H */
