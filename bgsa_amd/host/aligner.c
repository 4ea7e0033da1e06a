/*
 * aligner.c — the BGSA `aligner` command line on top of libbgsa_hip.so.
 *
 * Same contract as the reference's pipeline driver (original/BGSA_CPU/main.c:36-106 +
 * cal_cpu.c:121-476), written from scratch in C against the C ABI of include/bgsa_hip.h:
 *
 *   ./aligner -q <query file> -d <database file> -f <result file> [-N host threads]
 *             [-k threshold] [-a myers|banded|bitpal] [-n gpus] [-g first gpu | g0,g1,...]
 *             [-R ratio file] [-D] [-M match -I mismatch -G gap] [-s]
 *
 *   * input files: one sequence per line, all of one length (what `convert -f/-q` produces);
 *   * queries are mapped A,C,G,T,N -> 0..4 (file.c:117-140); the database is cut into read
 *     buckets of at most READ_BUCKET_SIZE bytes, every bucket but the last a multiple of
 *     HIP_V_NUM reads, the last padded up with all-'N' reads (file.c:44-115);
 *   * output: `result` = for every (read bucket, block of REF_BUCKET_COUNT queries) the
 *     row-major [queries][reads] scores (thread.c:150-160); `result.info` = int bucket count,
 *     int device count, int64 query count, then per bucket int64 reads + int padded reads
 *     (cal_cpu.c:247-249,350-351) — exactly what the reference's `convert -r` reads;
 *   * the report printed at the end keeps the reference's lines and its two GCUPS figures
 *     (cal_cpu.c:459-475): "cal" = time spent scoring (GPU time of the scoring launches, the reference's
 *     cal_total_times around its compute call, cal_cpu.c:111-118), "Total" = wall.
 *
 * The subject bucket lives in HBM: rows are uploaded once, preprocessed on the GPU, every launch scores a
 * few query blocks, and — whenever they fit — the scores of ALL queries against the bucket stay in HBM
 * too: the launches are queued back to back, the copy-out runs block by block on a second stream, and a
 * writer thread drains a ring of pinned buffers to disk (the reference's input/output pthreads,
 * thread.c:35-171, reduced to the one that matters here).  The GPU never waits for the file.
 *
 * With -n > 1 every bucket is cut into one contiguous slice of subject groups per GPU (the KNC
 * backend's dispatch by ratio, BGSA_KNC/global.c:374-431, -R gives the ratios), each GPU scores
 * all queries against its slice, and a query block is written as device 0's [queries][reads]
 * tile, then device 1's, ... with the per-device counts recorded in `.info`
 * (BGSA_KNC/cal_mic.c:475-476,535-536).  -D re-estimates the ratios after every bucket from the GPU time
 * each device spent on its slice (HIP events around its kernels), with the reference's rule
 * (adjust_device_ratio3, BGSA_KNC/global.c:120-168): device 0 is the unit, device i's ratio is scaled by
 * t0/ti and averaged over the rounds so far with round r weighing r.  One host thread drives all GPUs: launches and copies are
 * asynchronous, and each GPU alternates between two streams so that the copy-out of block i
 * overlaps the kernel of block i+1.
 */
#define _GNU_SOURCE
#include <fcntl.h>
#include <getopt.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/time.h>
#include <unistd.h>

#include "bgsa_hip.h"

#define READ_BUCKET_SIZE 114857600LL /* original/BGSA_CPU/config.h:6 */
#define REF_BUCKET_COUNT 100         /* original/BGSA_CPU/config.h:13 */
#define RING 4
#define MAX_DEV 16
/* Query blocks per scoring launch.  The file holds, per read bucket, block after block of REF_BUCKET_COUNT queries — with one
 * device that is simply the bucket's row-major [queries][reads], so several blocks can be scored by ONE launch and go down as one
 * piece; with several devices every block is still written as device 0's tile, device 1's, ... (one copy per block and device).
 * A launch of 100 queries x 760k subjects is 5.8 rounds of workgroups on the chip and its tail shows: per-block launches spent
 * 1.19 s of GPU time on 10k x 1M where one launch takes 1.06 s; four blocks per launch: BGSA_LAUNCH_BLOCKS (1..16). */
#define LAUNCH_BLOCKS_DEFAULT 4

static double now(void)
{
    struct timeval tv;
    gettimeofday(&tv, NULL);
    return tv.tv_sec + tv.tv_usec * 1e-6;
}

static void die(const char *what)
{
    printf("Error - %s: %s\n", what, bgsa_hip_last_error());
    exit(1);
}
#define CK(call) do { if ((call) != BGSA_HIP_OK) die(#call); } while (0)

static FILE *open_or_die(const char *name, const char *mode)
{
    FILE *fp = fopen(name, mode);
    if (!fp) {
        printf("Error - can't open or create file: %s\n", name);
        exit(1);
    }
    return fp;
}

static int64_t file_size(const char *name)
{
    struct stat st;
    if (stat(name, &st) != 0) {
        printf("Error - can't open or create file: %s\n", name);
        exit(1);
    }
    return (int64_t)st.st_size;
}

/* ---- writer thread: drains score blocks in order ------------------------------------------ */
typedef struct {
    void *host[RING];
    size_t bytes[RING];
    int state[RING]; /* 0 free, 1 filled, 2 handed out and being filled */
    int head, tail, done;
    int n_slots;      /* slots in use (<= RING) */
    int fd;           /* the result file; blocks are written at their own offsets by several threads at once */
    int64_t offset;   /* file offset of the next block */
    char *map;        /* the whole result file mapped MAP_SHARED (NULL: the pwrite() path) */
    int falloc;       /* BGSA_WRITER_MODE=mmap+falloc: a thread of its own allocates the file's pages ahead of the copies */
    int64_t file_bytes, allocated;   /* ... the file's final size, and how far that thread has come (under `lock`) */
    int n_writers;
    double seconds;
    pthread_mutex_t lock;
    pthread_cond_t cond;
} ring_t;

/* The result file is what bounds Total GCUPS: at 10k x 1M Myers the GPU produces 19 GB/s of scores.
 * pwrite() (the default): one stream into a tmpfs page cache takes 6.4 GB/s, and more streams make it WORSE — 4 / 8 / 16
 * threads measured 5.0 / 5.9 / 3.4 s against 3.1 s with one (round 2) — because write() holds the file's inode lock for
 * the whole call: concurrent writes to one file take turns, and pay for the hand-over.
 * BGSA_WRITER_MODE=mmap (round 3, measured and NOT the default): the file is sized up front (its final size is known from
 * the bucket plan), mapped MAP_SHARED, and a block is copied into the mapping by BGSA_WRITER_THREADS threads (default 8),
 * each faulting in its own pages — page faults take no inode lock.  10k x 1M into /dev/shm (profiles/r03_writer.txt):
 * 5.08 / 4.57 / 5.02 / 5.51 s of writing with 1 / 4 / 8 / 16 threads against 3.51 s for the single pwrite() stream: a
 * page of a shared mapping arrives through a fault (about a microsecond each, five million of them), zeroed first, and
 * the faults of one file do not scale over threads either.  (`mmap+populate` faults each slice in with one
 * madvise(MADV_POPULATE_WRITE) instead; `mmap+falloc` has a thread of its own fallocate() the file 256 MiB at a time ahead of
 * the copies, so that they meet pages that exist: 5.18 / 4.65 / 2.75 / 3.57 s of writing with 1 / 4 / 8 / 16 copy threads and
 * 3.95 s in total at best against 3.44-3.57 s for pwrite() on that box — a write fault per 4 KiB page of a shared mapping
 * costs a microsecond even when the page is there.)  dd from /dev/zero into one /dev/shm file runs at 7.7-8.4 GB/s on these
 * boxes and into 4 / 8 / 16 files at once at 30 / 52 / 72 GB/s (profiles/r03_writer.txt): tmpfs scales over files, not
 * inside one.  So the page cache of one tmpfs file takes about 6 GB/s from a copying writer whichever way the bytes
 * arrive, and Total GCUPS of a 20 GB result stays near 60-70k while the kernels (cal) run at 200k+. */
typedef struct {
    int fd;
    const char *src;
    size_t bytes;
    int64_t offset;
    int failed;
} slice_t;

static int g_populate = 0;   /* BGSA_WRITER_MODE=mmap+populate: fault the slice's pages in with one madvise() before the copy */
static void *copy_slice(void *arg)
{
    slice_t *w = (slice_t *)arg;   /* fd < 0: `offset` holds the destination address */
    char *dst = (char *)(intptr_t)w->offset;
#ifdef MADV_POPULATE_WRITE
    if (g_populate) {
        char *lo = (char *)(((uintptr_t)dst + 4095) & ~(uintptr_t)4095), *hi = (char *)(((uintptr_t)dst + w->bytes) & ~(uintptr_t)4095);
        if (hi > lo) (void)madvise(lo, (size_t)(hi - lo), MADV_POPULATE_WRITE);
    }
#endif
    memcpy(dst, w->src, w->bytes);
    return NULL;
}

static void *write_slice(void *arg)
{
    slice_t *w = (slice_t *)arg;
    size_t done = 0;
    while (done < w->bytes) {
        ssize_t n = pwrite(w->fd, w->src + done, w->bytes - done, w->offset + (int64_t)done);
        if (n <= 0) { w->failed = 1; return NULL; }
        done += (size_t)n;
    }
    return NULL;
}

/* BGSA_WRITER_MODE=mmap+falloc: the pages of the result file are allocated by fallocate(), 256 MiB at a time, from a thread
 * of its own that runs ahead of the copies (the file's size is known before the first block is scored); the copies then
 * meet pages that exist and only have to map them.  fallocate() holds the inode lock like write(), but it does not copy. */
static void *falloc_main(void *arg)
{
    ring_t *r = (ring_t *)arg;
    const int64_t step = (int64_t)256 << 20;
    for (int64_t at = 0; at < r->file_bytes; at += step) {
        const int64_t len = r->file_bytes - at < step ? r->file_bytes - at : step;
        if (fallocate(r->fd, 0, (off_t)at, (off_t)len) != 0) { /* not supported here: the copies fault their pages in themselves */
            pthread_mutex_lock(&r->lock);
            r->allocated = r->file_bytes;
            pthread_cond_broadcast(&r->cond);
            pthread_mutex_unlock(&r->lock);
            return NULL;
        }
        pthread_mutex_lock(&r->lock);
        r->allocated = at + len;
        pthread_cond_broadcast(&r->cond);
        pthread_mutex_unlock(&r->lock);
    }
    return NULL;
}

static void *writer_main(void *arg)
{
    ring_t *r = (ring_t *)arg;
    for (;;) {
        pthread_mutex_lock(&r->lock);
        while (r->state[r->tail] != 1 && !r->done) pthread_cond_wait(&r->cond, &r->lock);
        if (r->state[r->tail] != 1) {
            pthread_mutex_unlock(&r->lock);
            return NULL;
        }
        int slot = r->tail;
        if (r->falloc)   /* the block's pages first */
            while (r->allocated < r->offset + (int64_t)r->bytes[slot] && r->allocated < r->file_bytes) pthread_cond_wait(&r->cond, &r->lock);
        pthread_mutex_unlock(&r->lock);
        double t0 = now();
        {
            slice_t part[16];
            pthread_t th[16];
            int n = r->n_writers;
            if (r->bytes[slot] < ((size_t)4 << 20)) n = 1; /* small blocks: not worth the threads */
            const size_t each = (r->bytes[slot] / (size_t)n + 4095) & ~(size_t)4095;
            void *(*fn)(void *) = r->map ? copy_slice : write_slice;
            int used = 0;
            for (size_t at = 0; at < r->bytes[slot]; at += each, used++) {
                part[used].fd = r->map ? -1 : r->fd;
                part[used].src = (const char *)r->host[slot] + at;
                part[used].bytes = r->bytes[slot] - at < each ? r->bytes[slot] - at : each;
                part[used].offset = r->map ? (int64_t)(intptr_t)(r->map + r->offset + (int64_t)at) : r->offset + (int64_t)at;
                part[used].failed = 0;
                if (used > 0) pthread_create(&th[used], NULL, fn, &part[used]);
            }
            if (used > 0) fn(&part[0]);
            for (int i = 1; i < used; i++) pthread_join(th[i], NULL);
            for (int i = 0; i < used; i++)
                if (part[i].failed) {
                    printf("Error - short write to the result file\n");
                    exit(1);
                }
            r->offset += (int64_t)r->bytes[slot];
        }
        r->seconds += now() - t0;
        pthread_mutex_lock(&r->lock);
        r->state[slot] = 0;
        r->tail = (slot + 1) % r->n_slots;
        pthread_cond_broadcast(&r->cond);
        pthread_mutex_unlock(&r->lock);
    }
}

static int ring_acquire(ring_t *r)
{
    pthread_mutex_lock(&r->lock);
    while (r->state[r->head]) pthread_cond_wait(&r->cond, &r->lock);
    int slot = r->head;
    r->state[slot] = 2;
    r->head = (slot + 1) % r->n_slots;
    pthread_mutex_unlock(&r->lock);
    return slot;
}

static void ring_publish(ring_t *r, int slot, size_t bytes)
{
    pthread_mutex_lock(&r->lock);
    r->bytes[slot] = bytes;
    r->state[slot] = 1;
    pthread_cond_broadcast(&r->cond);
    pthread_mutex_unlock(&r->lock);
}

static void usage(void)
{
    printf("\nUsage: ./aligner [options]\n\nCommandline options:\n\n");
    printf("  -q <arg>\n\t Query file (one sequence per line; convert FASTA/FASTQ with ./convert). \n\n");
    printf("  -d <arg>\n\t Database file (same format). \n\n");
    printf("  -f <arg>\n\t Alignment result file. \n\n");
    printf("  -N <arg>\n\t Number of host threads. \n\n");
    printf("  -k <arg>\n\t Filter threshold (banded). \n\n");
    printf("  -a <arg>\n\t Algorithm: myers (default), banded, bitpal. \n\n");
    printf("  -M <arg> -I <arg> -G <arg>\n\t BitPAl match / mismatch / gap scores (a set the library was built with;\n\t default 2 / -3 / -5). Implies -a bitpal. \n\n");
    printf("  -s\n\t Semi-global: BitPAl - query end to end inside the subject; Myers - subject end to end\n\t inside the query (the generator's two orientations). \n\n");
    printf("  -n <arg>\n\t Number of GPUs. Default 1. \n\n");
    printf("  -g <arg>\n\t First GPU index, or a comma separated list of GPU indices. Default 0. \n\n");
    printf("  -R <arg>\n\t File with one work ratio per GPU (one number per line). Default: equal. \n\n");
    printf("  -D\n\t Dynamic ratios: re-balance the GPUs after every database bucket from their measured times. \n\n");
    exit(1);
}

/* ---- input thread: reads the next bucket while the GPUs score the current one (thread.c:35-113) */
typedef struct {
    FILE *fp;
    char *rows;
    int64_t want, row, count;
    int read_len, extra;
    double seconds;
} read_job_t;

static void *read_bucket(void *arg)
{
    read_job_t *j = (read_job_t *)arg;
    const double t0 = now();
    size_t n = fread(j->rows, 1, (size_t)(j->want * j->row), j->fp);
    if ((int64_t)n < j->want * j->row) j->rows[n++] = '\n'; /* file without a final newline */
    j->count = j->want;
    j->extra = 0;
    while (j->count % HIP_V_NUM) { /* pad the last bucket with all-'N' reads (file.c:84-112) */
        memset(j->rows + j->count * j->row, 'N', (size_t)j->read_len);
        j->rows[j->count * j->row + j->read_len] = '\n';
        j->count++;
        j->extra++;
    }
    j->seconds += now() - t0;
    return NULL;
}

/* One GPU's share of the pipeline. */
typedef struct {
    int gpu;
    double ratio;
    void *d_rows, *d_peq, *d_q;
    void *stream[2], *d_out[2], *d_work[2];
    void *ev_start[4], *ev_stop[4]; /* around the scoring launches of block n: index n & 3 (block n runs on stream[n & 1]) */
    double gpu_ms;                  /* GPU time this device spent scoring the current bucket */
    double busy_ms;                 /* ... and over the whole run: the reference's cal time for this device */
    int have_prev;                  /* ev_stop of the previous block of this bucket is valid */
    void *d_all;                    /* resident mode: the scores of ALL queries against this device's slice of the bucket */
    void **lev_start, **lev_stop;   /* resident mode: events around launch i of the bucket */
    void *ev_copied[2];             /* resident mode: behind the copy-out of the piece in ring slot parity 0 / 1 */
    int delay;                      /* test knob BGSA_DEBUG_DEVICE_DELAY: every block is scored 1 + delay times */
    int64_t first, count;           /* slice of the current bucket, in reads */
} device_t;

/* The reference's dynamic re-balancing (adjust_device_ratio3, BGSA_KNC/global.c:120-168): called after a
 * bucket with the time every device needed for its slice.  Device 0 is the unit; device i's new ratio is
 * its old one scaled by t0 / ti, then smoothed by a weighted mean over the rounds so far in which round r
 * weighs r (the first round, measured with the initial guess, does not enter the mean, global.c:145). */
#define MAX_ROUNDS 4096
static double ratio_history[MAX_ROUNDS][MAX_DEV];
static int ratio_rounds = 0;
static void adjust_device_ratios(device_t *dev, int n, const double *times)
{
    double next[MAX_DEV];
    next[0] = 1.0;
    for (int i = 1; i < n; i++) next[i] = dev[i].ratio * times[0] / times[i];
    const int rnd = ratio_rounds + 1; /* time_index of the reference */
    if (rnd > 1) {
        double total = (double)rnd, acc[MAX_DEV];
        for (int i = 0; i < n; i++) acc[i] = next[i] * rnd;
        for (int r = 1; r < rnd - 1; r++) {
            for (int i = 1; i < n; i++) acc[i] += ratio_history[r][i] * (r + 1);
            total += r + 1;
        }
        for (int i = 1; i < n; i++) next[i] = acc[i] / total;
    }
    if (ratio_rounds < MAX_ROUNDS) {
        for (int i = 0; i < n; i++) ratio_history[ratio_rounds][i] = next[i];
        ratio_rounds++;
    }
    for (int i = 0; i < n; i++) dev[i].ratio = next[i];
}

/* Cut `groups` subject groups into one contiguous run per device, proportional to the ratios.
 * The last device always gets a group: the padding reads at the end of a bucket are recorded in
 * `.info` as belonging to it (convert.c drops `extra_count` scores from the last device's tile
 * only, BGSA_KNC/convert.c:246-254).  What rounding leaves over goes round-robin from device 0. */
static void plan_slices(device_t *dev, int n, int64_t groups)
{
    double total = 0;
    for (int d = 0; d < n; d++) total += dev[d].ratio;
    int64_t given = 0;
    for (int d = 0; d < n; d++) {
        dev[d].count = (int64_t)((double)groups * dev[d].ratio / total);
        given += dev[d].count;
    }
    if (groups > 0 && dev[n - 1].count == 0) {
        dev[n - 1].count = 1;
        if (given < groups) {
            given++;
        } else { /* nothing left over: the widest slice gives one up */
            int widest = 0;
            for (int d = 1; d < n - 1; d++) if (dev[d].count > dev[widest].count) widest = d;
            dev[widest].count--;
        }
    }
    for (int d = 0; given < groups; d = (d + 1) % n) { dev[d].count++; given++; }
    int64_t first = 0;
    for (int d = 0; d < n; d++) {
        dev[d].count *= HIP_V_NUM;
        dev[d].first = first;
        first += dev[d].count;
    }
}

int main(int argc, char **argv)
{
    const char *file_query = NULL, *file_database = NULL, *file_result = "result.txt";
    const char *gpu_list = "0", *file_ratio = NULL;
    int algo = BGSA_ALGO_MYERS, n_dev = 1, c;
    int sc_match = 2, sc_mismatch = -3, sc_gap = -5, sc_given = 0; /* the generator's -M -I -G (README.md:58-66) */
    int semi = 0;                                                   /* the generator's -s */
    int dynamic = 0;                                                /* -D (BGSA_KNC/main.c:71-115) */
    threshold = HIP_BANDED_WORD_SIZE / 2 - 1; /* banded/BGSA_CPU/main.c:43 */
    while ((c = getopt(argc, argv, "t:q:d:f:n:N:M:I:G:k:a:g:R:Dsh")) != -1) {
        switch (c) {
        case 'q': file_query = optarg; break;
        case 'd': file_database = optarg; break;
        case 'f': file_result = optarg; break;
        case 'N': cpu_threads = atoi(optarg); break;
        case 'k': threshold = atoi(optarg); break;
        case 'g': gpu_list = optarg; break;
        case 'n': n_dev = atoi(optarg); break;
        case 'R': file_ratio = optarg; break;
        case 'a':
            if (!strcmp(optarg, "myers")) algo = BGSA_ALGO_MYERS;
            else if (!strcmp(optarg, "banded")) algo = BGSA_ALGO_BANDED;
            else if (!strcmp(optarg, "bitpal")) algo = BGSA_ALGO_BITPAL;
            else usage();
            break;
        case 'M': sc_match = atoi(optarg); sc_given = 1; break;
        case 'I': sc_mismatch = atoi(optarg); sc_given = 1; break;
        case 'G': sc_gap = atoi(optarg); sc_given = 1; break;
        case 's': semi = 1; break;
        case 'D': dynamic = 1; break;
        case 't': break; /* KNC-only knob: accepted, ignored */
        default: usage();
        }
    }
    if (!file_query) { printf("Query file can't be empty.\n"); exit(1); }
    if (!file_database) { printf("Database file can't be empty. \n"); exit(1); }
    if (n_dev < 1 || n_dev > MAX_DEV) { printf("Error - the number of GPUs must be 1..%d\n", MAX_DEV); exit(1); }

    /* ---- which GPUs, and their work ratios ------------------------------------------------------ */
    device_t dev[MAX_DEV];
    memset(dev, 0, sizeof dev);
    {
        int listed = 0;
        const char *s = gpu_list;
        while (*s && listed < MAX_DEV) {
            dev[listed++].gpu = atoi(s);
            while (*s && *s != ',') s++;
            if (*s == ',') s++;
        }
        if (listed > 1) n_dev = listed;                                   /* an explicit list wins */
        else for (int d = 1; d < n_dev; d++) dev[d].gpu = dev[0].gpu + d; /* -g first, -n count */
        const int present = bgsa_hip_device_count();
        for (int d = 0; d < n_dev; d++) {
            if (dev[d].gpu < 0 || dev[d].gpu >= present) {
                printf("Error - GPU %d does not exist (%d visible)\n", dev[d].gpu, present);
                exit(1);
            }
            dev[d].ratio = 1.0;
        }
        if (file_ratio) {
            FILE *fr = open_or_die(file_ratio, "r");
            for (int d = 0; d < n_dev; d++)
                if (fscanf(fr, "%lf", &dev[d].ratio) != 1 || dev[d].ratio <= 0) {
                    printf("Error - %s needs one positive ratio per GPU\n", file_ratio);
                    exit(1);
                }
            fclose(fr);
        }
        if (getenv("BGSA_DEBUG_DEVICE_DELAY")) { /* tests of -D: "0,2,0" = device 1 scores every block three times */
            const char *e = getenv("BGSA_DEBUG_DEVICE_DELAY");
            for (int d = 0; d < n_dev && *e; d++) {
                dev[d].delay = atoi(e);
                while (*e && *e != ',') e++;
                if (*e == ',') e++;
            }
        }
    }

    double total_start = now(), mem_time = 0, cal_time = 0, pipeline_time = 0;
    if (sc_given) algo = BGSA_ALGO_BITPAL;
    CK(bgsa_hip_select_algorithm(algo));
    if (sc_given) {
        CK(bgsa_hip_select_scores(sc_match, sc_mismatch, sc_gap));
        algo = bgsa_hip_current_algorithm(); /* -M 0 -I 1 -G 1 is Myers reporting +distance (generator -m 1) */
    }
    CK(bgsa_hip_select_alignment(semi ? BGSA_ALIGN_SEMIGLOBAL : BGSA_ALIGN_GLOBAL));
    init_mapping_table();
    const size_t esz = algo == BGSA_ALGO_BANDED ? sizeof(hip_banded_write_t) : sizeof(hip_write_t);

    /* ---- queries (get_ref_from_file, file.c:117-140) ------------------------------------- */
    int64_t qsize = file_size(file_query);
    char *qbuf = (char *)malloc_mem((uint64_t)qsize + 2);
    FILE *fq = open_or_die(file_query, "rb");
    if ((int64_t)fread(qbuf, 1, (size_t)qsize, fq) != qsize) { printf("Error - can't read %s\n", file_query); exit(1); }
    fclose(fq);
    if (qsize == 0) { printf("Query file can't be empty.\n"); exit(1); }
    if (qbuf[qsize - 1] != '\n') qbuf[qsize++] = '\n';
    int ref_len = 0;
    while (qbuf[ref_len] != '\n') ref_len++;
    const int64_t ref_count = qsize / (ref_len + 1);
    for (int64_t i = 0; i < qsize; i++)
        if (qbuf[i] != '\n') qbuf[i] = ((unsigned char)qbuf[i] < 128) ? (char)mapping_table[(unsigned char)qbuf[i]] : 0;

    /* ---- database bucket plan (get_read_from_file, file.c:44-115) ----------------------------- */
    const int64_t dsize = file_size(file_database);
    FILE *fd = open_or_die(file_database, "rb");
    char first[8192];
    size_t got = fread(first, 1, sizeof first, fd);
    int read_len = 0;
    while ((size_t)read_len < got && first[read_len] != '\n') read_len++;
    if (read_len == 0 || (size_t)read_len == got) { printf("Error - can't find the read length in %s\n", file_database); exit(1); }
    rewind(fd);
    const int64_t row = read_len + 1;
    const int64_t total_reads = (dsize + 1) / row; /* tolerates a missing final newline */
    int64_t bucket_bytes = READ_BUCKET_SIZE;
    if (getenv("BGSA_READ_BUCKET_SIZE")) bucket_bytes = atoll(getenv("BGSA_READ_BUCKET_SIZE")); /* tests */
    int64_t per_bucket = (bucket_bytes / row) / HIP_V_NUM * HIP_V_NUM;
    if (per_bucket < HIP_V_NUM) per_bucket = HIP_V_NUM;
    const int bucket_num = (int)((total_reads + per_bucket - 1) / per_bucket);
    if (total_reads == 0) { printf("Database file can't be empty. \n"); exit(1); }

    const int word_num = bgsa_hip_word_num(algo, ref_len, read_len, threshold);
    const int64_t max_reads = total_reads < per_bucket ? (total_reads + HIP_V_NUM - 1) / HIP_V_NUM * HIP_V_NUM : per_bucket;
    const size_t rows_bytes = (size_t)(max_reads * row);
    int launch_blocks = LAUNCH_BLOCKS_DEFAULT;
    if (getenv("BGSA_LAUNCH_BLOCKS")) launch_blocks = atoi(getenv("BGSA_LAUNCH_BLOCKS"));
    if (launch_blocks < 1) launch_blocks = 1;
    if (launch_blocks > 16) launch_blocks = 16;
    while (launch_blocks > 1 && (int64_t)(launch_blocks - 1) * REF_BUCKET_COUNT >= ref_count) launch_blocks--;   /* few queries: no more than needed */
    const int64_t launch_queries = (int64_t)REF_BUCKET_COUNT * launch_blocks;
    const size_t work_bytes = bgsa_hip_workspace_bytes(algo, ref_len, read_len, (int)launch_queries);

    /* Resident results (the default when they fit): the GPU scores ALL queries against the bucket into HBM at its own pace — the
     * launches are queued back to back — and the copy-out and the writer drain behind on a second stream.  The kernels then never
     * wait for a ring slot: with a 20 GB result on tmpfs the chunked pipeline left the GPU idle two thirds of the time, clocking
     * down between launches (cal 1.21 s on 10k x 1M against 1.10 s with a sink that keeps up).  15 GB of int16 scores per
     * 760k-subject bucket at 10k queries: what 288 GB of HBM is for.  BGSA_RESULT_RESIDENT=0, or a bucket whose scores exceed
     * BGSA_RESULT_RESIDENT_GB (default 64) per device, keeps the two-launches-in-flight ring pipeline. */
    const int64_t n_launch = (ref_count + launch_queries - 1) / launch_queries;
    double resident_gb = 64.0;
    if (getenv("BGSA_RESULT_RESIDENT_GB")) resident_gb = atof(getenv("BGSA_RESULT_RESIDENT_GB"));
    int resident = !(getenv("BGSA_RESULT_RESIDENT") && getenv("BGSA_RESULT_RESIDENT")[0] == '0') &&
                   (double)ref_count * (double)max_reads * (double)esz <= resident_gb * 1e9 && n_launch <= 65536;
    /* the widest slice any device can be handed: sizes every per-device allocation once */
    plan_slices(dev, n_dev, max_reads / HIP_V_NUM);
    /* ... and only when they fit the memory each device really has free: the limit above is a constant, the card may be
     * smaller, shared, or already hold another job's buffers.  Everything this pipeline allocates per device — rows, Peq,
     * queries, two launch tiles, two workspaces — plus the bucket's scores plus BGSA_RESULT_RESIDENT_HEADROOM_GB (2) must
     * fit into what hipMemGetInfo reports; otherwise the ring pipeline of round 2 runs the same job.
     * (BGSA_DEVICE_FREE_GB overrides the reported figure: tests.) */
    if (resident) {
        double headroom_gb = 2.0;
        if (getenv("BGSA_RESULT_RESIDENT_HEADROOM_GB")) headroom_gb = atof(getenv("BGSA_RESULT_RESIDENT_HEADROOM_GB"));
        for (int d = 0; d < n_dev && resident; d++) {
            size_t free_b = 0, total_b = 0;
            CK(bgsa_hip_set_device(dev[d].gpu));
            CK(bgsa_hip_mem_info(&free_b, &total_b));
            double free_bytes = (double)free_b;
            if (getenv("BGSA_DEVICE_FREE_GB")) free_bytes = atof(getenv("BGSA_DEVICE_FREE_GB")) * 1e9;
            const double cap = (double)(dynamic ? max_reads : dev[d].count + 2 * HIP_V_NUM);
            /* several entries of -g may name the same card (tests, -g 0,0,0): they share its memory */
            int sharers = 0;
            for (int o = 0; o < n_dev; o++) sharers += dev[o].gpu == dev[d].gpu;
            const double base = cap * (double)row + (double)bgsa_hip_group_words(algo, word_num, threshold) * sizeof(hip_read_t) * (cap / HIP_V_NUM) +
                                (double)qsize + 2.0 * ((double)launch_queries * cap * (double)esz + (double)work_bytes);
            const double need = (base + (double)ref_count * cap * (double)esz) * sharers + headroom_gb * 1e9;
            if (need > free_bytes) {
                fprintf(stderr, "[aligner] the bucket's scores (%.2f GB per device) do not fit device %d's free memory (%.2f GB free, "
                                "%.2f GB needed with %.1f GB headroom): ring pipeline instead of resident results\n",
                        (double)ref_count * cap * (double)esz / 1e9, dev[d].gpu, free_bytes / 1e9, need / 1e9, headroom_gb);
                resident = 0;
            }
        }
    }

    /* what goes through a ring slot: a whole launch in the ring pipeline; one block of REF_BUCKET_COUNT queries in resident mode,
     * where the copy-out is independent of the launches (less page-locked memory to set up: 4 x 152 MB instead of 3 x 609 MB) */
    const int64_t slot_queries = resident ? REF_BUCKET_COUNT : launch_queries;
    const size_t block_bytes = (size_t)slot_queries * (size_t)max_reads * esz;

    void *h_rows;
    CK(bgsa_hip_malloc_host(&h_rows, rows_bytes));
    for (int d = 0; d < n_dev; d++) {
        device_t *v = &dev[d];
        /* rounding moves at most two groups between buckets; with -D a slice can grow to the whole bucket */
        const int64_t cap = dynamic ? max_reads : v->count + 2 * HIP_V_NUM;
        CK(bgsa_hip_set_device(v->gpu));
        CK(bgsa_hip_malloc(&v->d_rows, (size_t)(cap * row)));
        CK(bgsa_hip_malloc(&v->d_peq, bgsa_hip_group_words(algo, word_num, threshold) * sizeof(hip_read_t) * (size_t)(cap / HIP_V_NUM)));
        CK(bgsa_hip_malloc(&v->d_q, (size_t)qsize + 8));
        for (int s = 0; s < 2; s++) {
            CK(bgsa_hip_stream_create(&v->stream[s]));
            CK(bgsa_hip_malloc(&v->d_out[s], (size_t)launch_queries * (size_t)cap * esz));
            CK(bgsa_hip_malloc(&v->d_work[s], work_bytes ? work_bytes : 8));
        }
        for (int e = 0; e < 4; e++) {
            CK(bgsa_hip_event_create(&v->ev_start[e]));
            CK(bgsa_hip_event_create(&v->ev_stop[e]));
        }
        if (resident) {
            CK(bgsa_hip_malloc(&v->d_all, (size_t)ref_count * (size_t)cap * esz));
            v->lev_start = (void **)calloc((size_t)n_launch, sizeof(void *));
            v->lev_stop = (void **)calloc((size_t)n_launch, sizeof(void *));
            for (int64_t i = 0; i < n_launch; i++) {
                CK(bgsa_hip_event_create(&v->lev_start[i]));
                CK(bgsa_hip_event_create(&v->lev_stop[i]));
            }
            for (int e = 0; e < 2; e++) CK(bgsa_hip_event_create(&v->ev_copied[e]));
        }
        /* on the device's own stream, and complete before anything is launched (the streams do not
         * synchronise with the NULL stream) */
        CK(bgsa_hip_memcpy_h2d(v->d_q, qbuf, (size_t)qsize, v->stream[0]));
        CK(bgsa_hip_stream_synchronize(v->stream[0]));
    }

    ring_t ring;
    memset(&ring, 0, sizeof ring);
    pthread_mutex_init(&ring.lock, NULL);
    pthread_cond_init(&ring.cond, NULL);
    ring.n_slots = resident ? 4 : 3;   /* one-block pieces: 4 x 152 MB; whole launches: 3 x 609 MB (10k x 1M) */
    for (int i = 0; i < ring.n_slots; i++) CK(bgsa_hip_malloc_host(&ring.host[i], block_bytes));
    ring.fd = open(file_result, O_CREAT | O_TRUNC | O_RDWR, 0644);
    if (ring.fd < 0) { printf("Error - can't open or create file: %s\n", file_result); exit(1); }
    /* the file's final size: every (bucket, query) row of scores, the last bucket padded to whole groups */
    int64_t result_bytes = 0;
    for (int b = 0; b < bucket_num; b++) {
        int64_t n = total_reads - (int64_t)b * per_bucket;
        if (n > per_bucket) n = per_bucket;
        n = (n + HIP_V_NUM - 1) / HIP_V_NUM * HIP_V_NUM;
        result_bytes += n * ref_count * (int64_t)esz;
    }
    const char *wmode = getenv("BGSA_WRITER_MODE");
    const int want_map = wmode && !strncmp(wmode, "mmap", 4);   /* "mmap", or "mmap+populate" */
    g_populate = wmode && strstr(wmode, "populate") != NULL;
    if (want_map && result_bytes > 0 && ftruncate(ring.fd, (off_t)result_bytes) == 0) {
        void *m = mmap(NULL, (size_t)result_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, ring.fd, 0);
        if (m != MAP_FAILED) ring.map = (char *)m;
        else if (ftruncate(ring.fd, 0) != 0) { printf("Error - can't size the result file\n"); exit(1); }
    }
    pthread_t falloc_thread;
    if (ring.map && wmode && strstr(wmode, "falloc") != NULL) {
        ring.falloc = 1;
        ring.file_bytes = result_bytes;
        pthread_create(&falloc_thread, NULL, falloc_main, &ring);
    }
    ring.n_writers = ring.map ? 8 : 1;
    if (getenv("BGSA_WRITER_THREADS")) ring.n_writers = atoi(getenv("BGSA_WRITER_THREADS"));
    if (ring.n_writers < 1) ring.n_writers = 1;
    if (ring.n_writers > 16) ring.n_writers = 16;
    char *info_name = (char *)malloc(strlen(file_result) + 6);
    sprintf(info_name, "%s.info", file_result);
    FILE *finfo = open_or_die(info_name, "wb+");
    fwrite(&bucket_num, sizeof(int), 1, finfo);
    fwrite(&n_dev, sizeof(int), 1, finfo);
    fwrite(&ref_count, sizeof(int64_t), 1, finfo);
    pthread_t writer;
    pthread_create(&writer, NULL, writer_main, &ring);

    int64_t subjects_done = 0;
    read_job_t job = {fd, (char *)h_rows, 0, row, 0, read_len, 0, 0.0};
    pthread_t reader;
    int reading = 0;
    for (int b = 0; b < bucket_num; b++) {
        /* ---- the bucket the input thread read while the previous one was being scored -------- */
        if (reading) {
            pthread_join(reader, NULL);
            reading = 0;
        } else {
            job.want = total_reads < per_bucket ? total_reads : per_bucket;
            read_bucket(&job);
        }
        double t0;
        char *rows = job.rows;
        const int64_t count = job.count;
        const int extra = job.extra;
        plan_slices(dev, n_dev, count / HIP_V_NUM);
        for (int d = 0; d < n_dev; d++) fwrite(&dev[d].count, sizeof(int64_t), 1, finfo);
        fwrite(&extra, sizeof(int), 1, finfo);
        fflush(finfo);

        /* ---- upload + preprocess on the GPUs ("mem" time of the reference report) ----------- */
        t0 = now();
        for (int d = 0; d < n_dev; d++) {
            device_t *v = &dev[d];
            if (!v->count) continue;
            CK(bgsa_hip_set_device(v->gpu));
            CK(bgsa_hip_memcpy_h2d(v->d_rows, rows + v->first * row, (size_t)(v->count * row), v->stream[0]));
            CK(bgsa_hip_handle_reads_dev(algo, (const char *)v->d_rows, v->count * row, read_len, v->count, word_num,
                                         threshold, (hip_read_t *)v->d_peq, v->stream[0]));
        }
        for (int d = 0; d < n_dev; d++) {
            CK(bgsa_hip_set_device(dev[d].gpu));
            CK(bgsa_hip_stream_synchronize(dev[d].stream[0]));
        }
        mem_time += now() - t0;
        if (b == 0 && dynamic) { /* first launches load code objects: keep that out of the device times -D balances on */
            for (int d = 0; d < n_dev; d++) {
                device_t *v = &dev[d];
                if (!v->count) continue;
                CK(bgsa_hip_set_device(v->gpu));
                for (int s = 0; s < 2; s++) {
                    CK(bgsa_hip_cal_align_score_dev(algo, (const char *)v->d_q, (const hip_read_t *)v->d_peq, v->d_out[s], ref_len,
                                                    read_len, v->count, 0, 1, word_num, threshold, v->d_work[s], work_bytes, v->stream[s]));
                    CK(bgsa_hip_stream_synchronize(v->stream[s]));
                }
            }
        }
        if (b + 1 < bucket_num) { /* the rows are in HBM now: the host buffer is free for the next bucket */
            job.want = total_reads - (int64_t)(b + 1) * per_bucket;
            if (job.want > per_bucket) job.want = per_bucket;
            pthread_create(&reader, NULL, read_bucket, &job);
            reading = 1;
        }

        if (resident) {
            /* ---- every launch of the bucket queued at once on stream[0]; pieces copied out in order on stream[1] ---------- */
            t0 = now();
            for (int64_t i = 0; i < n_launch; i++) {
                const int64_t ref_start = i * launch_queries;
                int64_t ref_end = ref_start + launch_queries;
                if (ref_end > ref_count) ref_end = ref_count;
                for (int d = 0; d < n_dev; d++) {
                    device_t *v = &dev[d];
                    if (!v->count) continue;
                    CK(bgsa_hip_set_device(v->gpu));
                    CK(bgsa_hip_event_record(v->lev_start[i], v->stream[0]));
                    for (int rep = 0; rep <= v->delay; rep++)
                        CK(bgsa_hip_cal_align_score_dev(algo, (const char *)v->d_q, (const hip_read_t *)v->d_peq,
                                                        (char *)v->d_all + (size_t)ref_start * (size_t)v->count * esz, ref_len, read_len,
                                                        v->count, (int)ref_start, (int)ref_end, word_num, threshold, v->d_work[0],
                                                        work_bytes, v->stream[0]));
                    CK(bgsa_hip_event_record(v->lev_stop[i], v->stream[0]));
                }
            }
            int pending_slot = -1;
            size_t pending_bytes = 0;
            const int64_t n_piece = (ref_count + slot_queries - 1) / slot_queries;
            for (int64_t i = 0; i <= n_piece; i++) {
                int slot = -1;
                size_t bytes = 0;
                if (i < n_piece) {
                    const int64_t ref_start = i * slot_queries;
                    int64_t ref_end = ref_start + slot_queries;
                    if (ref_end > ref_count) ref_end = ref_count;
                    const int64_t nq = ref_end - ref_start;
                    const int64_t launch = ref_start / launch_queries;   /* the launch that scores this piece */
                    slot = ring_acquire(&ring);
                    char *dst = (char *)ring.host[slot];
                    for (int d = 0; d < n_dev; d++) {
                        device_t *v = &dev[d];
                        if (!v->count) continue;
                        CK(bgsa_hip_set_device(v->gpu));
                        CK(bgsa_hip_stream_wait_event(v->stream[1], v->lev_stop[launch]));
                        const char *src = (const char *)v->d_all + (size_t)ref_start * (size_t)v->count * esz;
                        /* device tiles one after another inside the block (cal_mic.c:535-536) */
                        CK(bgsa_hip_memcpy_d2h(dst + (size_t)nq * (size_t)v->first * esz, src, (size_t)nq * (size_t)v->count * esz, v->stream[1]));
                        CK(bgsa_hip_event_record(v->ev_copied[i & 1], v->stream[1]));
                    }
                    bytes = (size_t)nq * (size_t)count * esz;
                }
                if (pending_slot >= 0) { /* the piece issued one step ago: its copies are done on every device -> to the writer */
                    for (int d = 0; d < n_dev; d++) {
                        if (!dev[d].count) continue;
                        CK(bgsa_hip_set_device(dev[d].gpu));
                        CK(bgsa_hip_event_synchronize(dev[d].ev_copied[(i - 1) & 1]));
                    }
                    ring_publish(&ring, pending_slot, pending_bytes);
                }
                pending_slot = slot;
                pending_bytes = bytes;
            }
            /* the devices' scoring time: the union of the launches' intervals (they run back to back on one stream) */
            for (int d = 0; d < n_dev; d++) {
                if (!dev[d].count) continue;
                CK(bgsa_hip_set_device(dev[d].gpu));
                for (int64_t i = 0; i < n_launch; i++) {
                    float ms = 0, since_prev = 0;
                    CK(bgsa_hip_event_elapsed_ms(dev[d].lev_start[i], dev[d].lev_stop[i], &ms));
                    if (i > 0) {
                        CK(bgsa_hip_event_elapsed_ms(dev[d].lev_stop[i - 1], dev[d].lev_stop[i], &since_prev));
                        if (since_prev < ms) ms = since_prev > 0 ? since_prev : 0;
                    }
                    dev[d].gpu_ms += ms;
                }
            }
            pipeline_time += now() - t0;
        }
        /* ---- query blocks of REF_BUCKET_COUNT (cal_cpu.c:363-401), launch_blocks of them per launch, two launches in flight per GPU */
        t0 = now();
        double stalled = 0;
        int slot_of[2] = {-1, -1};
        size_t bytes_of[2] = {0, 0};
        int64_t issued = 0;
        for (int64_t ref_start = resident ? ref_count : 0;; ref_start += launch_queries, issued++) {
            const int s = (int)(issued & 1);
            if (slot_of[s] >= 0) { /* the block issued two steps ago on this stream pair: finish and hand over */
                /* block n = issued - 2 is complete on every device.  Its share of the device's scoring time is the part of
                 * [start, stop] that lies behind the previous block's stop: the two streams of a device overlap (block n's
                 * launches are queued while block n - 1 still runs), and time must not be counted twice. */
                const int e = (int)((issued - 2) & 3), ep = (int)((issued - 3) & 3);
                for (int d = 0; d < n_dev; d++) {
                    CK(bgsa_hip_set_device(dev[d].gpu));
                    CK(bgsa_hip_stream_synchronize(dev[d].stream[s]));
                    if (dev[d].count) {
                        float ms = 0, since_prev = 0;
                        CK(bgsa_hip_event_elapsed_ms(dev[d].ev_start[e], dev[d].ev_stop[e], &ms));
                        if (dev[d].have_prev) {
                            CK(bgsa_hip_event_elapsed_ms(dev[d].ev_stop[ep], dev[d].ev_stop[e], &since_prev));
                            if (since_prev < ms) ms = since_prev > 0 ? since_prev : 0;
                        }
                        dev[d].have_prev = 1;
                        dev[d].gpu_ms += ms;
                    }
                }
                ring_publish(&ring, slot_of[s], bytes_of[s]);
                slot_of[s] = -1;
            }
            if (ref_start >= ref_count) {
                if (slot_of[s ^ 1] < 0) break;
                continue; /* one more turn drains the other stream pair */
            }
            int64_t ref_end = ref_start + launch_queries;
            if (ref_end > ref_count) ref_end = ref_count;
            const int64_t nq = ref_end - ref_start;
            double w0 = now();
            const int slot = ring_acquire(&ring);
            if (slot_of[s ^ 1] < 0) stalled += now() - w0; /* nothing in flight: the GPUs sat idle */
            char *dst = (char *)ring.host[slot];
            for (int d = 0; d < n_dev; d++) {
                device_t *v = &dev[d];
                if (!v->count) continue;
                CK(bgsa_hip_set_device(v->gpu));
                CK(bgsa_hip_event_record(v->ev_start[issued & 3], v->stream[s]));
                for (int rep = 0; rep <= v->delay; rep++)
                    CK(bgsa_hip_cal_align_score_dev(algo, (const char *)v->d_q, (const hip_read_t *)v->d_peq, v->d_out[s],
                                                    ref_len, read_len, v->count, (int)ref_start, (int)ref_end, word_num,
                                                    threshold, v->d_work[s], work_bytes, v->stream[s]));
                CK(bgsa_hip_event_record(v->ev_stop[issued & 3], v->stream[s]));
                /* device tiles one after another inside every block of REF_BUCKET_COUNT queries (cal_mic.c:535-536); with one
                 * device the blocks of a launch are one contiguous piece */
                if (n_dev == 1) {
                    CK(bgsa_hip_memcpy_d2h(dst, v->d_out[s], (size_t)nq * (size_t)v->count * esz, v->stream[s]));
                } else {
                    for (int64_t q0 = 0; q0 < nq; q0 += REF_BUCKET_COUNT) {
                        const int64_t nb = nq - q0 < REF_BUCKET_COUNT ? nq - q0 : REF_BUCKET_COUNT;
                        CK(bgsa_hip_memcpy_d2h(dst + (size_t)q0 * (size_t)count * esz + (size_t)nb * (size_t)v->first * esz,
                                               (char *)v->d_out[s] + (size_t)q0 * (size_t)v->count * esz,
                                               (size_t)nb * (size_t)v->count * esz, v->stream[s]));
                    }
                }
            }
            slot_of[s] = slot;
            bytes_of[s] = (size_t)nq * (size_t)count * esz;
        }
        if (!resident) pipeline_time += now() - t0 - stalled; /* wall time with scoring work in flight on the GPUs (copies and the wait for the writer included) */
        subjects_done += count;
        for (int d = 0; d < n_dev; d++) { /* a damaged query stream is an error, not a wrong score (bgsa_hip.h) */
            CK(bgsa_hip_set_device(dev[d].gpu));
            if (bgsa_hip_stream_faults(1) != 0) die("stream fault");
        }
        if (dynamic && n_dev > 1) {
            double times[MAX_DEV];
            int usable = 1;
            for (int d = 0; d < n_dev; d++) {
                /* what the reference measures is the time of device i for a share proportional to its ratio;
                 * slices are whole groups, so take the time per read times the ratio instead of the raw time */
                times[d] = dev[d].count ? dev[d].gpu_ms / (double)dev[d].count * dev[d].ratio : 0;
                if (times[d] <= 0) usable = 0;
            }
            if (usable) {
                adjust_device_ratios(dev, n_dev, times);
                printf("bucket %d: device times per read", b);
                for (int d = 0; d < n_dev; d++) printf(" %.3fus", 1e3 * dev[d].gpu_ms / (double)dev[d].count);
                printf(" -> ratios");
                for (int d = 0; d < n_dev; d++) printf(" %.3f", dev[d].ratio);
                printf("\n");
            }
        }
        for (int d = 0; d < n_dev; d++) {
            dev[d].busy_ms += dev[d].gpu_ms;
            dev[d].gpu_ms = 0;
            dev[d].have_prev = 0;
        }
    }
    for (int d = 0; d < n_dev; d++)   /* the devices work side by side: the slowest one's scoring time */
        if (dev[d].busy_ms * 1e-3 > cal_time) cal_time = dev[d].busy_ms * 1e-3;
    pthread_mutex_lock(&ring.lock);
    ring.done = 1;
    pthread_cond_broadcast(&ring.cond);
    pthread_mutex_unlock(&ring.lock);
    pthread_join(writer, NULL);
    if (ring.falloc) pthread_join(falloc_thread, NULL);
    if (ring.map) {
        if (ring.offset != result_bytes) { printf("Error - the result file is %ld bytes, planned %ld\n", (long)ring.offset, (long)result_bytes); exit(1); }
        munmap(ring.map, (size_t)result_bytes);
    }
    close(ring.fd);
    fclose(finfo);
    fclose(fd);
    const double total = now() - total_start;

    /* ---- the reference's report (cal_cpu.c:459-475) ---------------------------------------- */
    printf("score is %d, %d, %d\n", match_score, mismatch_score, gap_score);
    printf("read_total_time  is %.2fs\n", job.seconds);
    printf("write_total_time is %.2fs\n", ring.seconds);
    printf("mem_total_time is   %.2fs\n\n", mem_time);
    printf("query_len    is %d\n", ref_len);
    printf("query_count  is %ld\n", (long)ref_count);
    printf("subject_len   is %d\n", read_len);
    printf("subject_count is %ld\n\n", (long)subjects_done);
    printf("gpu_count     is %d\n", n_dev);
    /* cal = time spent scoring, as the reference measures it around its compute call only (cal_cpu.c:111-118; per device
     * in the KNC backend, cal_mic.c:150-152): here the GPU time of the scoring launches of every block (HIP events on the
     * block's stream; with several GPUs the slowest device's), copies and file I/O outside, as they are outside there. */
    printf("pipeline_busy_time  is %.2fs\n", pipeline_time);
    printf("result_pipeline     is %s\n", resident ? "resident (the bucket's scores stay in HBM, copy-out behind the kernels)"
                                                   : "ring (two launches in flight)");
    printf("cal_total_times     is %.2fs\n", cal_time);
    printf("total time          is %.2fs\n", total);
    const double cells = 1.0 * ref_len * ref_count * read_len * subjects_done;
    printf("cal GCUPS is %.2f\n", cells / cal_time / 1e9);
    printf("Total GCUPS is %.2f\n\n\n", cells / total / 1e9);

    for (int i = 0; i < ring.n_slots; i++) bgsa_hip_free_host(ring.host[i]);
    bgsa_hip_free_host(h_rows);
    for (int d = 0; d < n_dev; d++) {
        device_t *v = &dev[d];
        bgsa_hip_set_device(v->gpu);
        bgsa_hip_free(v->d_rows); bgsa_hip_free(v->d_peq); bgsa_hip_free(v->d_q);
        for (int s = 0; s < 2; s++) {
            bgsa_hip_stream_destroy(v->stream[s]);
            bgsa_hip_free(v->d_out[s]); bgsa_hip_free(v->d_work[s]);
        }
        for (int e = 0; e < 4; e++) { bgsa_hip_event_destroy(v->ev_start[e]); bgsa_hip_event_destroy(v->ev_stop[e]); }
        if (resident) {
            bgsa_hip_free(v->d_all);
            for (int64_t i = 0; i < n_launch; i++) { bgsa_hip_event_destroy(v->lev_start[i]); bgsa_hip_event_destroy(v->lev_stop[i]); }
            for (int e = 0; e < 2; e++) bgsa_hip_event_destroy(v->ev_copied[e]);
            free(v->lev_start); free(v->lev_stop);
        }
    }
    free_mem(qbuf);
    free(info_name);
    return 0;
}
