/*
 * aligner.c — the BGSA `aligner` command line on top of libbgsa_hip.so.
 *
 * Same contract as the reference's pipeline driver (original/BGSA_CPU/main.c:36-106 +
 * cal_cpu.c:121-476), written from scratch in C against the C ABI of include/bgsa_hip.h:
 *
 *   ./aligner -q <query file> -d <database file> -f <result file> [-N host threads]
 *             [-k threshold] [-a myers|banded|bitpal] [-g gpu]
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
 *     (cal_cpu.c:459-475): "cal" = time inside the scoring calls, "Total" = wall.
 *
 * The subject bucket lives in HBM: rows are uploaded once, preprocessed on the GPU, and every
 * query block is one asynchronous launch; a writer thread drains finished score blocks to disk
 * from a ring of pinned buffers while the GPU works on the next block (the reference's
 * input/output pthreads, thread.c:35-171, reduced to the one that matters here).
 */
#define _GNU_SOURCE
#include <getopt.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <sys/time.h>

#include "bgsa_hip.h"

#define READ_BUCKET_SIZE 114857600LL /* original/BGSA_CPU/config.h:6 */
#define REF_BUCKET_COUNT 100         /* original/BGSA_CPU/config.h:13 */
#define RING 3

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
    int state[RING]; /* 0 free, 1 filled */
    int head, tail, done;
    FILE *fp;
    double seconds;
    pthread_mutex_t lock;
    pthread_cond_t cond;
} ring_t;

static void *writer_main(void *arg)
{
    ring_t *r = (ring_t *)arg;
    for (;;) {
        pthread_mutex_lock(&r->lock);
        while (!r->state[r->tail] && !r->done) pthread_cond_wait(&r->cond, &r->lock);
        if (!r->state[r->tail] && r->done) {
            pthread_mutex_unlock(&r->lock);
            return NULL;
        }
        int slot = r->tail;
        pthread_mutex_unlock(&r->lock);
        double t0 = now();
        if (fwrite(r->host[slot], 1, r->bytes[slot], r->fp) != r->bytes[slot]) {
            printf("Error - short write to the result file\n");
            exit(1);
        }
        r->seconds += now() - t0;
        pthread_mutex_lock(&r->lock);
        r->state[slot] = 0;
        r->tail = (slot + 1) % RING;
        pthread_cond_broadcast(&r->cond);
        pthread_mutex_unlock(&r->lock);
    }
}

static int ring_acquire(ring_t *r)
{
    pthread_mutex_lock(&r->lock);
    while (r->state[r->head]) pthread_cond_wait(&r->cond, &r->lock);
    int slot = r->head;
    pthread_mutex_unlock(&r->lock);
    return slot;
}

static void ring_publish(ring_t *r, int slot, size_t bytes)
{
    pthread_mutex_lock(&r->lock);
    r->bytes[slot] = bytes;
    r->state[slot] = 1;
    r->head = (slot + 1) % RING;
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
    printf("  -g <arg>\n\t GPU index. Default 0. \n\n");
    exit(1);
}

int main(int argc, char **argv)
{
    const char *file_query = NULL, *file_database = NULL, *file_result = "result.txt";
    int algo = BGSA_ALGO_MYERS, gpu = 0, c;
    threshold = HIP_BANDED_WORD_SIZE / 2 - 1; /* banded/BGSA_CPU/main.c:43 */
    while ((c = getopt(argc, argv, "t:q:d:f:n:N:k:a:g:R:Dh")) != -1) {
        switch (c) {
        case 'q': file_query = optarg; break;
        case 'd': file_database = optarg; break;
        case 'f': file_result = optarg; break;
        case 'N': cpu_threads = atoi(optarg); break;
        case 'k': threshold = atoi(optarg); break;
        case 'g': gpu = atoi(optarg); break;
        case 'a':
            if (!strcmp(optarg, "myers")) algo = BGSA_ALGO_MYERS;
            else if (!strcmp(optarg, "banded")) algo = BGSA_ALGO_BANDED;
            else if (!strcmp(optarg, "bitpal")) algo = BGSA_ALGO_BITPAL;
            else usage();
            break;
        case 't': case 'n': case 'R': case 'D': break; /* KNC-only knobs: accepted, ignored */
        default: usage();
        }
    }
    if (!file_query) { printf("Query file can't be empty.\n"); exit(1); }
    if (!file_database) { printf("Database file can't be empty. \n"); exit(1); }

    double total_start = now(), read_time = 0, mem_time = 0, cal_time = 0;
    CK(bgsa_hip_select_algorithm(algo));
    CK(bgsa_hip_set_device(gpu));
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
    const size_t peq_bytes = bgsa_hip_group_words(algo, word_num, threshold) * sizeof(hip_read_t) * (size_t)(max_reads / HIP_V_NUM);
    const size_t block_bytes = (size_t)REF_BUCKET_COUNT * (size_t)max_reads * esz;
    const size_t work_bytes = bgsa_hip_workspace_bytes(algo, ref_len, read_len, REF_BUCKET_COUNT);

    void *h_rows, *d_rows, *d_peq, *d_q, *d_out, *d_work;
    CK(bgsa_hip_malloc_host(&h_rows, rows_bytes));
    CK(bgsa_hip_malloc(&d_rows, rows_bytes));
    CK(bgsa_hip_malloc(&d_peq, peq_bytes));
    CK(bgsa_hip_malloc(&d_q, (size_t)qsize + 8));
    CK(bgsa_hip_malloc(&d_out, block_bytes));
    CK(bgsa_hip_malloc(&d_work, work_bytes));
    CK(bgsa_hip_memcpy_h2d(d_q, qbuf, (size_t)qsize, NULL));

    ring_t ring;
    memset(&ring, 0, sizeof ring);
    pthread_mutex_init(&ring.lock, NULL);
    pthread_cond_init(&ring.cond, NULL);
    for (int i = 0; i < RING; i++) CK(bgsa_hip_malloc_host(&ring.host[i], block_bytes));
    ring.fp = open_or_die(file_result, "wb+");
    char *info_name = (char *)malloc(strlen(file_result) + 6);
    sprintf(info_name, "%s.info", file_result);
    FILE *finfo = open_or_die(info_name, "wb+");
    const int device_num = 1;
    fwrite(&bucket_num, sizeof(int), 1, finfo);
    fwrite(&device_num, sizeof(int), 1, finfo);
    fwrite(&ref_count, sizeof(int64_t), 1, finfo);
    pthread_t writer;
    pthread_create(&writer, NULL, writer_main, &ring);

    int64_t subjects_done = 0;
    for (int b = 0; b < bucket_num; b++) {
        /* ---- read one bucket of rows, pad the last one with 'N' reads ---------------------- */
        double t0 = now();
        int64_t want = total_reads - (int64_t)b * per_bucket;
        if (want > per_bucket) want = per_bucket;
        size_t n = fread(h_rows, 1, (size_t)(want * row), fd);
        char *rows = (char *)h_rows;
        if ((int64_t)n < want * row) rows[n++] = '\n'; /* file without a final newline */
        int64_t count = want;
        int extra = 0;
        while (count % HIP_V_NUM) {
            memset(rows + count * row, 'N', (size_t)read_len);
            rows[count * row + read_len] = '\n';
            count++;
            extra++;
        }
        read_time += now() - t0;
        fwrite(&count, sizeof(int64_t), 1, finfo);
        fwrite(&extra, sizeof(int), 1, finfo);
        fflush(finfo);

        /* ---- upload + preprocess on the GPU ("mem" time of the reference report) ------------ */
        t0 = now();
        CK(bgsa_hip_memcpy_h2d(d_rows, rows, (size_t)(count * row), NULL));
        CK(bgsa_hip_handle_reads_dev(algo, (const char *)d_rows, count * row, read_len, count, word_num,
                                     threshold, (hip_read_t *)d_peq, NULL));
        CK(bgsa_hip_stream_synchronize(NULL));
        mem_time += now() - t0;

        /* ---- query blocks of REF_BUCKET_COUNT (cal_cpu.c:363-401) --------------------------------- */
        for (int64_t ref_start = 0; ref_start < ref_count; ref_start += REF_BUCKET_COUNT) {
            int64_t ref_end = ref_start + REF_BUCKET_COUNT;
            if (ref_end > ref_count) ref_end = ref_count;
            const size_t bytes = (size_t)(ref_end - ref_start) * (size_t)count * esz;
            t0 = now();
            CK(bgsa_hip_cal_align_score_dev(algo, (const char *)d_q, (const hip_read_t *)d_peq, d_out, ref_len,
                                            read_len, count, (int)ref_start, (int)ref_end, word_num, threshold,
                                            d_work, work_bytes, NULL));
            CK(bgsa_hip_stream_synchronize(NULL));
            cal_time += now() - t0;
            int slot = ring_acquire(&ring);
            CK(bgsa_hip_memcpy_d2h(ring.host[slot], d_out, bytes, NULL));
            CK(bgsa_hip_stream_synchronize(NULL));
            ring_publish(&ring, slot, bytes);
        }
        subjects_done += count;
    }
    pthread_mutex_lock(&ring.lock);
    ring.done = 1;
    pthread_cond_broadcast(&ring.cond);
    pthread_mutex_unlock(&ring.lock);
    pthread_join(writer, NULL);
    fclose(ring.fp);
    fclose(finfo);
    fclose(fd);
    const double total = now() - total_start;

    /* ---- the reference's report (cal_cpu.c:459-475) ---------------------------------------- */
    printf("score is %d, %d, %d\n", match_score, mismatch_score, gap_score);
    printf("read_total_time  is %.2fs\n", read_time);
    printf("write_total_time is %.2fs\n", ring.seconds);
    printf("mem_total_time is   %.2fs\n\n", mem_time);
    printf("query_len    is %d\n", ref_len);
    printf("query_count  is %ld\n", (long)ref_count);
    printf("subject_len   is %d\n", read_len);
    printf("subject_count is %ld\n\n", (long)subjects_done);
    printf("cal_total_times     is %.2fs\n", cal_time);
    printf("total time          is %.2fs\n", total);
    const double cells = 1.0 * ref_len * ref_count * read_len * subjects_done;
    printf("cal GCUPS is %.2f\n", cells / cal_time / 1e9);
    printf("Total GCUPS is %.2f\n\n\n", cells / total / 1e9);

    for (int i = 0; i < RING; i++) bgsa_hip_free_host(ring.host[i]);
    bgsa_hip_free_host(h_rows);
    bgsa_hip_free(d_rows); bgsa_hip_free(d_peq); bgsa_hip_free(d_q); bgsa_hip_free(d_out); bgsa_hip_free(d_work);
    free_mem(qbuf);
    free(info_name);
    return 0;
}
