/*
 * convert.c — BGSA's `convert` tool for the HIP backend, written from
 * scratch against the formats of the reference (original/BGSA_CPU/convert.c:31-277):
 *
 *   ./convert -f <fasta>  [-o out]   FASTA  -> one sequence per line
 *   ./convert -q <fastq>  [-o out]   FASTQ  -> one sequence per line
 *   ./convert -r <result> [-o out]   binary result + result.info -> one score per line,
 *                                    query-major, subjects in file order, padding dropped
 *
 * The result element width (int16 for Myers/BitPAl, int8 for banded — a compile-time choice in
 * the reference, config.h common_write_t) is derived from the file size.
 */
#include <getopt.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>

#define REF_BUCKET_COUNT 100 /* original/BGSA_CPU/config.h:13 */

static FILE *open_or_die(const char *name, const char *mode)
{
    FILE *fp = fopen(name, mode);
    if (!fp) {
        printf("Error - can't open or create file: %s\n", name);
        exit(1);
    }
    return fp;
}

/* Records start with `mark`; the header line is dropped; FASTA joins the following lines,
 * FASTQ keeps only the first one (sequence) and skips '+' and quality lines. */
static void convert_records(const char *in, const char *out, char mark, int fastq)
{
    FILE *fi = open_or_die(in, "rb"), *fo = open_or_die(out, "w+");
    int c, at_line_start = 1, in_header = 0, line_in_record = 0, wrote_any = 0, skip_line = 0;
    while ((c = fgetc(fi)) != EOF) {
        if (at_line_start && c == mark && (!fastq || line_in_record == 0 || line_in_record >= 4)) {
            if (wrote_any) fputc('\n', fo);
            in_header = 1;
            line_in_record = 0;
            at_line_start = 0;
            wrote_any = 1;
            continue;
        }
        if (c == '\n') {
            if (in_header) in_header = 0;
            line_in_record++;
            at_line_start = 1;
            skip_line = fastq && line_in_record >= 2; /* '+' line and qualities */
            continue;
        }
        at_line_start = 0;
        if (in_header || skip_line || c == '\r') continue;
        fputc(c, fo);
    }
    fputc('\n', fo);
    fclose(fi);
    fclose(fo);
}

static void convert_result(const char *result, const char *out)
{
    char *info_name = (char *)malloc(strlen(result) + 6);
    sprintf(info_name, "%s.info", result);
    FILE *fr = fopen(result, "rb"), *fi = fopen(info_name, "rb");
    if (!fr) { printf("Can't read result file\n"); exit(1); }
    if (!fi) { printf("Can't read result info file\n"); exit(1); }
    FILE *fo = fopen(out, "w+");
    if (!fo) { printf("Can't create output file\n"); exit(1); }

    int bucket_num = 0, device_num = 0;
    int64_t ref_count = 0;
    if (fread(&bucket_num, sizeof(int), 1, fi) != 1 || fread(&device_num, sizeof(int), 1, fi) != 1 ||
        fread(&ref_count, sizeof(int64_t), 1, fi) != 1 || bucket_num <= 0 || device_num <= 0) {
        printf("Can't read result info file\n");
        exit(1);
    }
    int64_t *counts = (int64_t *)malloc(sizeof(int64_t) * (size_t)bucket_num * device_num);
    int *extra = (int *)malloc(sizeof(int) * (size_t)bucket_num);
    int64_t reads_total = 0, widest = 0;
    for (int b = 0; b < bucket_num; b++) {
        if (fread(&counts[(size_t)b * device_num], sizeof(int64_t), device_num, fi) != (size_t)device_num ||
            fread(&extra[b], sizeof(int), 1, fi) != 1) {
            printf("Can't read result info file\n");
            exit(1);
        }
        for (int d = 0; d < device_num; d++) {
            reads_total += counts[(size_t)b * device_num + d];
            if (counts[(size_t)b * device_num + d] > widest) widest = counts[(size_t)b * device_num + d];
            printf("read_count[%d][%d] is %ld\n", b, d, (long)counts[(size_t)b * device_num + d]);
        }
    }
    struct stat st;
    stat(result, &st);
    const int64_t pairs = ref_count * reads_total;
    const int esz = pairs > 0 ? (int)(st.st_size / pairs) : 2;
    if (esz != 1 && esz != 2) { printf("Result file size does not match its info file\n"); exit(1); }

    /* File order: for each read bucket, for each block of REF_BUCKET_COUNT queries, for each
     * device: [queries in block][reads of device] (thread.c:150-160, cal_mic.c:535-536). */
    const int64_t n_blocks = (ref_count + REF_BUCKET_COUNT - 1) / REF_BUCKET_COUNT;
    int64_t *bucket_off = (int64_t *)malloc(sizeof(int64_t) * (size_t)(bucket_num + 1));
    bucket_off[0] = 0;
    for (int b = 0; b < bucket_num; b++) {
        int64_t reads = 0;
        for (int d = 0; d < device_num; d++) reads += counts[(size_t)b * device_num + d];
        bucket_off[b + 1] = bucket_off[b] + ref_count * reads * esz;
    }
    void *buf = malloc((size_t)widest * esz);
    for (int64_t q = 0; q < ref_count; q++) {
        const int64_t block = q / REF_BUCKET_COUNT, in_block = q % REF_BUCKET_COUNT;
        int64_t block_rows = ref_count - block * REF_BUCKET_COUNT;
        if (block_rows > REF_BUCKET_COUNT) block_rows = REF_BUCKET_COUNT;
        (void)n_blocks;
        for (int b = 0; b < bucket_num; b++) {
            int64_t reads = 0;
            for (int d = 0; d < device_num; d++) reads += counts[(size_t)b * device_num + d];
            int64_t off = bucket_off[b] + block * REF_BUCKET_COUNT * reads * esz;
            for (int d = 0; d < device_num; d++) {
                const int64_t n = counts[(size_t)b * device_num + d];
                const int64_t drop = (d == device_num - 1) ? extra[b] : 0;
                fseek(fr, off + in_block * n * esz, SEEK_SET);
                if ((int64_t)fread(buf, (size_t)esz, (size_t)n, fr) != n) { printf("Can't read result file\n"); exit(1); }
                for (int64_t i = 0; i < n - drop; i++)
                    fprintf(fo, "%d\n", esz == 2 ? (int)((int16_t *)buf)[i] : (int)((int8_t *)buf)[i]);
                off += block_rows * n * esz;
            }
        }
    }
    free(buf); free(bucket_off); free(counts); free(extra); free(info_name);
    fclose(fr); fclose(fi); fclose(fo);
}

static void usage(void)
{
    printf("\nUsage: ./convert [options]\n\nCommandline options:\n\n");
    printf("  -f <arg>\n\t Convert the FASTA file to needed format. \n\n");
    printf("  -q <arg>\n\t Convert the FASTQ file to needed format. \n\n");
    printf("  -r <arg>\n\t Convert the result file to readable format. \n\n");
    printf("  -o <arg>\n\t Output file. \n\n");
    exit(1);
}

int main(int argc, char **argv)
{
    const char *in = NULL, *out = "convert_result.txt";
    int kind = -1, c;
    if (argc == 1) usage();
    while ((c = getopt(argc, argv, "f:q:r:o:h")) != -1) {
        switch (c) {
        case 'f': kind = 0; in = optarg; break;
        case 'q': kind = 1; in = optarg; break;
        case 'r': kind = 2; in = optarg; break;
        case 'o': out = optarg; break;
        default: usage();
        }
    }
    if (!in) { printf("Input file can't be empty.\n"); exit(1); }
    if (kind == 0) convert_records(in, out, '>', 0);
    else if (kind == 1) convert_records(in, out, '@', 1);
    else convert_result(in, out);
    return 0;
}
