/* The "use the kernel alignment method directly" demo of the reference's README (README.md:94-165), on the HIP
 * backend: one query against a handful of subjects through the three seams a BGSA backend exports —
 * hip_handle_reads (preprocess), align_hip (score), malloc_mem / free_mem (buffers).
 *
 *     make -C examples/demo && examples/demo/demo_hip            # needs an MI355X: there is no CPU fallback
 *
 * Differences from the SSE demo are the backend's constants only: HIP_V_NUM = 64 subjects per group (the subject
 * set is padded to a multiple of it with all-'N' reads, as get_read_from_file pads the last bucket, file.c:98-112),
 * 32 data bits per word (bgsa_hip_word_num), and no scratch buffer (the DP state lives in registers). */
#include <stdio.h>
#include <string.h>

#include "bgsa_hip.h"

static void align(const char *query, const char **subjects, int subject_count)
{
    const int query_len = (int)strlen(query);
    const int subject_len = (int)strlen(subjects[0]);
    const int total = (subject_count + HIP_V_NUM - 1) / HIP_V_NUM * HIP_V_NUM;

    /* subjects: one row of subject_len characters + '\n' each, the layout of the subject file */
    seq_t seq;
    memset(&seq, 0, sizeof seq);
    seq.len = subject_len;
    seq.count = subject_count;
    seq.size = (int64_t)total * (subject_len + 1);
    seq.content = malloc_mem((uint64_t)seq.size);
    for (int i = 0; i < total; i++) {
        char *row = seq.content + (size_t)i * (subject_len + 1);
        if (i < subject_count)
            memcpy(row, subjects[i], (size_t)subject_len);
        else
            memset(row, 'N', (size_t)subject_len);
        row[subject_len] = '\n';
    }

    const int word_num = bgsa_hip_word_num(bgsa_hip_current_algorithm(), query_len, subject_len, threshold);
    const size_t peq_words = bgsa_hip_group_words(bgsa_hip_current_algorithm(), word_num, threshold) * (size_t)(total / HIP_V_NUM);
    hip_read_t *peq = malloc_mem(peq_words * sizeof(hip_read_t));
    memset(peq, 0, peq_words * sizeof(hip_read_t));          /* the caller zeroes it, cal_cpu.c:273 */
    init_mapping_table();
    hip_handle_reads(&seq, peq, word_num, 0, total);

    char *mapped = malloc_mem((uint64_t)query_len + 1);        /* the query through the alphabet map, file.c:134-139 */
    for (int i = 0; i < query_len; i++) mapped[i] = (char)mapping_table[(unsigned char)query[i] & 127];
    mapped[query_len] = '\n';

    hip_write_t *results = malloc_mem(sizeof(hip_write_t) * (size_t)total);
    align_hip(mapped, peq, query_len, subject_len, word_num, total / HIP_V_NUM, 0, results, NULL);
    for (int i = 0; i < subject_count; i++) printf("%d\n", results[i]);

    free_mem(seq.content);
    free_mem(peq);
    free_mem(mapped);
    free_mem(results);
}

int main(int argc, char **argv)
{
    const char *query = "AAAA";
    const char *subjects[4] = {"AAAA", "AACA", "CAAC", "AGGG"};
    if (argc > 1 && strcmp(argv[1], "bitpal") == 0)
        bgsa_hip_select_algorithm(BGSA_ALGO_BITPAL);           /* 2 / -3 / -5, the reference's committed BitPAl kernel */
    else
        bgsa_hip_select_algorithm(BGSA_ALGO_MYERS);            /* -edit distance */
    align(query, subjects, 4);
    return 0;
}
