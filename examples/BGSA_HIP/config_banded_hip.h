/*
 * config_banded_hip.h — config.h of a banded/BGSA_HIP backend directory (banded column),
 * in the form that lets the reference's OWN banded host sources be compiled against libbgsa_hip.so
 * without touching them:
 *
 *     gcc -D_CONFIG_H_ -include examples/BGSA_HIP/config_banded_hip.h \
 *         file.c thread.c cal_cpu.c main.c examples/BGSA_HIP/select_banded.c -lbgsa_hip
 *
 * (run in banded/BGSA_CPU).  Unlike the original/ build, the element type and word size stay the
 * reference's: banded/BGSA_CPU/cal_cpu.c:253-254 sizes the preprocessed buffer as
 * word_num = (read_len - h_threshold + CPU_WORD_SIZE - 1) / CPU_WORD_SIZE + 1 words of cpu_read_t, and the
 * library's host seams accept exactly that word_num: the buffer then holds the offset match string in
 * 2 * word_num 32-bit words per class and lane, and hip_cal_align_score / align_hip re-pitch it to the
 * device layout on upload (include/bgsa_hip.h: hip_handle_reads).  Only the lane count changes
 * (CPU_V_NUM 1 -> 64), so that file.c:84-112 rounds bucket counts to whole wavefront groups.
 * main.c keeps its own `int threshold` (-k); the library reads that very variable.
 */
#ifndef BGSA_HIP_BANDED_CONFIG_SHIM_H
#define BGSA_HIP_BANDED_CONFIG_SHIM_H

#include <stdint.h>

#define READ_BUCKET_SIZE 114857600   /* banded/BGSA_CPU/config.h:6 */
#define REF_BUCKET_COUNT 100         /* :13 */
#define CHAR_NUM 5                   /* :18 */
#define MAX_ERROR 127                /* :19 */
#define batch_size 16                /* :20 */
#define common_write_t int8_t        /* :21 */

#define CPU_V_NUM 64                 /* HIP_V_NUM: one subject per wavefront lane (was 1, :23) */
#define CPU_WORD_SIZE 64             /* :24, unchanged: it only sizes word_num and the default threshold */
#define CPU_SIZE 64
#define cpu_read_t uint64_t          /* :26, unchanged */
#define cpu_write_t common_write_t
#define cpu_data_t uint64_t
#define cpu_cmp_result_t int64_t
#define cpu_result_t int64_t

#define cpu_handle_reads hip_handle_reads
#define align_cpu align_hip

#endif
