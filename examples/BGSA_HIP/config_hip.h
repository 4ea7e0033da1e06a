/*
 * config_hip.h — what a BGSA_HIP backend directory's config.h says, in the form
 * that lets the reference's OWN host sources be compiled against libbgsa_hip.so without touching them:
 *
 *     gcc -D_CONFIG_H_ -include examples/BGSA_HIP/config_hip.h  file.c thread.c cal_cpu.c main.c  -lbgsa_hip
 *
 * -D_CONFIG_H_ switches off original/BGSA_CPU/config.h (its include guard, config.h:1-2); this header
 * then provides the same names with the HIP backend's values, and maps the two ISA-specific symbols the
 * host files call — cpu_handle_reads (thread.c:110, cal_cpu.c:274) and align_cpu (cal_cpu.c:81) — onto
 * the library's hip_handle_reads / align_hip.  global.c and align_core.c are simply not compiled: the
 * library exports mapping_table, init_mapping_table, malloc_mem, free_mem and the score ints they held.
 *
 * oracle/Makefile (target ref) builds exactly this into oracle/_ref/original_hip/ and
 * tests/test_cli_gpu.py runs it on the golden fixtures: the reference's main.c / file.c / thread.c /
 * cal_cpu.c, unmodified, produce the reference's result files through the GPU library.
 */
#ifndef BGSA_HIP_CONFIG_SHIM_H
#define BGSA_HIP_CONFIG_SHIM_H

#include <stdint.h>

#define READ_BUCKET_SIZE 114857600   /* original/BGSA_CPU/config.h:6 */
#ifndef REF_BUCKET_COUNT             /* (-DREF_BUCKET_COUNT=<n>: the builds that check the seam with other query blocks) */
#define REF_BUCKET_COUNT 100         /* :13 */
#endif
#define CHAR_NUM 5                   /* :18 */
#define common_write_t int16_t       /* :20 */

#define CPU_V_NUM 64                 /* HIP_V_NUM: one subject per wavefront lane (was 1, :22) */
#define CPU_WORD_SIZE 32             /* HIP_WORD_SIZE (was 64, :23) */
#define CPU_SIZE 64
#define cpu_read_t uint32_t          /* hip_read_t (was uint64_t, :26) */
#define cpu_write_t common_write_t
#define cpu_data_t uint32_t          /* hip_data_t: the scratch argument is accepted and ignored */

#define cpu_handle_reads hip_handle_reads
#define align_cpu align_hip

#endif
