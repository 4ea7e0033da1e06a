/*
 * bgsa_oracle.h — CPU restatement of BGSA's all-pairs bit-parallel alignment hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is linked into, imported by, or executed from
 * the product (libbgsa_hip.so / bgsa_amd).  Allowed users: tests/, __graft_entry__.smoke(), and
 * bench.py's cpu_baseline leg — always as the checker or the timed CPU baseline, never as the
 * thing shipped.
 *
 * Parity status: PINNED.  The reference (sdu-hpcl/BGSA) ships no golden vectors,
 * so the pin is the reference itself compiled here from its own sources (oracle/Makefile `ref`
 * -> oracle/_ref/<variant>/aligner) and run on seeded inputs; its `convert -r` text is committed
 * under tests/golden/ by scripts/make_golden.py, and tests/test_oracle.py checks every function
 * below against those fixtures (plus independent textbook DP).
 *
 * All sequence inputs are "row buffers" in the reference's own file format: `count` rows of
 * `len` ASCII bytes followed by '\n' (row stride len+1), exactly what get_read_from_file /
 * get_ref_from_file hold in memory (reference original/BGSA_CPU/file.c:44-140).
 * All score outputs are row-major [query][subject] like cpu_cal_align_score
 * (reference original/BGSA_CPU/cal_cpu.c:43-85).
 */
#ifndef BGSA_ORACLE_H
#define BGSA_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Alphabet map of init_mapping_table (reference original/BGSA_CPU/global.c:9-15):
 * A,C,G,T,N -> 0..4, every other byte < 128 -> 0 (zero-initialised global table).
 * Bytes >= 128 index out of bounds in the reference; here they map to 0. */
uint8_t bgsa_oracle_map_char(uint8_t ch);

/* ---- Myers unit-cost global, scalar 64-bit words with 63 data bits --------------------------
 * Restates cpu_handle_reads (original/BGSA_CPU/global.c:25-70) + align_cpu
 * (original/BGSA_CPU/align_core.c:19-148).  out[q*ns+s] = -(edit distance). */
void bgsa_oracle_myers64(const char *queries, int64_t nq, int qlen,
                         const char *subjects, int64_t ns, int slen,
                         int16_t *out, int threads);

/* ---- Myers unit-cost global, 32-bit words with 31 data bits (the SSE/AVX lane form) ---------
 * Restates sse_handle_reads + align_sse (original/BGSA_SSE/global.c, align_core.c:19-152),
 * one lane at a time.  Must equal bgsa_oracle_myers64 everywhere. */
void bgsa_oracle_myers31(const char *queries, int64_t nq, int qlen,
                         const char *subjects, int64_t ns, int slen,
                         int16_t *out, int threads);

/* ---- Banded Myers, scalar 64-bit single word ------------------------------------------------
 * Restates banded cpu_handle_reads (banded/BGSA_CPU/global.c:25-84) + align_cpu
 * (banded/BGSA_CPU/align_core.c:69-252).  out is int8 (MAX_ERROR = 127 on early exit).
 * `subjects` must have at least `threshold` readable bytes after the last row in the reference
 * (its preprocess over-reads); here bytes past the buffer are treated as '\n' (-> plane 0),
 * and they are never consumed when qlen == slen >= 64. */
void bgsa_oracle_banded64(const char *queries, int64_t nq, int qlen,
                          const char *subjects, int64_t ns, int slen,
                          int threshold, int8_t *out, int threads);

/* ---- BitPAl packed, match 2 / mismatch -3 / gap -5, 32-bit words with 31 data bits ----------
 * Restates avx_handle_reads + align_avx (original/BGSA_AVX2/global.c:27-71,
 * align_core.c:19-484), one lane at a time.  out = Needleman-Wunsch linear-gap score. */
void bgsa_oracle_bitpal(const char *queries, int64_t nq, int qlen,
                        const char *subjects, int64_t ns, int slen,
                        int16_t *out, int threads);

/* ---- Independent textbook DP cross-checks (not restatements) -------------------------------- */
/* -(unit-cost global edit distance) with the BGSA alphabet map applied to both sides. */
void bgsa_oracle_dp_edit(const char *queries, int64_t nq, int qlen,
                         const char *subjects, int64_t ns, int slen,
                         int16_t *out, int threads);
/* -min over query prefixes y of D[slen][y], D[x][0] = x, D[0][y] = 0 (the generator's Myers -s). */
void bgsa_oracle_dp_edit_semiglobal(const char *queries, int64_t nq, int qlen,
                         const char *subjects, int64_t ns, int slen,
                         int16_t *out, int threads);
/* Needleman-Wunsch, linear gap. */
void bgsa_oracle_dp_nw(const char *queries, int64_t nq, int qlen,
                       const char *subjects, int64_t ns, int slen,
                       int match, int mismatch, int gap,
                       int16_t *out, int threads);
/* S[0][j] = 0, S[i][0] = i*gap, result = max over the last row (the generator's -s mode for BitPAl). */
void bgsa_oracle_dp_semiglobal(const char *queries, int64_t nq, int qlen,
                       const char *subjects, int64_t ns, int slen,
                       int match, int mismatch, int gap,
                       int16_t *out, int threads);
/* Closed form of the banded kernel's output for qlen == slen. */
void bgsa_oracle_dp_banded(const char *queries, int64_t nq, int qlen,
                           const char *subjects, int64_t ns, int slen,
                           int threshold, int8_t *out, int threads);

/* ---- Timed CPU baseline ("port"): Myers global on AVX2, 8 subjects x 31 data bits per vector.
 * Same algorithm as the reference's SIMD Myers (align_sse widened to 256-bit, which is what the
 * reference's generator emits for -a avx2; that output is not committed upstream).  ns is
 * processed in groups of 8; ns must be a multiple of 8.  Returns seconds spent in the scoring
 * loop (preprocess excluded), like the reference's cal_total_times (cal_cpu.c:111-118). */
double bgsa_oracle_myers_avx2(const char *queries, int64_t nq, int qlen,
                              const char *subjects, int64_t ns, int slen,
                              int16_t *out, int threads);

#ifdef __cplusplus
}
#endif
#endif
