/*
 * bgsa_oracle.c — CPU restatement of BGSA's bit-parallel all-pairs alignment hot path.
 *
 * TEST INFRASTRUCTURE ONLY — see bgsa_oracle.h.  Parity: pinned against the compiled reference
 * (oracle/_ref, fixtures under tests/golden/).
 *
 * Every function cites the reference file:line it follows (paths relative to the reference
 * root).  The code is a restatement — written from the algorithm, one scalar lane at a time —
 * not a copy: the reference's scratch-buffer / chunk / OpenMP-thread-id plumbing is replaced by
 * plain per-pair locals.
 */
#include "bgsa_oracle.h"

#include <immintrin.h>
#include <omp.h>
#include <stdlib.h>
#include <string.h>

#define NCHAR 5 /* CHAR_NUM, original/BGSA_CPU/config.h:18 */

/* init_mapping_table, original/BGSA_CPU/global.c:9-15 */
uint8_t bgsa_oracle_map_char(uint8_t ch)
{
    switch (ch) {
    case 'A': return 0;
    case 'C': return 1;
    case 'G': return 2;
    case 'T': return 3;
    case 'N': return 4;
    default:  return 0;
    }
}

static int pick_threads(int threads)
{
    return threads > 0 ? threads : omp_get_max_threads();
}

/* get_ref_from_file maps the query bytes in place (original/BGSA_CPU/file.c:134-139). */
static uint8_t *map_rows(const char *rows, int64_t n, int len)
{
    uint8_t *m = (uint8_t *)malloc((size_t)n * (size_t)len + 1);
    for (int64_t r = 0; r < n; r++)
        for (int i = 0; i < len; i++)
            m[r * len + i] = bgsa_oracle_map_char((uint8_t)rows[r * (int64_t)(len + 1) + i]);
    return m;
}

/* ------------------------------------------------------------------------------------------- */
/* Myers global, W-bit words carrying W-1 data bits                                            */
/* ------------------------------------------------------------------------------------------- */

/*
 * Per-subject match masks: bit (p mod D) of word (p div D) in plane c is set iff subject[p]==c,
 * D = data bits per word.  cpu_handle_reads, original/BGSA_CPU/global.c:44-67 with V_NUM lanes
 * collapsed to one; word count per cal_cpu.c:255.
 */
static void peq_build64(const char *s, int slen, int nw, uint64_t *peq /* [5][nw] */)
{
    memset(peq, 0, sizeof(uint64_t) * NCHAR * nw);
    for (int p = 0; p < slen; p++)
        peq[bgsa_oracle_map_char((uint8_t)s[p]) * nw + p / 63] |= 1ULL << (p % 63);
}

static void peq_build32(const char *s, int slen, int nw, uint32_t *peq /* [5][nw] */)
{
    memset(peq, 0, sizeof(uint32_t) * NCHAR * nw);
    for (int p = 0; p < slen; p++)
        peq[bgsa_oracle_map_char((uint8_t)s[p]) * nw + p / 31] |= 1u << (p % 31);
}

/*
 * One (query, subject) pair of align_cpu, original/BGSA_CPU/align_core.c:54-146.
 * The word's top bit is the inter-word carry of the addition (:76-83) and of the HP/HN shifts
 * (:91-96); the score follows cell (row, slen) through the last word's HP/HN bit (:121-124).
 */
static int16_t myers64_pair(const uint8_t *q, int qlen, const uint64_t *peq, int slen, int nw,
                            uint64_t *vp, uint64_t *vn)
{
    const uint64_t LOW = 0x7fffffffffffffffULL;
    const uint64_t top = 1ULL << ((slen - 1) % 63);
    uint64_t score = (uint64_t)slen;

    for (int w = 0; w < nw; w++) { vn[w] = 0; vp[w] = LOW; }

    for (int r = 0; r < qlen; r++) {
        const uint64_t *eq = peq + (size_t)q[r] * nw;
        uint64_t hp_in = 1, hn_in = 0, sum = 0;
        for (int w = 0; w < nw; w++) {
            uint64_t pv = vp[w], mv = vn[w];
            uint64_t pm = eq[w] | mv;
            uint64_t cin = sum >> 63;
            sum = (pv & pm) + pv + cin;
            uint64_t d0 = ((sum & LOW) ^ pv) | pm;
            uint64_t hp = ~(d0 | pv) | mv;
            uint64_t hn = d0 & pv;
            if (w == nw - 1) {
                if (hn & top) score--;
                else if (hp & top) score++;
            }
            hp = (hp << 1) | hp_in; hp_in = hp >> 63;
            hn = (hn << 1) | hn_in; hn_in = hn >> 63;
            vp[w] = (~(d0 | hp) | hn) & LOW;
            vn[w] = d0 & hp & LOW;
        }
    }
    /* score *= -1 on the unsigned word, low 32 bits reinterpreted as int, stored to int16
     * (align_core.c:137-144). */
    score = score * (uint64_t)-1;
    return (int16_t)(int32_t)(uint32_t)score;
}

/* Same recurrence on 32-bit lanes, 31 data bits; branch-free score update of
 * original/BGSA_SSE/align_core.c:121-128. */
static int16_t myers31_pair(const uint8_t *q, int qlen, const uint32_t *peq, int slen, int nw,
                            uint32_t *vp, uint32_t *vn)
{
    const uint32_t LOW = 0x7fffffffu;
    const uint32_t top = 1u << ((slen - 1) % 31);
    uint32_t score = (uint32_t)slen;

    for (int w = 0; w < nw; w++) { vn[w] = 0; vp[w] = LOW; }

    for (int r = 0; r < qlen; r++) {
        const uint32_t *eq = peq + (size_t)q[r] * nw;
        uint32_t hp_in = 1, hn_in = 0, sum = 0;
        for (int w = 0; w < nw; w++) {
            uint32_t pv = vp[w], mv = vn[w];
            uint32_t pm = eq[w] | mv;
            uint32_t cin = sum >> 31;
            sum = (pv & pm) + pv + cin;
            uint32_t d0 = ((sum & LOW) ^ pv) | pm;
            uint32_t hp = ~(d0 | pv) | mv;
            uint32_t hn = d0 & pv;
            if (w == nw - 1) {
                score += ((hp & top) == top);
                score -= ((hn & top) == top);
            }
            hp = (hp << 1) | hp_in; hp_in = hp >> 31;
            hn = (hn << 1) | hn_in; hn_in = hn >> 31;
            vp[w] = (~(d0 | hp) | hn) & LOW;
            vn[w] = d0 & hp & LOW;
        }
    }
    return (int16_t)(int32_t)(score * (uint32_t)-1);
}

void bgsa_oracle_myers64(const char *queries, int64_t nq, int qlen, const char *subjects,
                         int64_t ns, int slen, int16_t *out, int threads)
{
    if (nq <= 0 || ns <= 0) return;
    const int nw = (slen + 64 - 2) / (64 - 1); /* cal_cpu.c:255 */
    uint8_t *q = map_rows(queries, nq, qlen);
#pragma omp parallel num_threads(pick_threads(threads))
    {
        uint64_t *peq = (uint64_t *)malloc(sizeof(uint64_t) * (NCHAR + 2) * (nw > 0 ? nw : 1));
        uint64_t *vp = peq + NCHAR * nw, *vn = vp + nw;
#pragma omp for schedule(dynamic, 16)
        for (int64_t s = 0; s < ns; s++) {
            peq_build64(subjects + s * (int64_t)(slen + 1), slen, nw, peq);
            for (int64_t i = 0; i < nq; i++)
                out[i * ns + s] = myers64_pair(q + i * qlen, qlen, peq, slen, nw, vp, vn);
        }
        free(peq);
    }
    free(q);
}

void bgsa_oracle_myers31(const char *queries, int64_t nq, int qlen, const char *subjects,
                         int64_t ns, int slen, int16_t *out, int threads)
{
    if (nq <= 0 || ns <= 0) return;
    const int nw = (slen + 32 - 2) / (32 - 1);
    uint8_t *q = map_rows(queries, nq, qlen);
#pragma omp parallel num_threads(pick_threads(threads))
    {
        uint32_t *peq = (uint32_t *)malloc(sizeof(uint32_t) * (NCHAR + 2) * (nw > 0 ? nw : 1));
        uint32_t *vp = peq + NCHAR * nw, *vn = vp + nw;
#pragma omp for schedule(dynamic, 16)
        for (int64_t s = 0; s < ns; s++) {
            peq_build32(subjects + s * (int64_t)(slen + 1), slen, nw, peq);
            for (int64_t i = 0; i < nq; i++)
                out[i * ns + s] = myers31_pair(q + i * qlen, qlen, peq, slen, nw, vp, vn);
        }
        free(peq);
    }
    free(q);
}

/* ------------------------------------------------------------------------------------------- */
/* Banded Myers, one 64-bit word sliding along the diagonal                                    */
/* ------------------------------------------------------------------------------------------- */

/*
 * Banded preprocess, banded/BGSA_CPU/global.c:44-82: word 0 holds the first k subject
 * characters at bits k+1 .. 2k; words 1.. hold, 64 per word, `slen` characters starting at
 * index k — i.e. the loop runs k characters past the row.  `avail` = readable bytes from the
 * row start; bytes beyond are treated as '\n' (plane 0).  Only words < nw are written.
 */
static void banded_peq_build(const char *row, int64_t avail, int slen, int k, int nw,
                             uint64_t *peq /* [5][nw] */)
{
    memset(peq, 0, sizeof(uint64_t) * NCHAR * nw);
    for (int p = 0; p < k; p++) {
        uint8_t ch = p < avail ? (uint8_t)row[p] : (uint8_t)'\n';
        if (nw > 0) peq[bgsa_oracle_map_char(ch) * nw + 0] |= 1ULL << (k + 1 + p);
    }
    for (int i = 0; i < slen; i++) {
        int p = k + i;
        int w = 1 + i / 64;
        if (w >= nw) break;
        uint8_t ch = p < avail ? (uint8_t)row[p] : (uint8_t)'\n';
        peq[bgsa_oracle_map_char(ch) * nw + w] |= 1ULL << (i % 64);
    }
}

typedef struct {
    uint64_t peq[NCHAR]; /* the sliding window of match bits, one per plane */
    uint64_t nxt[NCHAR]; /* the word new window bits are taken from */
    uint64_t vp, vn, d0, err;
} band_t;

/* cpu_cal_D0, banded/BGSA_CPU/align_core.c:19-33 */
static inline void band_step(band_t *b, int c)
{
    uint64_t x = b->peq[c] | b->vn;
    uint64_t d0 = (((x & b->vp) + b->vp) ^ b->vp) | x;
    uint64_t hn = d0 & b->vp;
    uint64_t hp = ~(d0 | b->vp) | b->vn;
    x = d0 >> 1;
    b->vn = x & hp;
    b->vp = ~(hp | x) | hn;
    b->d0 = d0;
}
/* cpu_cal_score, :64-67 */
static inline void band_score(band_t *b) { b->err += 1 - (b->d0 & 1); }
/* cpu_move_peq, :35-40 */
static inline void band_shift(band_t *b)
{
    for (int c = 0; c < NCHAR; c++) b->peq[c] >>= 1;
}
/* cpu_or_peq, :42-62 */
static inline void band_feed(band_t *b, int bit, int band_down)
{
    for (int c = 0; c < NCHAR; c++) b->peq[c] |= ((b->nxt[c] >> bit) & 1ULL) << band_down;
}
static inline void band_load(band_t *b, const uint64_t *peq, int nw, int w)
{
    /* The reference reads word w unguarded (align_core.c:156-160,183-187); a word at or past
     * word_num belongs to the next block and is never consumed — read it as zero. */
    for (int c = 0; c < NCHAR; c++) b->nxt[c] = w < nw ? peq[c * nw + w] : 0;
}

/* One pair of banded align_cpu, banded/BGSA_CPU/align_core.c:80-250. */
static int8_t banded_pair(const uint8_t *q, int qlen, const uint64_t *peq, int slen, int nw, int k)
{
    const int h = k + slen - qlen;  /* h_threshold :70 */
    const int band_down = k + h;    /* band_length - 1 :71-72 */
    band_t b;
    for (int c = 0; c < NCHAR; c++) b.peq[c] = nw > 0 ? peq[c * nw] : 0;
    band_load(&b, peq, nw, 1);
    b.vp = b.vn = b.d0 = 0;
    b.err = (uint64_t)k;
    const uint64_t max_err = (uint64_t)(k + h + 1);
    int i_bd = h, bit = 0, qi = 0;

    for (; qi < k; qi++) { /* :116-123 */
        band_step(&b, q[qi]);
        band_shift(&b); band_feed(&b, bit, band_down);
        bit++; i_bd++;
    }
    const int first = qlen < 64 ? qlen : 64;
    for (; qi < first; qi++) { /* :125-134 */
        band_step(&b, q[qi]); band_score(&b);
        band_shift(&b); band_feed(&b, bit, band_down);
        bit++; i_bd++;
    }
    if (b.err > max_err) return 127; /* :136-140 */

    if (qlen > 64) { /* :142-227 */
        bit = 0;
        const int rest = slen - i_bd;
        const int batch_count = rest / 16;
        const int word_count = rest / 64;
        int batch_index = 0, word_index = 2;
        band_load(&b, peq, nw, word_index);
        for (int i = 0; i < word_count; i++) {
            for (int j = 0; j < 4; j++) {
                for (int t = 0; t < 16; t++) {
                    band_step(&b, q[qi]); band_score(&b);
                    band_shift(&b); band_feed(&b, bit, band_down);
                    bit++; i_bd++; qi++;
                }
                if (b.err > max_err) return 127;
                batch_index++;
            }
            bit = 0;
            word_index++;
            band_load(&b, peq, nw, word_index);
        }
        for (; batch_index < batch_count; batch_index++) {
            for (int t = 0; t < 16; t++) {
                band_step(&b, q[qi]); band_score(&b);
                band_shift(&b); band_feed(&b, bit, band_down);
                bit++; i_bd++; qi++;
            }
            if (b.err > max_err) return 127;
        }
        for (; i_bd < slen; i_bd++) {
            band_step(&b, q[qi]); band_score(&b);
            band_shift(&b); band_feed(&b, bit, band_down);
            bit++; qi++;
        }
        if (b.err > max_err) return 127;
        for (; qi < qlen; qi++) { /* tail: window only shrinks, no check afterwards :221-226 */
            band_step(&b, q[qi]); band_score(&b);
            band_shift(&b);
        }
    }

    /* :230-245 — walk the last row across the band, keep the minimum. */
    uint64_t err = b.err, best = b.err;
    for (int i = 0; i <= h; i++) {
        err += (b.vp >> i) & 1ULL;
        err -= (b.vn >> i) & 1ULL;
        if (err < best) best = err; /* unsigned compare, as the reference (:239) */
    }
    return (int8_t)(int64_t)best;
}

void bgsa_oracle_banded64(const char *queries, int64_t nq, int qlen, const char *subjects,
                          int64_t ns, int slen, int threshold, int8_t *out, int threads)
{
    if (nq <= 0 || ns <= 0) return;
    const int h = threshold + slen - qlen;
    const int nw = (slen - h + 63) / 64 + 1; /* banded/BGSA_CPU/cal_cpu.c:253-254 */
    const int64_t total = ns * (int64_t)(slen + 1);
    uint8_t *q = map_rows(queries, nq, qlen);
#pragma omp parallel num_threads(pick_threads(threads))
    {
        uint64_t *peq = (uint64_t *)malloc(sizeof(uint64_t) * NCHAR * (nw > 0 ? nw : 1));
#pragma omp for schedule(dynamic, 16)
        for (int64_t s = 0; s < ns; s++) {
            int64_t off = s * (int64_t)(slen + 1);
            banded_peq_build(subjects + off, total - off, slen, threshold, nw, peq);
            for (int64_t i = 0; i < nq; i++)
                out[i * ns + s] = banded_pair(q + i * qlen, qlen, peq, slen, nw, threshold);
        }
        free(peq);
    }
    free(q);
}

/* ------------------------------------------------------------------------------------------- */
/* BitPAl packed, match 2 / mismatch -3 / gap -5                                               */
/* ------------------------------------------------------------------------------------------- */

/* Values that cross a word boundary inside one row (original/BGSA_AVX2/align_core.c:176-179):
 * five add overflows, four "init" shift-out bits and the five plane shift-out bits. */
typedef struct {
    uint32_t ov[5];
    uint32_t init_prev[4];
    uint32_t plane_prev[5];
} bitpal_carry_t;

/* The repeated "shift the seed left by one across words, add it to the remaining dH=-5 run,
 * keep what the add toggled" block (:229-238, :240-250, :252-264, :266-280). */
static inline uint32_t bitpal_run(uint32_t seed, uint32_t run, uint32_t not_match, uint32_t *prev,
                                  uint32_t *ov)
{
    uint32_t v = (seed << 1) | *prev;
    *prev = v >> 31;
    v &= 0x7fffffffu;
    uint32_t s = v + run + *ov;
    *ov = s >> 31;
    return (s ^ run) & not_match;
}

/* 5-bit ripple add of two bit-sliced numbers (:299-325 and :393-419). */
static inline void bitpal_add5(const uint32_t a[5], const uint32_t b[5], uint32_t s[5])
{
    uint32_t carry = a[0] & b[0];
    s[0] = a[0] ^ b[0];
    for (int i = 1; i < 5; i++) {
        uint32_t x = a[i] ^ b[i];
        s[i] = x ^ carry;
        carry = (a[i] & b[i]) | (x & carry);
    }
}

/* One (query char, word) step of align_avx, original/BGSA_AVX2/align_core.c:183-428. */
static inline void bitpal_word(uint32_t match, uint32_t h[5], bitpal_carry_t *cy)
{
    const uint32_t LOW = 0x7fffffffu;
    const uint32_t nm = ~match;
    const uint32_t h1 = h[0], h2 = h[1], h4 = h[2], h8 = h[3], h16 = h[4];

    /* decode the planes to one-hot dH classes (:191-214) */
    const uint32_t top = h16 & h8;
    const uint32_t a = top & ~h4, b = top & h4;
    const uint32_t pos2 = a & ~h2 & h1;
    const uint32_t pos1 = a & h2 & ~h1;
    const uint32_t zero = a & h2 & h1;
    const uint32_t neg1 = b & ~h2 & ~h1;
    const uint32_t neg2 = b & ~h2 & h1;
    const uint32_t neg3 = b & h2 & ~h1;
    const uint32_t neg4 = b & h2 & h1;
    const uint32_t neg5 = ~h16 & ~h8 & ~h4 & ~h2 & ~h1 & LOW;

    /* dV one-hots 7..3 (:216-279) */
    const uint32_t seed7 = neg5 & match;
    uint32_t s = seed7 + neg5 + cy->ov[0];
    const uint32_t dv7 = (s ^ neg5 ^ seed7) & LOW;
    cy->ov[0] = s >> 31;
    const uint32_t run = neg5 ^ seed7;
    const uint32_t dv7m = dv7 | match;
    const uint32_t dv6 = bitpal_run(neg4 & dv7m, run, nm, &cy->init_prev[0], &cy->ov[1]);
    const uint32_t dv5 = bitpal_run((neg3 & dv7m) | (neg4 & dv6), run, nm, &cy->init_prev[1], &cy->ov[2]);
    const uint32_t dv4 = bitpal_run((neg2 & dv7m) | (neg3 & dv6) | (neg4 & dv5), run, nm,
                                    &cy->init_prev[2], &cy->ov[3]);
    const uint32_t dv3 = bitpal_run((neg1 & dv7m) | (neg2 & dv6) | (neg3 & dv5) | (neg4 & dv4), run, nm,
                                    &cy->init_prev[3], &cy->ov[4]);

    /* encode dV to planes (:281-297) */
    const uint32_t rest = ~(dv7m | dv6 | dv5 | dv4 | dv3);
    uint32_t v[5];
    v[0] = rest | dv4 | dv6;
    v[1] = dv5 | dv6 | rest;
    v[2] = rest | dv7m;
    v[3] = dv5 | dv6 | dv3 | dv4 | dv7m;
    v[4] = 0;

    /* dH + dV, clamp negatives to zero (:299-331) */
    uint32_t t[5];
    bitpal_add5(h, v, t);
    const uint32_t keep = ~t[4];
    for (int i = 0; i < 5; i++) t[i] &= keep;

    /* shift one column up, carrying bit 30 into the next word (:333-360) */
    for (int i = 0; i < 5; i++) {
        uint32_t out = (t[i] & 0x40000000u) >> 30;
        t[i] = (t[i] << 1) | cy->plane_prev[i];
        cy->plane_prev[i] = out;
    }

    /* new dH seed from match / mismatch (:368-391); dh_pos7 is identically zero there */
    const uint32_t any = (neg5 | neg4 | neg3 | neg2 | neg1 | zero | pos1 | pos2) & nm;
    uint32_t g[5];
    g[0] = (h1 | any) & nm;
    g[1] = (h2 & ~any) & nm;
    g[2] = (h4 & ~any) | match;
    g[3] = (h8 | any) & nm;
    g[4] = (h16 | any) | match;

    /* add the shifted sum and mask by sign (:393-426) */
    uint32_t r[5];
    bitpal_add5(g, t, r);
    for (int i = 0; i < 4; i++) h[i] = r[i] & r[4];
    h[4] = r[4];
}

static int16_t bitpal_pair(const uint8_t *q, int qlen, const uint32_t *peq, int slen, int nw,
                           uint32_t *planes /* [nw][5] */)
{
    memset(planes, 0, sizeof(uint32_t) * 5 * nw); /* :167-171 */
    for (int r = 0; r < qlen; r++) {
        const uint32_t *eq = peq + (size_t)q[r] * nw;
        bitpal_carry_t cy;
        memset(&cy, 0, sizeof cy); /* :176-179 */
        for (int w = 0; w < nw; w++) bitpal_word(eq[w], planes + 5 * w, &cy);
    }
    /* :433-471 */
    int32_t score = -5 * qlen;
    for (int p = 0; p < slen; p++) {
        const uint32_t *h = planes + 5 * (p / 31);
        int bit = p % 31;
        score += 16 * (int32_t)((h[4] >> bit) & 1) - (int32_t)((h[0] >> bit) & 1) -
                 2 * (int32_t)((h[1] >> bit) & 1) - 4 * (int32_t)((h[2] >> bit) & 1) -
                 8 * (int32_t)((h[3] >> bit) & 1) - 5;
    }
    return (int16_t)score;
}

void bgsa_oracle_bitpal(const char *queries, int64_t nq, int qlen, const char *subjects,
                        int64_t ns, int slen, int16_t *out, int threads)
{
    if (nq <= 0 || ns <= 0) return;
    const int nw = (slen + 32 - 2) / (32 - 1); /* original/BGSA_AVX2/cal_avx.c word_num */
    uint8_t *q = map_rows(queries, nq, qlen);
#pragma omp parallel num_threads(pick_threads(threads))
    {
        uint32_t *peq = (uint32_t *)malloc(sizeof(uint32_t) * (NCHAR + 5) * (nw > 0 ? nw : 1));
        uint32_t *planes = peq + NCHAR * nw;
#pragma omp for schedule(dynamic, 16)
        for (int64_t s = 0; s < ns; s++) {
            peq_build32(subjects + s * (int64_t)(slen + 1), slen, nw, peq);
            for (int64_t i = 0; i < nq; i++)
                out[i * ns + s] = bitpal_pair(q + i * qlen, qlen, peq, slen, nw, planes);
        }
        free(peq);
    }
    free(q);
}

/* ------------------------------------------------------------------------------------------- */
/* Independent textbook DP                                                                     */
/* ------------------------------------------------------------------------------------------- */

void bgsa_oracle_dp_edit(const char *queries, int64_t nq, int qlen, const char *subjects,
                         int64_t ns, int slen, int16_t *out, int threads)
{
    if (nq <= 0 || ns <= 0) return;
    uint8_t *q = map_rows(queries, nq, qlen);
    uint8_t *s = map_rows(subjects, ns, slen);
#pragma omp parallel num_threads(pick_threads(threads))
    {
        int *row = (int *)malloc(sizeof(int) * (size_t)(slen + 1));
#pragma omp for schedule(dynamic, 16) collapse(2)
        for (int64_t i = 0; i < nq; i++)
            for (int64_t j = 0; j < ns; j++) {
                const uint8_t *a = q + i * qlen, *b = s + j * slen;
                for (int x = 0; x <= slen; x++) row[x] = x;
                for (int y = 1; y <= qlen; y++) {
                    int diag = row[0];
                    row[0] = y;
                    for (int x = 1; x <= slen; x++) {
                        int up = row[x];
                        int best = diag + (a[y - 1] != b[x - 1]);
                        if (up + 1 < best) best = up + 1;
                        if (row[x - 1] + 1 < best) best = row[x - 1] + 1;
                        row[x] = best;
                        diag = up;
                    }
                }
                out[i * ns + j] = (int16_t)(-row[slen]);
            }
        free(row);
    }
    free(q); free(s);
}

/*
 * Semi-global unit-cost DP as the reference's generator defines it for Myers (`-m 0 -s`,
 * MyersGenerator.java:56-223 genSemiGlobal): the subject is the pattern (VP starts all ones:
 * D[x][0] = x), the query is the text (h_in = 0 on every row: D[0][y] = 0), and the kernel keeps
 * the minimum of D[slen][y] over y = 0..qlen, starting from D[slen][0] = slen (:115-117, :205-208).
 * Result = -min (factor -1 as for the global kernel).  Note the orientation: here the SUBJECT is
 * aligned end to end inside the query — the opposite of BitPAl's semi-global above.
 */
void bgsa_oracle_dp_edit_semiglobal(const char *queries, int64_t nq, int qlen, const char *subjects,
                                    int64_t ns, int slen, int16_t *out, int threads)
{
    if (nq <= 0 || ns <= 0) return;
    uint8_t *q = map_rows(queries, nq, qlen);
    uint8_t *s = map_rows(subjects, ns, slen);
#pragma omp parallel num_threads(pick_threads(threads))
    {
        int *row = (int *)malloc(sizeof(int) * (size_t)(slen + 1));
#pragma omp for schedule(dynamic, 16) collapse(2)
        for (int64_t i = 0; i < nq; i++)
            for (int64_t j = 0; j < ns; j++) {
                const uint8_t *a = q + i * qlen, *b = s + j * slen;
                for (int x = 0; x <= slen; x++) row[x] = x;
                int lowest = row[slen];
                for (int y = 1; y <= qlen; y++) {
                    int diag = row[0];
                    row[0] = 0;
                    for (int x = 1; x <= slen; x++) {
                        int up = row[x];
                        int best = diag + (a[y - 1] != b[x - 1]);
                        if (up + 1 < best) best = up + 1;
                        if (row[x - 1] + 1 < best) best = row[x - 1] + 1;
                        row[x] = best;
                        diag = up;
                    }
                    if (row[slen] < lowest) lowest = row[slen];
                }
                out[i * ns + j] = (int16_t)(-lowest);
            }
        free(row);
    }
    free(q); free(s);
}

void bgsa_oracle_dp_nw(const char *queries, int64_t nq, int qlen, const char *subjects,
                       int64_t ns, int slen, int match, int mismatch, int gap, int16_t *out,
                       int threads)
{
    if (nq <= 0 || ns <= 0) return;
    uint8_t *q = map_rows(queries, nq, qlen);
    uint8_t *s = map_rows(subjects, ns, slen);
#pragma omp parallel num_threads(pick_threads(threads))
    {
        int *row = (int *)malloc(sizeof(int) * (size_t)(slen + 1));
#pragma omp for schedule(dynamic, 16) collapse(2)
        for (int64_t i = 0; i < nq; i++)
            for (int64_t j = 0; j < ns; j++) {
                const uint8_t *a = q + i * qlen, *b = s + j * slen;
                for (int x = 0; x <= slen; x++) row[x] = x * gap;
                for (int y = 1; y <= qlen; y++) {
                    int diag = row[0];
                    row[0] = y * gap;
                    for (int x = 1; x <= slen; x++) {
                        int up = row[x];
                        int best = diag + (a[y - 1] == b[x - 1] ? match : mismatch);
                        if (up + gap > best) best = up + gap;
                        if (row[x - 1] + gap > best) best = row[x - 1] + gap;
                        row[x] = best;
                        diag = up;
                    }
                }
                out[i * ns + j] = (int16_t)row[slen];
            }
        free(row);
    }
    free(q); free(s);
}

/*
 * Semi-global DP as the reference's generator defines it for BitPAl (`-s`, generator/source/.../
 * BitPAlGenerator.java: writeBitInitStr :2201-2218 makes every dH of row 0 zero, genPackedScore
 * :78-116 starts at gap*query_len and keeps the maximum while it walks the last row): the query is
 * aligned end to end, subject overhangs on both sides are free.
 *   S[0][j] = 0,  S[i][0] = i*gap,  result = max over j in [0, slen] of S[qlen][j].
 * No committed reference output exists for this mode (generator option only): the DP definition is
 * the checker.
 */
void bgsa_oracle_dp_semiglobal(const char *queries, int64_t nq, int qlen, const char *subjects,
                               int64_t ns, int slen, int match, int mismatch, int gap, int16_t *out,
                               int threads)
{
    if (nq <= 0 || ns <= 0) return;
    uint8_t *q = map_rows(queries, nq, qlen);
    uint8_t *s = map_rows(subjects, ns, slen);
#pragma omp parallel num_threads(pick_threads(threads))
    {
        int *row = (int *)malloc(sizeof(int) * (size_t)(slen + 1));
#pragma omp for schedule(dynamic, 16) collapse(2)
        for (int64_t i = 0; i < nq; i++)
            for (int64_t j = 0; j < ns; j++) {
                const uint8_t *a = q + i * qlen, *b = s + j * slen;
                for (int x = 0; x <= slen; x++) row[x] = 0;
                for (int y = 1; y <= qlen; y++) {
                    int diag = row[0];
                    row[0] = y * gap;
                    for (int x = 1; x <= slen; x++) {
                        int up = row[x];
                        int best = diag + (a[y - 1] == b[x - 1] ? match : mismatch);
                        if (up + gap > best) best = up + gap;
                        if (row[x - 1] + gap > best) best = row[x - 1] + gap;
                        row[x] = best;
                        diag = up;
                    }
                }
                int best = row[0];
                for (int x = 1; x <= slen; x++)
                    if (row[x] > best) best = row[x];
                out[i * ns + j] = (int16_t)best;
            }
        free(row);
    }
    free(q); free(s);
}

/*
 * Closed form of the banded kernel (validated there against the compiled
 * reference for qlen == slen): unit-cost DP with D[i][0] = i, D[0][j] = 0, cells restricted to
 * diagonals j - i in [-(k+1), h]; 127 if D[m-k][m-2k-1] > k+h+1; else min over j in [m-k-1, n]
 * of D[m][j].  Only meaningful for qlen == slen > 2k+1.
 */
void bgsa_oracle_dp_banded(const char *queries, int64_t nq, int qlen, const char *subjects,
                           int64_t ns, int slen, int threshold, int8_t *out, int threads)
{
    if (nq <= 0 || ns <= 0) return;
    const int k = threshold, m = qlen, n = slen, h = k + n - m;
    const int INF = 1 << 20;
    uint8_t *q = map_rows(queries, nq, qlen);
    uint8_t *s = map_rows(subjects, ns, slen);
#pragma omp parallel num_threads(pick_threads(threads))
    {
        int *prev = (int *)malloc(sizeof(int) * (size_t)(n + 1));
        int *cur = (int *)malloc(sizeof(int) * (size_t)(n + 1));
#pragma omp for schedule(dynamic, 16) collapse(2)
        for (int64_t qi = 0; qi < nq; qi++)
            for (int64_t sj = 0; sj < ns; sj++) {
                const uint8_t *a = q + qi * m, *b = s + sj * n;
                int check = INF;
                for (int j = 0; j <= n; j++) prev[j] = (j <= h) ? 0 : INF;
                for (int i = 1; i <= m; i++) {
                    for (int j = 0; j <= n; j++) {
                        int d = j - i;
                        if (d < -(k + 1) || d > h) { cur[j] = INF; continue; }
                        if (j == 0) { cur[j] = i; continue; }
                        int best = prev[j - 1] + (a[i - 1] != b[j - 1]);
                        if (prev[j] + 1 < best) best = prev[j] + 1;
                        if (cur[j - 1] + 1 < best) best = cur[j - 1] + 1;
                        cur[j] = best;
                    }
                    if (i == m - k && m - 2 * k - 1 >= 0) check = cur[m - 2 * k - 1];
                    int *t = prev; prev = cur; cur = t;
                }
                int8_t r;
                if (check > k + h + 1) r = 127;
                else {
                    int best = INF;
                    for (int j = m - k - 1; j <= n; j++)
                        if (j >= 0 && prev[j] < best) best = prev[j];
                    r = (int8_t)best;
                }
                out[qi * ns + sj] = r;
            }
        free(prev); free(cur);
    }
    free(q); free(s);
}

/* ------------------------------------------------------------------------------------------- */
/* Timed CPU baseline: Myers global, AVX2, 8 subjects x 31 data bits                           */
/* ------------------------------------------------------------------------------------------- */

/*
 * The reference's SIMD Myers (original/BGSA_SSE/align_core.c:19-152) with 256-bit vectors —
 * what its generator emits for `-a avx2` (MyersGenerator.java:225-401 + AVX2Arch.java:48-60);
 * that output is not committed upstream, so this is our own port, checked against myers64.
 * Layout per group of 8 subjects: peq[5][nw] vectors (avx_handle_reads, BGSA_AVX2/global.c:27-71).
 */
double bgsa_oracle_myers_avx2(const char *queries, int64_t nq, int qlen, const char *subjects,
                              int64_t ns, int slen, int16_t *out, int threads)
{
    if (nq <= 0 || ns <= 0 || (ns % 8) != 0) return -1.0;
    const int nw = (slen + 32 - 2) / (32 - 1);
    const int64_t ngroups = ns / 8;
    uint8_t *q = map_rows(queries, nq, qlen);
    __m256i *peq_all = (__m256i *)_mm_malloc(sizeof(__m256i) * (size_t)ngroups * NCHAR * nw, 64);
    const int nthreads = pick_threads(threads);

#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (int64_t g = 0; g < ngroups; g++) {
        uint32_t *dst = (uint32_t *)(peq_all + g * NCHAR * nw);
        memset(dst, 0, sizeof(__m256i) * NCHAR * nw);
        for (int lane = 0; lane < 8; lane++) {
            const char *row = subjects + (g * 8 + lane) * (int64_t)(slen + 1);
            for (int p = 0; p < slen; p++)
                dst[(bgsa_oracle_map_char((uint8_t)row[p]) * nw + p / 31) * 8 + lane] |= 1u << (p % 31);
        }
    }

    const __m256i LOW = _mm256_set1_epi32(0x7fffffff);
    const __m256i ONES = _mm256_set1_epi32(-1);
    const __m256i top = _mm256_set1_epi32((int)(1u << ((slen - 1) % 31)));
    double t0 = omp_get_wtime();
#pragma omp parallel num_threads(nthreads)
    {
        __m256i *vp = (__m256i *)_mm_malloc(sizeof(__m256i) * 2 * (size_t)nw, 64);
        __m256i *vn = vp + nw;
#pragma omp for schedule(dynamic, 4) collapse(2)
        for (int64_t i = 0; i < nq; i++)
            for (int64_t g = 0; g < ngroups; g++) {
                const __m256i *peq = peq_all + g * NCHAR * nw;
                const uint8_t *qq = q + i * qlen;
                for (int w = 0; w < nw; w++) { vn[w] = _mm256_setzero_si256(); vp[w] = LOW; }
                __m256i score = _mm256_set1_epi32(slen);
                for (int r = 0; r < qlen; r++) {
                    const __m256i *eq = peq + (size_t)qq[r] * nw;
                    __m256i hp_in = _mm256_set1_epi32(1), hn_in = _mm256_setzero_si256();
                    __m256i sum = _mm256_setzero_si256();
                    for (int w = 0; w < nw; w++) {
                        __m256i pv = vp[w], mv = vn[w];
                        __m256i pm = _mm256_or_si256(eq[w], mv);
                        __m256i cin = _mm256_srli_epi32(sum, 31);
                        sum = _mm256_add_epi32(_mm256_add_epi32(_mm256_and_si256(pv, pm), pv), cin);
                        __m256i d0 = _mm256_or_si256(_mm256_xor_si256(_mm256_and_si256(sum, LOW), pv), pm);
                        __m256i hp = _mm256_or_si256(_mm256_andnot_si256(_mm256_or_si256(d0, pv), ONES), mv);
                        __m256i hn = _mm256_and_si256(d0, pv);
                        if (w == nw - 1) {
                            __m256i up = _mm256_srli_epi32(_mm256_cmpeq_epi32(_mm256_and_si256(hp, top), top), 31);
                            __m256i dn = _mm256_srli_epi32(_mm256_cmpeq_epi32(_mm256_and_si256(hn, top), top), 31);
                            score = _mm256_sub_epi32(_mm256_add_epi32(score, up), dn);
                        }
                        hp = _mm256_or_si256(_mm256_slli_epi32(hp, 1), hp_in);
                        hp_in = _mm256_srli_epi32(hp, 31);
                        hn = _mm256_or_si256(_mm256_slli_epi32(hn, 1), hn_in);
                        hn_in = _mm256_srli_epi32(hn, 31);
                        vp[w] = _mm256_and_si256(_mm256_or_si256(_mm256_andnot_si256(_mm256_or_si256(d0, hp), ONES), hn), LOW);
                        vn[w] = _mm256_and_si256(_mm256_and_si256(d0, hp), LOW);
                    }
                }
                int32_t lanes[8];
                _mm256_storeu_si256((__m256i *)lanes, score);
                for (int lane = 0; lane < 8; lane++) out[i * ns + g * 8 + lane] = (int16_t)(-lanes[lane]);
            }
        _mm_free(vp);
    }
    double t1 = omp_get_wtime();
    _mm_free(peq_all);
    free(q);
    return t1 - t0;
}
