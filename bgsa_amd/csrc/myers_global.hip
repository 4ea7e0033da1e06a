// myers_global.hip — Myers unit-cost global alignment, one subject per lane, gfx950.
//
// Replaces the reference's align_cpu / align_sse hot loop (original/BGSA_CPU/align_core.c:54-146,
// original/BGSA_SSE/align_core.c:55-150) and its OpenMP grid (cal_cpu.c:66-84).
//
// Mapping to the machine
//   * lane  = one subject; a wavefront = one "group" of HIP_V_NUM = 64 subjects, which is exactly
//     the reference's SIMD-lane layout [group][char][word][lane] widened from 4/8/16 to 64 lanes.
//   * The subject's match masks Peq[5][NW] stay in VGPRs for the whole task; a task scores a
//     tile of queries against the group, so each Peq block is read from HBM once per tile.
//   * The query character is wave-uniform: it is fetched through the scalar cache and selects
//     one of five copies of the row body by a scalar branch, so `Eq = Peq[c][w]` costs no VALU
//     work (the reference pays a pointer add + a vector load per word, align_core.c:67,74).
//   * Words are full 32-bit (the reference keeps bit W-1 free as a software carry,
//     align_core.c:79-83,91-96): the add carry rides the hardware carry chain
//     (v_add_co/v_addc_co), and the HP/HN shift carry is one v_alignbit_b32 funnel shift.
//   * The score is not tracked per row (align_core.c:121-124); after the last row
//     D[m][n] = m + popcount(VP & mask) - popcount(VN & mask), two v_bcnt per word.
//   Per (query row, word): 10 VALU ops (v_and, v_addc_co, 5 x v_bitop3/v_or/v_and, 2 x
//   v_alignbit) against the reference's 24.
//
// Integer/bitwise only; no LDS, no MFMA.  The kernel is VALU-issue bound.
#include "bgsa_common.h"

namespace bgsa {

// One DP row: in-place update of the vertical delta vectors for query character class `eq`.
template <int NW>
__device__ __forceinline__ void myers_row(uint32_t (&vp)[NW], uint32_t (&vn)[NW],
                                          const uint32_t (&eq)[NW])
{
    uint32_t hp_prev = 0, hn_prev = 0;
    unsigned carry = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) {
        const uint32_t pv = vp[w], mv = vn[w], e = eq[w];
        const uint32_t pm = e | mv;
        // (pv & pm) == (pv & e) because pv & mv == 0 is an invariant of the recurrence.
        unsigned cout;
        const uint32_t sum = __builtin_addc(pv & e, pv, carry, &cout);
        carry = cout;
        const uint32_t d0 = (sum ^ pv) | pm;
        const uint32_t hp = ~(d0 | pv) | mv;
        const uint32_t hn = d0 & pv;
        // Shift one column along the subject; row boundary D[i][0]-D[i-1][0] = +1 enters word 0.
        const uint32_t hps = (w == 0) ? ((hp << 1) | 1u) : ((hp << 1) | (hp_prev >> 31));
        const uint32_t hns = (w == 0) ? (hn << 1) : ((hn << 1) | (hn_prev >> 31));
        hp_prev = hp;
        hn_prev = hn;
        vp[w] = ~(d0 | hps) | hns;
        vn[w] = d0 & hps;
    }
}

// grid.x = ceil(n_groups / 4), grid.y = number of query tiles; block = 4 waves = 4 groups.
template <int NW>
__global__ __launch_bounds__(256) void myers_global_kernel(
    const char *__restrict__ content, const uint32_t *__restrict__ peq, int16_t *__restrict__ out,
    int ref_len, int read_len, long long ld, int n_groups, int word_num, int ref_start,
    int ref_end, int q_tile)
{
    const int lane = threadIdx.x & (kLanes - 1);
    const int group = __builtin_amdgcn_readfirstlane(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));
    if (group >= n_groups) return;

    // Peq block of this group: [char][word][lane], coalesced 256-B rows.
    uint32_t P[kChars][NW];
    const uint32_t *g = peq + static_cast<size_t>(group) * kChars * word_num * kLanes + lane;
#pragma unroll
    for (int c = 0; c < kChars; c++)
#pragma unroll
        for (int w = 0; w < NW; w++)
            P[c][w] = (w < word_num) ? g[(c * word_num + w) * kLanes] : 0u;

    const int q0 = ref_start + blockIdx.y * q_tile;
    const int q1 = (q0 + q_tile < ref_end) ? q0 + q_tile : ref_end;
    int16_t *dst = out + static_cast<size_t>(group) * kLanes + lane;

    for (int q = q0; q < q1; q++) {
        uint32_t vp[NW], vn[NW];
#pragma unroll
        for (int w = 0; w < NW; w++) {
            vp[w] = ~0u;
            vn[w] = 0u;
        }
        UniformBytes qs(content + static_cast<size_t>(q) * (ref_len + 1));
        for (int r = 0; r < ref_len; r++) {
            if ((r & 3) == 0) qs.refill(r, ref_len - r);
            const uint32_t c = __builtin_amdgcn_readfirstlane(qs.next());
            switch (c) {
            case 0: myers_row<NW>(vp, vn, P[0]); break;
            case 1: myers_row<NW>(vp, vn, P[1]); break;
            case 2: myers_row<NW>(vp, vn, P[2]); break;
            case 3: myers_row<NW>(vp, vn, P[3]); break;
            default: myers_row<NW>(vp, vn, P[4]); break;
            }
        }
        // D[m][n] = m + sum over the n subject columns of (VP - VN).
        int score = ref_len;
#pragma unroll
        for (int w = 0; w < NW; w++) {
            const int rem = read_len - 32 * w;
            const uint32_t m = rem >= 32 ? ~0u : (rem <= 0 ? 0u : ((1u << rem) - 1u));
            score += __popc(vp[w] & m) - __popc(vn[w] & m);
        }
        dst[static_cast<size_t>(q - ref_start) * ld] = static_cast<int16_t>(-score);
    }
}

namespace {

// Register-resident word counts that are instantiated; a subject uses the smallest one that
// holds it (extra words are all-zero Peq and masked out of the score).
constexpr int kMyersNW[] = {1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16, 20, 24, 28, 32};

int pick_nw(int word_num)
{
    for (int nw : kMyersNW)
        if (nw >= word_num) return nw;
    return -1;
}

// Queries per task.  Small enough that the grid has >> 256 CUs x 8 waves of tasks even for a
// few thousand subjects, large enough that the 5*NW Peq loads are noise next to
// q_tile * ref_len * 10 * NW VALU ops.
int pick_q_tile(int nq, int64_t n_groups)
{
    int q_tile = 32;
    while (q_tile > 1 && ((nq + q_tile - 1) / q_tile) * ((n_groups + 3) / 4) < 4096) q_tile >>= 1;
    return q_tile;
}

template <int NW>
int launch_nw(const char *d_content, const uint32_t *d_peq, int16_t *d_results, int ref_len,
              int read_len, int64_t read_count, int ref_start, int ref_end, int word_num,
              hipStream_t stream)
{
    const int nq = ref_end - ref_start;
    const int64_t n_groups = read_count / kLanes;
    const int q_tile = pick_q_tile(nq, n_groups);
    dim3 grid(static_cast<unsigned>((n_groups + kWavesPerBlock - 1) / kWavesPerBlock),
              static_cast<unsigned>((nq + q_tile - 1) / q_tile));
    if (grid.y > 65535u) {
        set_error_text("myers: too many query tiles for one launch");
        return BGSA_HIP_EUNSUPPORTED;
    }
    hipLaunchKernelGGL(myers_global_kernel<NW>, grid, dim3(256), 0, stream, d_content, d_peq,
                       d_results, ref_len, read_len, static_cast<long long>(read_count),
                       static_cast<int>(n_groups), word_num, ref_start, ref_end, q_tile);
    BGSA_HIP_TRY(hipGetLastError());
    return BGSA_HIP_OK;
}

}  // namespace

const char *myers_kernel_name(int word_num)
{
    static thread_local char name[64];
    const int nw = pick_nw(word_num);
    if (nw < 0) return "myers_global_kernel<unsupported>";
    snprintf(name, sizeof name, "myers_global_kernel<%d>", nw);
    return name;
}

int launch_myers(const char *d_content, const uint32_t *d_peq, int16_t *d_results, int ref_len,
                 int read_len, int64_t read_count, int ref_start, int ref_end, int word_num,
                 hipStream_t stream)
{
    if (ref_end <= ref_start || read_count == 0) return BGSA_HIP_OK;
    switch (pick_nw(word_num)) {
#define BGSA_CASE(N)                                                                            \
    case N:                                                                                     \
        return launch_nw<N>(d_content, d_peq, d_results, ref_len, read_len, read_count,         \
                            ref_start, ref_end, word_num, stream);
        BGSA_CASE(1) BGSA_CASE(2) BGSA_CASE(3) BGSA_CASE(4) BGSA_CASE(5) BGSA_CASE(6)
        BGSA_CASE(7) BGSA_CASE(8) BGSA_CASE(10) BGSA_CASE(12) BGSA_CASE(14) BGSA_CASE(16)
        BGSA_CASE(20) BGSA_CASE(24) BGSA_CASE(28) BGSA_CASE(32)
#undef BGSA_CASE
    default:
        set_error_text("myers: subjects longer than 1024 bp are not supported yet");
        return BGSA_HIP_EUNSUPPORTED;
    }
}

}  // namespace bgsa
