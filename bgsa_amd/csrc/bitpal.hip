// bitpal.hip — BitPAl packed scoring on gfx950: the registry of compiled score sets and the dispatch.
//
// The reference generates one kernel per (match, mismatch, gap) with its Java generator
// (generator/.../BitPAlGenerator.java, `-M -I -G`, README.md:26-82) and commits only 2/-3/-5
// (original/BGSA_AVX2/align_core.c).  Here gen_bitpal_sets.py plays that role at build time: for
// every score set in the Makefile's BITPAL_SETS it emits the row loops (rows_ir.py: bitpal_body) and a
// translation unit that instantiates bitpal_kernels.inl for it; bitpal_sets.inc lists them.  Which set
// a call runs is decided in capi.hip (make_plan) from the caller's scores.
#include <stdlib.h>

#include "bgsa_common.h"

namespace bgsa {

#if BGSA_AB_KERNELS
#include "_gen/bitpal_sets_ab.inc"   // accessor declarations + kBitpalSets[] (generated): the sets of BITPAL_SETS_AB
#else
#include "_gen/bitpal_sets.inc"      // ... of BITPAL_SETS
#endif

int bitpal_set_count() { return static_cast<int>(sizeof(kBitpalSets) / sizeof(kBitpalSets[0])); }
const BitpalSet *bitpal_set_at(int i) { return (i >= 0 && i < bitpal_set_count()) ? kBitpalSets[i]() : nullptr; }

const BitpalSet *bitpal_find_set(int match, int mismatch, int gap)
{
    for (auto get : kBitpalSets) {
        const BitpalSet *s = get();
        if (s->match == match && s->mismatch == mismatch && s->gap == gap) return s;
    }
    return nullptr;
}

// commonFactor() of the generator (Main.java:213-238): the largest integer dividing all three
int bitpal_common_factor(int match, int mismatch, int gap)
{
    auto g = [](int x, int y) {
        x = x < 0 ? -x : x;
        y = y < 0 ? -y : y;
        while (y) { const int r = x % y; x = y; y = r; }
        return x;
    };
    const int f = g(g(match, mismatch), gap);
    return f < 1 ? 1 : f;
}

namespace {
bool bitpal_c_impl(const BitpalSet *s)
{
    static const bool want = [] {
        const char *e = getenv("BGSA_BITPAL_IMPL");
        return e && e[0] == 'c';
    }();
    // the compiler-scheduled state-in-memory kernel (long_kernels.hip) exists for 2/-3/-5 only
    return want && s->match == 2 && s->mismatch == -3 && s->gap == -5;
}
}  // namespace

const char *bitpal_kernel_name(const BitpalSet *s, int word_num)
{
    if (!s) return "bitpal: score set not compiled";
    if (word_num > s->max_plain && bitpal_c_impl(s)) return "bitpal_long_kernel";
    return s->kernel_name(word_num);
}

int launch_bitpal(const BitpalSet *s, const char *d_content, const uint32_t *d_peq, int16_t *d_results, int ref_len,
                  int read_len, int64_t read_count, int ref_start, int ref_end, int word_num,
                  void *d_workspace, hipStream_t stream, int semi_global)
{
    if (ref_end <= ref_start || read_count == 0) return BGSA_HIP_OK;
#if BGSA_AB_KERNELS
    if (word_num > s->max_plain && bitpal_c_impl(s) && !semi_global)  // A/B: the state-in-memory C++ kernel
        return launch_long(BGSA_ALGO_BITPAL, d_content, d_peq, d_results, ref_len, read_len, read_count, ref_start,
                           ref_end, word_num, d_workspace, stream);
#else
    if (word_num > s->max_plain && bitpal_c_impl(s) && !semi_global) return ab_knob_refused("BGSA_BITPAL_IMPL=c");
#endif
    return s->launch(d_content, d_peq, d_results, ref_len, read_len, read_count, ref_start, ref_end, word_num,
                     d_workspace, stream, semi_global);
}

}  // namespace bgsa
