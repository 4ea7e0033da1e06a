// probe.hip — the sustained shader clock under a running launch, for the bench line.
//
// The kernels of this library are bound by the VALU issue rate, so their throughput is cycles x clock: the same
// binary takes the same number of cycles on every MI355X (GRBM_GUI_ACTIVE per launch is constant) but the boxes sustain
// different clocks under this load (2.12 - 2.29 GHz seen).  A reader of one bench line cannot tell a slow
// box from a regression unless the line carries the clock.  This measures it while the timed kernels run, without
// touching them: a few one-wave workgroups, started before the timed region on a high-priority stream, sleep in a loop
// and read two counters — s_memtime, which counts shader clocks, and s_memrealtime, which counts the constant reference
// clock (hipDeviceAttributeWallClockRate) — until the host raises a flag or a time bound passes.  Workgroups are handed
// to the XCDs round robin, so eight of them sample all eight XCDs.  clock = d(s_memtime) / d(s_memrealtime) x reference.
#include "bgsa_common.h"

namespace bgsa {

constexpr int kProbeMax = 16;
struct ProbeRecord {
    unsigned long long cycles, ticks;
    unsigned xcc, iters;
};

__global__ __launch_bounds__(64) void clock_probe_kernel(ProbeRecord *__restrict__ out, const unsigned *__restrict__ stop,
                                                         unsigned long long max_ticks, unsigned max_iters)
{
    if (threadIdx.x != 0) return;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
    unsigned long long t1 = t0, c1 = c0;
    unsigned i = 0;
    // exit conditions every probe reaches on its own: the reference clock passes the bound, or the iteration count does
    for (; i < max_iters && t1 - t0 < max_ticks; i++) {
        __builtin_amdgcn_s_sleep(127);
        c1 = __builtin_amdgcn_s_memtime();
        t1 = __builtin_amdgcn_s_memrealtime();
        if (__hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) break;
    }
    out[blockIdx.x] = ProbeRecord{c1 - c0, t1 - t0, xcc & 0xfu, i};
}

namespace {
struct ProbeState {
    hipStream_t stream = nullptr;
    ProbeRecord *d_out = nullptr;
    unsigned *h_stop = nullptr;    // page-locked, mapped: the host raises it, the probes poll it
    int device = -1, n = 0;
    bool running = false;
};
ProbeState g_probe;
}  // namespace

}  // namespace bgsa

using namespace bgsa;

extern "C" {

int bgsa_hip_clock_probe_start(int n_probes, unsigned max_ms)
{
    if (n_probes < 1 || n_probes > kProbeMax || max_ms < 1 || max_ms > 600000) {
        set_error_text("clock probe: 1..16 probes, 1..600000 ms");
        return BGSA_HIP_EINVAL;
    }
    if (g_probe.running) {
        set_error_text("clock probe: already running");
        return BGSA_HIP_EINVAL;
    }
    int dev = 0, khz = 0;
    BGSA_HIP_TRY(hipGetDevice(&dev));
    BGSA_HIP_TRY(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev));
    if (khz <= 0) {
        set_error_text("clock probe: the device reports no wall clock rate");
        return BGSA_HIP_EUNSUPPORTED;
    }
    if (g_probe.stream && g_probe.device != dev) {
        (void)hipStreamDestroy(g_probe.stream);
        (void)hipFree(g_probe.d_out);
        g_probe.stream = nullptr;
        g_probe.d_out = nullptr;
    }
    if (!g_probe.stream) {
        int lo = 0, hi = 0;
        BGSA_HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
        BGSA_HIP_TRY(hipStreamCreateWithPriority(&g_probe.stream, hipStreamNonBlocking, hi));
        BGSA_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&g_probe.d_out), sizeof(ProbeRecord) * kProbeMax));
        g_probe.device = dev;
    }
    if (!g_probe.h_stop) BGSA_HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&g_probe.h_stop), 64, hipHostMallocMapped | hipHostMallocPortable));
    *static_cast<volatile unsigned *>(g_probe.h_stop) = 0u;
    unsigned *d_stop = nullptr;
    BGSA_HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void **>(&d_stop), g_probe.h_stop, 0));
    BGSA_HIP_TRY(hipMemsetAsync(g_probe.d_out, 0, sizeof(ProbeRecord) * kProbeMax, g_probe.stream));
    const unsigned long long max_ticks = static_cast<unsigned long long>(khz) * max_ms;
    // one iteration sleeps 127 x 64 clocks (> 3 us at any clock this chip runs at): the bound in iterations is generous
    const unsigned max_iters = max_ms >= 4000000u / 1000u ? 0xffffffffu : max_ms * 1000u;
    hipLaunchKernelGGL(clock_probe_kernel, dim3(n_probes), dim3(64), 0, g_probe.stream, g_probe.d_out, d_stop, max_ticks, max_iters);
    BGSA_HIP_TRY(hipGetLastError());
    g_probe.n = n_probes;
    g_probe.running = true;
    return BGSA_HIP_OK;
}

int bgsa_hip_clock_probe_stop(double *mhz, int *xcc, int cap, int *n_out, double *seconds)
{
    if (!g_probe.running) {
        set_error_text("clock probe: not running");
        return BGSA_HIP_EINVAL;
    }
    g_probe.running = false;
    *static_cast<volatile unsigned *>(g_probe.h_stop) = 1u;
    BGSA_HIP_TRY(hipStreamSynchronize(g_probe.stream));
    ProbeRecord rec[kProbeMax];
    BGSA_HIP_TRY(hipMemcpy(rec, g_probe.d_out, sizeof(ProbeRecord) * g_probe.n, hipMemcpyDeviceToHost));
    int khz = 0;
    BGSA_HIP_TRY(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, g_probe.device));
    int n = 0;
    double longest = 0.0;
    for (int i = 0; i < g_probe.n && n < cap; i++) {
        if (rec[i].ticks == 0) continue;   // a probe that never got onto the chip
        const double secs = static_cast<double>(rec[i].ticks) / (khz * 1e3);
        if (mhz) mhz[n] = static_cast<double>(rec[i].cycles) / secs / 1e6;
        if (xcc) xcc[n] = static_cast<int>(rec[i].xcc);
        if (secs > longest) longest = secs;
        n++;
    }
    if (n_out) *n_out = n;
    if (seconds) *seconds = longest;
    return BGSA_HIP_OK;
}

}  // extern "C"
