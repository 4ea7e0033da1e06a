// probe.hip — the sustained shader clock under a running launch, for the bench line.
//
// The kernels of this library are bound by the VALU issue rate, so their throughput is cycles x clock: the same
// binary takes the same number of cycles on every MI355X (GRBM_GUI_ACTIVE per launch is constant) but the boxes sustain
// different clocks under this load (2.12 - 2.29 GHz seen).  A reader of one bench line cannot tell a slow
// box from a regression unless the line carries the clock.  This measures it while the timed kernels run, without
// touching them: a few one-wave workgroups, started before the timed region on a stream of their own, sleep in a loop
// and read two counters — s_memtime, which counts shader clocks, and s_memrealtime, which counts the constant reference
// clock (hipDeviceAttributeWallClockRate) — until the host raises a flag or a time bound passes.  Workgroups are handed
// to the XCDs round robin, so eight of them sample all eight XCDs.  clock = d(s_memtime) / d(s_memrealtime) x reference.
#include <unistd.h>

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
constexpr int kProbeStreams = 6;
struct ProbeState {
    hipStream_t streams[kProbeStreams] = {};   // candidates: HIP multiplexes streams onto a few hardware queues
    hipStream_t stream = nullptr;              // the one the running probes are on
    ProbeRecord *d_out = nullptr;
    unsigned *d_scratch = nullptr;
    hipEvent_t progress = nullptr;
    unsigned *h_stop = nullptr;    // page-locked, mapped: the host raises it, the probes poll it
    int device = -1, n = 0;
    bool running = false;
};
ProbeState g_probe;
}  // namespace

}  // namespace bgsa

using namespace bgsa;

extern "C" {

// HIP maps its streams onto a handful of hardware queues, and a queue runs its packets in order: probes that land on the
// queue of the stream they are meant to observe do not run BESIDE its kernels, they hold them back until the probes' time
// bound (first version of this file: wall time per step x 4.6 with unchanged kernel times).  So the probes are started on
// one candidate stream after another until a marker on the CALLER's stream is seen to complete while they run.
// (One probe set per process, started and stopped by one thread: measurement plumbing, not part of the scoring path.)
int bgsa_hip_clock_probe_start(int n_probes, unsigned max_ms, void *caller_stream)
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
    if (g_probe.d_out && g_probe.device != dev) {
        for (hipStream_t &st : g_probe.streams) { if (st) (void)hipStreamDestroy(st); st = nullptr; }
        (void)hipFree(g_probe.d_out);
        (void)hipFree(g_probe.d_scratch);
        (void)hipEventDestroy(g_probe.progress);
        g_probe.d_out = nullptr;
        g_probe.d_scratch = nullptr;
        g_probe.progress = nullptr;
    }
    if (!g_probe.d_out) {
        for (hipStream_t &st : g_probe.streams) BGSA_HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        BGSA_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&g_probe.d_out), sizeof(ProbeRecord) * kProbeMax));
        BGSA_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&g_probe.d_scratch), 64));
        BGSA_HIP_TRY(hipEventCreateWithFlags(&g_probe.progress, hipEventDisableTiming));
        g_probe.device = dev;
    }
    if (!g_probe.h_stop) BGSA_HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&g_probe.h_stop), 64, hipHostMallocMapped | hipHostMallocPortable));
    unsigned *d_stop = nullptr;
    BGSA_HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void **>(&d_stop), g_probe.h_stop, 0));
    const unsigned long long max_ticks = static_cast<unsigned long long>(khz) * max_ms;
    // one iteration sleeps 127 x 64 clocks (> 3 us at any clock this chip runs at): the bound in iterations is generous
    const unsigned max_iters = max_ms >= 4000000u / 1000u ? 0xffffffffu : max_ms * 1000u;
    hipStream_t caller = static_cast<hipStream_t>(caller_stream);
    for (hipStream_t cand : g_probe.streams) {
        *static_cast<volatile unsigned *>(g_probe.h_stop) = 0u;
        BGSA_HIP_TRY(hipMemsetAsync(g_probe.d_out, 0, sizeof(ProbeRecord) * kProbeMax, cand));
        hipLaunchKernelGGL(clock_probe_kernel, dim3(n_probes), dim3(64), 0, cand, g_probe.d_out, d_stop, max_ticks, max_iters);
        BGSA_HIP_TRY(hipGetLastError());
        // does the caller's stream make progress while the probes run?
        BGSA_HIP_TRY(hipMemsetAsync(g_probe.d_scratch, 0, 4, caller));
        BGSA_HIP_TRY(hipEventRecord(g_probe.progress, caller));
        bool concurrent = false;
        for (int spin = 0; spin < 400 && !concurrent; spin++) {   // up to ~80 ms
            const hipError_t q = hipEventQuery(g_probe.progress);
            if (q == hipSuccess) concurrent = true;
            else if (q != hipErrorNotReady) {
                *static_cast<volatile unsigned *>(g_probe.h_stop) = 1u;   // do not leave probes behind an error
                set_error("hipEventQuery", q, __FILE__, __LINE__);
                return BGSA_HIP_EHIP;
            }
            else usleep(200);
        }
        if (concurrent) {
            g_probe.stream = cand;
            g_probe.n = n_probes;
            g_probe.running = true;
            return BGSA_HIP_OK;
        }
        *static_cast<volatile unsigned *>(g_probe.h_stop) = 1u;   // same hardware queue: let the probes go, try the next stream
        BGSA_HIP_TRY(hipStreamSynchronize(cand));
        BGSA_HIP_TRY(hipEventSynchronize(g_probe.progress));
    }
    set_error_text("clock probe: no stream of the library runs beside the caller's (every candidate shares its hardware queue)");
    return BGSA_HIP_EUNSUPPORTED;
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
