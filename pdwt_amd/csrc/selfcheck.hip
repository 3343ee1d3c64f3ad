// selfcheck.hip -- one-time check of the hardware behaviour the hand-counted memory pipelines rely on (stream_dev.hpp):
//   * a wave's global loads and stores retire IN ORDER with respect to each other, so `s_waitcnt vmcnt(N)` with N younger stores
//     outstanding means the older loads have landed;
//   * a store issued with EXEC = 0 (the loop forms of the cascade kernels store rows a wave does not own that way) still takes its
//     place in that order and in the count.
// Neither is documented; both were established with tools/probes/vmcnt_order.hip on gfx950 (MI355X, ROCm 7.2).  pdwt_selfcheck_vmcnt_order()
// runs a compact form of that probe on the current device: a non-zero count of stale registers means the counted waits under-count
// there.  The PRODUCT runs it by itself: counted_waits_ok() -- asked by every dispatcher of a hand-counted kernel (dwt_stream.hip:
// stream_enabled(), which also gates the cascade kernels; the fused SWT levels) -- runs it once per device on first use, caches the
// verdict and, when stale registers were seen, sends every such geometry to the compiler-counted kernels (LDS-tiled / two-pass) with
// one warning on stderr.  PDWT_SELFCHECK=0 skips the check (the build itself refuses any other architecture, stream_dev.hpp).
#include <atomic>
#include <cstdio>
#include <mutex>

#include "common.hpp"

namespace pdwt {
namespace {
typedef float sc_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float sc_val(size_t i) { return (float)(i % 1000003u); }

__global__ __launch_bounds__(256) void k_sc_fill(float* tab, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) tab[i] = sc_val(i);
}
// per step: 2 loads (far apart), NST stores -- every second step with all lanes off --, vmcnt(NST), check the loaded registers
template <int NST>
__global__ __launch_bounds__(256) void k_sc_order(const float* __restrict__ tab, size_t n4, float* sink, size_t sink4, int steps, unsigned long long* bad)
{
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nthr = (size_t)gridDim.x * blockDim.x;
    sc_v4f a = {0, 0, 0, 0}, b = {0, 0, 0, 0};
    unsigned long long nbad = 0;
    size_t i0 = gid % n4, i1 = (gid + n4 / 2) % n4;
    asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(a) : "v"(tab + 4 * i0) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(b) : "v"(tab + 4 * i1) : "memory");
    for (int s = 0; s < steps; s++) {
        if (s > 0) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(NST) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b) : : "memory");
        sc_v4f ca, cb;  // opaque copies out of the load registers
        asm volatile("v_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7" : "=&v"(ca.x), "=&v"(ca.y), "=&v"(ca.z), "=&v"(ca.w) : "v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w));
        asm volatile("v_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7" : "=&v"(cb.x), "=&v"(cb.y), "=&v"(cb.z), "=&v"(cb.w) : "v"(b.x), "v"(b.y), "v"(b.z), "v"(b.w));
        const size_t c0 = i0, c1 = i1;
        i0 = (i0 + nthr) % n4;
        i1 = (i1 + nthr) % n4;
        asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(a) : "v"(tab + 4 * i0) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(b) : "v"(tab + 4 * i1) : "memory");
        nbad += (ca.x != sc_val(4 * c0)) + (ca.w != sc_val(4 * c0 + 3)) + (cb.x != sc_val(4 * c1)) + (cb.w != sc_val(4 * c1 + 3));
        const sc_v4f o = ca + cb;
#pragma unroll
        for (int q = 0; q < NST; q++) {
            float* p = sink + 4 * ((gid + (size_t)(s * NST + q) * nthr) % sink4);
            unsigned long long saved, mask = (s & 1) ? 0ull : ~0ull;
            asm volatile("s_and_saveexec_b64 %0, %3\n\tglobal_store_dwordx4 %1, %2, off\n\ts_nop 1\n\ts_mov_b64 exec, %0" : "=&s"(saved) : "v"(p), "v"(o), "s"(mask) : "memory", "scc");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b) : : "memory");
    if (nbad) atomicAdd(bad, nbad);
}
}  // namespace
}  // namespace pdwt

using namespace pdwt;

namespace pdwt {
static long long selfcheck_run(size_t table_mib, size_t sink_mib);
bool counted_waits_ok()
{
    static std::atomic<int> verdict[64];  // 0 = not yet asked, 1 = ok, 2 = stale registers seen
    static std::mutex mu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return true;
    if (knob(KN_SELFCHECK) == 2) return false;  // (test hook: behave as if the check had FAILED on this device -- the fallback path must stay correct)
    int v = verdict[dev].load(std::memory_order_acquire);
    if (v) return v == 1;
    if (knob(KN_SELFCHECK) == 0) return true;
    // (a check needs allocations and a host read-back: not inside a stream capture -- that call keeps the default, the next one outside decides)
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream(), &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
        (void)hipGetLastError();
        return true;
    }
    std::lock_guard<std::mutex> lk(mu);
    v = verdict[dev].load(std::memory_order_acquire);
    if (v) return v == 1;
    // the implicit check runs the SMALL probe (64 + 64 MiB, ~3 ms: it sits inside somebody's first forward() / inverse(), possibly on a
    // nearly full device); the full-size one (256 MiB table: every load misses the caches) stays with pdwt_selfcheck_vmcnt_order()
    long long bad = selfcheck_run(64, 64);
    if (bad < 0) bad = selfcheck_run(8, 8);
    if (bad < 0) {  // the probe could not run at all (no memory): no verdict is cached -- the next call asks again -- and it is said once
        static std::atomic<int> warned{0};
        if (!warned.exchange(1))
            fprintf(stderr, "pdwt: the self-check of the hand-counted waits could not run on device %d (error %lld); keeping the gfx950 build's kernels, will retry\n", dev, bad);
        return true;
    }
    if (bad > 0)
        fprintf(stderr, "pdwt: device %d retires loads and stores out of the order the hand-counted waits assume (%lld stale registers in the self-check): "
                        "using the compiler-counted kernels\n", dev, bad);
    verdict[dev].store(bad > 0 ? 2 : 1, std::memory_order_release);
    return bad <= 0;
}
}  // namespace pdwt

extern "C" long long pdwt_selfcheck_vmcnt_order(void) { return pdwt::selfcheck_run(256, 512); }  // 256 MiB table (loads miss the caches), 512 MiB store target

static long long pdwt::selfcheck_run(size_t table_mib, size_t sink_mib)
{
    const size_t n4 = (table_mib << 20) / 16, sink4 = (sink_mib << 20) / 16;
    float* tab = (float*)pdwt_malloc(n4 * 16);
    float* sink = (float*)pdwt_malloc(sink4 * 16);
    unsigned long long* bad = (unsigned long long*)pdwt_malloc(8);
    long long result = PDWT_ENOMEM;
    if (tab && sink && bad && pdwt_memset(bad, 0, 8) == PDWT_OK) {
        hipLaunchKernelGGL(k_sc_fill, dim3(4096), dim3(256), 0, stream(), tab, 4 * n4);
        hipLaunchKernelGGL(k_sc_order<4>, dim3(2048), dim3(256), 0, stream(), (const float*)tab, n4, sink, sink4, 200, bad);
        hipLaunchKernelGGL(k_sc_order<1>, dim3(2048), dim3(256), 0, stream(), (const float*)tab, n4, sink, sink4, 200, bad);
        unsigned long long h = 0;
        if (hipGetLastError() == hipSuccess && pdwt_memcpy_d2h(&h, bad, 8) == PDWT_OK) result = (long long)h;
        else result = PDWT_EHIP;
    }
    pdwt_free(tab);
    pdwt_free(sink);
    pdwt_free(bad);
    return result;
}
