// utils.hip -- coefficient utilities on the hot path: soft threshold and L1 norm.
//
// Path replaced:
//   soft threshold  reference w_call_soft_thresh + w_kern_soft_thresh{,_1d,_appcoeffs}
//                   (src/common.cu:13-52, 219-249): L (+1) launches of 16x16-thread blocks.
//   norm1           reference Wavelets::norm1 (src/wt.cu:398-418): 3L+1 cuBLAS asum calls, each a
//                   device->host sync, partial sums added in DTYPE.
// MI355X design: ONE launch over a device-side band table for each utility (grid-stride over
// fixed-size chunks, 16-byte accesses); the norm is per-lane double accumulation -> wave64 shuffle
// reduction -> LDS across the 4 waves -> one double partial per block -> a tiny deterministic
// second stage (no float atomics), result rounded to T once.
#include <math.h>

#include <mutex>

#include "common.hpp"

namespace pdwt {

constexpr int kUThreads = 256;
constexpr int kChunk = kUThreads * 16;  // elements per block-iteration (4 x 16-byte vectors per lane for f32)
constexpr int kMaxBands = 3 * 32 + 1;
constexpr int kMaxBlocks = 2048;        // 256 CUs x 8 blocks

template <typename T>
struct BandTable {
    T* ptr[kMaxBands];
    unsigned long long n[kMaxBands];
    unsigned int chunk0[kMaxBands + 1];  // first chunk id of each band; [nb] = total
    T beta[kMaxBands];
    int nb;
};

template <typename T> struct V16;
template <> struct V16<float> { using type = float4; static constexpr int N = 4; };
template <> struct V16<double> { using type = double2; static constexpr int N = 2; };

template <typename T> __device__ __forceinline__ T soft1(T x, T beta);
// type-correct forms (the reference calls fabsf/copysignf even in the double build, SURVEY B-3)
template <> __device__ __forceinline__ float soft1<float>(float x, float b) { return copysignf(fmaxf(fabsf(x) - b, 0.0f), x); }
template <> __device__ __forceinline__ double soft1<double>(double x, double b) { return copysign(fmax(fabs(x) - b, 0.0), x); }

__device__ __forceinline__ int find_band(const unsigned int* chunk0, int nb, unsigned int chunk)
{
    int k = 0;
    while (k + 1 < nb && chunk >= chunk0[k + 1]) k++;
    return k;
}

template <typename T, bool VEC>
__global__ __launch_bounds__(kUThreads) void k_soft_thresh(BandTable<T> tab)
{
    const unsigned int total = tab.chunk0[tab.nb];
    for (unsigned int chunk = blockIdx.x; chunk < total; chunk += gridDim.x) {
        const int k = find_band(tab.chunk0, tab.nb, chunk);
        T* __restrict__ p = tab.ptr[k];
        const unsigned long long n = tab.n[k];
        const T beta = tab.beta[k];
        const unsigned long long base = (unsigned long long)(chunk - tab.chunk0[k]) * kChunk;
        if constexpr (VEC) {
            using V = typename V16<T>::type;
            constexpr int NV = V16<T>::N;
#pragma unroll
            for (int u = 0; u < kChunk / (kUThreads * NV); u++) {
                const unsigned long long i = base + ((unsigned long long)u * kUThreads + threadIdx.x) * NV;
                if (i + NV <= n) {
                    V v = *reinterpret_cast<V*>(p + i);
                    T* e = reinterpret_cast<T*>(&v);
#pragma unroll
                    for (int q = 0; q < NV; q++) e[q] = soft1<T>(e[q], beta);
                    *reinterpret_cast<V*>(p + i) = v;
                } else {
                    for (unsigned long long j = i; j < n; j++) p[j] = soft1<T>(p[j], beta);
                }
            }
        } else {
            for (int u = 0; u < kChunk / kUThreads; u++) {
                const unsigned long long i = base + (unsigned long long)u * kUThreads + threadIdx.x;
                if (i < n) p[i] = soft1<T>(p[i], beta);
            }
        }
    }
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

template <typename T, bool VEC>
__global__ __launch_bounds__(kUThreads) void k_abs_sum(BandTable<T> tab, double* __restrict__ partial)
{
    __shared__ double s_w[kUThreads / 64];
    double acc = 0.0;
    const unsigned int total = tab.chunk0[tab.nb];
    for (unsigned int chunk = blockIdx.x; chunk < total; chunk += gridDim.x) {
        const int k = find_band(tab.chunk0, tab.nb, chunk);
        const T* __restrict__ p = tab.ptr[k];
        const unsigned long long n = tab.n[k];
        const unsigned long long base = (unsigned long long)(chunk - tab.chunk0[k]) * kChunk;
        if constexpr (VEC) {
            using V = typename V16<T>::type;
            constexpr int NV = V16<T>::N;
#pragma unroll
            for (int u = 0; u < kChunk / (kUThreads * NV); u++) {
                const unsigned long long i = base + ((unsigned long long)u * kUThreads + threadIdx.x) * NV;
                if (i + NV <= n) {
                    const V v = *reinterpret_cast<const V*>(p + i);
                    const T* e = reinterpret_cast<const T*>(&v);
#pragma unroll
                    for (int q = 0; q < NV; q++) acc += (double)(e[q] < 0 ? -e[q] : e[q]);
                } else {
                    for (unsigned long long j = i; j < n; j++) acc += (double)(p[j] < 0 ? -p[j] : p[j]);
                }
            }
        } else {
            for (int u = 0; u < kChunk / kUThreads; u++) {
                const unsigned long long i = base + (unsigned long long)u * kUThreads + threadIdx.x;
                if (i < n) acc += (double)(p[i] < 0 ? -p[i] : p[i]);
            }
        }
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
}

// second stage: one block, fixed summation order -> run-to-run deterministic
__global__ __launch_bounds__(kUThreads) void k_abs_sum_final(const double* __restrict__ partial, int n, double* __restrict__ out)
{
    __shared__ double s_w[kUThreads / 64];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += kUThreads) acc += partial[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
}

#define PDWT_CHECK_LAUNCH() PDWT_HIP_TRY(hipGetLastError())

// per-device scratch for the reduction partials (+1 slot for the result)
static std::mutex g_mu;
static double* g_partials[64] = {};
static double* partials()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_partials[dev]) {
        if (hipMalloc(&g_partials[dev], (kMaxBlocks + 8) * sizeof(double)) != hipSuccess) g_partials[dev] = nullptr;
    }
    return g_partials[dev];
}

template <typename T>
static bool table_push(BandTable<T>& t, T* p, size_t n, T beta, bool& vec_ok)
{
    if (t.nb >= kMaxBands || !p) return false;
    const int k = t.nb++;
    t.ptr[k] = p;
    t.n[k] = n;
    t.beta[k] = beta;
    t.chunk0[k + 1] = t.chunk0[k] + (unsigned int)((n + kChunk - 1) / kChunk);
    if (((uintptr_t)p & 15) != 0) vec_ok = false;
    return true;
}

// w_call_soft_thresh, src/common.cu:219-249
template <typename T>
static int soft_thresh(T** c, T beta, pdwt_info w, int do_thresh_appcoeffs, int normalize)
{
    if (!c) return PDWT_EINVAL;
    BandGeom g;
    if (band_geometry(w, &g) != PDWT_OK) return PDWT_EINVAL;
    BandTable<T> tab;
    tab.nb = 0;
    tab.chunk0[0] = 0;
    bool vec = true;
    const int per = (w.ndims == 2) ? 3 : 1;
    if (do_thresh_appcoeffs) {
        T beta2 = beta;
        if (normalize > 0) {  // beta / sqrt(2)^nlevels, src/common.cu:231-235
            const int nl2 = w.nlevels / 2;
            beta2 /= (T)(1 << nl2);
            if (nl2 * 2 != w.nlevels) beta2 = (T)(beta2 / 1.4142135623730951);
        }
        // the reference sweeps the whole level-1-sized allocation of band 0; only its first
        // Nr_L*Nc_L elements are coefficients, the rest is scratch -> threshold the coefficients only
        if (!table_push<T>(tab, c[0], (size_t)g.Nr[0] * g.Nc[0], beta2, vec)) return PDWT_EINVAL;
    }
    for (int lev = 0; lev < w.nlevels; lev++) {
        if (normalize > 0) beta = (T)(beta / 1.4142135623730951);  // src/common.cu:244
        for (int b = 0; b < per; b++) {
            const int k = per * lev + 1 + b;
            if (!table_push<T>(tab, c[k], (size_t)g.Nr[k] * g.Nc[k], beta, vec)) return PDWT_EINVAL;
        }
    }
    const unsigned int total = tab.chunk0[tab.nb];
    if (total == 0) return PDWT_OK;
    const int blocks = (int)(total < (unsigned)kMaxBlocks ? total : (unsigned)kMaxBlocks);
    KTimer kt(K_SOFT_THRESH);
    if (vec) hipLaunchKernelGGL((k_soft_thresh<T, true>), dim3(blocks), dim3(kUThreads), 0, stream(), tab);
    else hipLaunchKernelGGL((k_soft_thresh<T, false>), dim3(blocks), dim3(kUThreads), 0, stream(), tab);
    PDWT_CHECK_LAUNCH();
    return PDWT_OK;
}

// Wavelets::norm1, src/wt.cu:398-418: sum of |c| over all bands including band 0
template <typename T>
static int norm1_double(T** c, pdwt_info w, double* out)
{
    if (!c || !out) return PDWT_EINVAL;
    BandGeom g;
    if (band_geometry(w, &g) != PDWT_OK) return PDWT_EINVAL;
    BandTable<T> tab;
    tab.nb = 0;
    tab.chunk0[0] = 0;
    bool vec = true;
    for (int k = 0; k < g.nbands; k++)
        if (!table_push<T>(tab, c[k], (size_t)g.Nr[k] * g.Nc[k], T(0), vec)) return PDWT_EINVAL;
    double* part = partials();
    if (!part) return PDWT_ENOMEM;
    const unsigned int total = tab.chunk0[tab.nb];
    const int blocks = (int)(total < (unsigned)kMaxBlocks ? (total ? total : 1) : (unsigned)kMaxBlocks);
    {
        KTimer kt(K_ABS_SUM);
        if (vec) hipLaunchKernelGGL((k_abs_sum<T, true>), dim3(blocks), dim3(kUThreads), 0, stream(), tab, part);
        else hipLaunchKernelGGL((k_abs_sum<T, false>), dim3(blocks), dim3(kUThreads), 0, stream(), tab, part);
        PDWT_CHECK_LAUNCH();
    }
    {
        KTimer kt(K_ABS_SUM_FINAL);
        hipLaunchKernelGGL(k_abs_sum_final, dim3(1), dim3(kUThreads), 0, stream(), (const double*)part, blocks, part + kMaxBlocks);
        PDWT_CHECK_LAUNCH();
    }
    return pdwt_memcpy_d2h(out, part + kMaxBlocks, sizeof(double));
}

}  // namespace pdwt

using namespace pdwt;

extern "C" {
int pdwt_soft_thresh_f32(float** c, float beta, pdwt_info w, int app, int norm) { return soft_thresh<float>(c, beta, w, app, norm); }
int pdwt_soft_thresh_f64(double** c, double beta, pdwt_info w, int app, int norm) { return soft_thresh<double>(c, beta, w, app, norm); }
int pdwt_norm1_as_double_f32(float** c, pdwt_info w, double* out) { return norm1_double<float>(c, w, out); }
int pdwt_norm1_as_double_f64(double** c, pdwt_info w, double* out) { return norm1_double<double>(c, w, out); }
int pdwt_norm1_f32(float** c, pdwt_info w, float* out)
{
    double d = 0;
    int rc = norm1_double<float>(c, w, &d);
    if (rc == PDWT_OK && out) *out = (float)d;
    return out ? rc : PDWT_EINVAL;
}
int pdwt_norm1_f64(double** c, pdwt_info w, double* out) { return norm1_double<double>(c, w, out); }
}
