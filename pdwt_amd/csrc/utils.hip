// utils.hip -- coefficient utilities: thresholds / projections / scaling, norms, coefficient axpy, circular shift.
//
// Path replaced:
//   hard / group-soft threshold, proj_linf, shrink   reference w_call_hard_thresh, w_call_group_soft_thresh,
//                   w_call_proj_linf, w_shrink (src/common.cu:57-198, 252-371): L (+1) launches (or 3L+1 cuBLAS scal)
//   norm2sq         reference Wavelets::norm2sq (src/wt.cu:370-395): 3L+1 cuBLAS nrm2 calls
//   add_coeffs      reference w_add_coeffs{,_1d} (src/common.cu:499-526): 3L+1 cuBLAS axpy calls
//   circshift       reference w_kern_circshift + w_call_circshift (src/common.cu:202-211, 378-396)
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

// elementwise operators of the threshold family; `b` is the per-band parameter of the table
enum EwOp { OP_SOFT = 0, OP_HARD, OP_PROJ, OP_SCALE };
__device__ __forceinline__ float abs_t(float x) { return fabsf(x); }
__device__ __forceinline__ double abs_t(double x) { return fabs(x); }
__device__ __forceinline__ float copysign_t(float a, float s) { return copysignf(a, s); }
__device__ __forceinline__ double copysign_t(double a, double s) { return copysign(a, s); }
// type-correct forms (the reference calls fabsf/copysignf even in the double build, SURVEY B-3)
template <int OP, typename T>
__device__ __forceinline__ T ew_op(T x, T b)
{
    if constexpr (OP == OP_SOFT) {  // src/common.cu:19: copysign(max(|x|-b, 0), x)
        const T m = abs_t(x) - b;
        return copysign_t(m > T(0) ? m : T(0), x);
    } else if constexpr (OP == OP_HARD) {  // src/common.cu:63: max(W_SIGN(|x|-b), 0)*x  -> x if |x| > b, else 0*x (keeps the sign of zero)
        return (abs_t(x) - b > T(0) ? T(1) : T(0)) * x;
    } else if constexpr (OP == OP_PROJ) {  // src/common.cu:107: copysign(min(|x|, b), x)
        const T a = abs_t(x);
        return copysign_t(a < b ? a : b, x);
    } else {  // OP_SCALE, cublas scal: x * b
        return x * b;
    }
}
template <typename T> __device__ __forceinline__ T soft1(T x, T beta) { return ew_op<OP_SOFT, T>(x, beta); }

__device__ __forceinline__ int find_band(const unsigned int* chunk0, int nb, unsigned int chunk)
{
    int k = 0;
    while (k + 1 < nb && chunk >= chunk0[k + 1]) k++;
    return k;
}

template <typename T, bool VEC, int OP>
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
                    for (int q = 0; q < NV; q++) e[q] = ew_op<OP, T>(e[q], beta);
                    *reinterpret_cast<V*>(p + i) = v;
                } else {
                    for (unsigned long long j = i; j < n; j++) p[j] = ew_op<OP, T>(p[j], beta);
                }
            }
        } else {
            for (int u = 0; u < kChunk / kUThreads; u++) {
                const unsigned long long i = base + (unsigned long long)u * kUThreads + threadIdx.x;
                if (i < n) p[i] = ew_op<OP, T>(p[i], beta);
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

// Soft threshold that also leaves  sum |c|  over ALL bands of the table (after the threshold) behind: the norm1() that
// usually follows a threshold (reference call sequence: soft_threshold -> norm1, src/demo.cpp:203-205, and every
// proximal-gradient loop built on the class) then costs one 8-byte copy instead of a second pass over every band.
// Bands with beta < 0 are only summed (the approximation band when it is not thresholded).
// scratch: [0, kMaxBlocks) per-block partials, [kMaxBlocks] result, [kMaxBlocks+1] arrival counter (as double bits are
// not used: the counter lives in the first 4 bytes).  The LAST block to arrive adds the partials in a fixed order, so the
// result does not depend on the order in which blocks finish.
template <typename T, bool VEC>
__global__ __launch_bounds__(kUThreads) void k_soft_thresh_sum(BandTable<T> tab, double* __restrict__ scratch)
{
    __shared__ double s_w[kUThreads / 64];
    __shared__ int s_last;
    constexpr int NVV = VEC ? V16<T>::N : 1;
    constexpr int U = kChunk / (kUThreads * NVV);
    double accs[VEC ? U : 1];  // one partial per unrolled position: independent add chains, all loads of a chunk in flight together
#pragma unroll
    for (int u = 0; u < (VEC ? U : 1); u++) accs[u] = 0.0;
    const unsigned int total = tab.chunk0[tab.nb];
    for (unsigned int chunk = blockIdx.x; chunk < total; chunk += gridDim.x) {
        const int k = find_band(tab.chunk0, tab.nb, chunk);
        T* __restrict__ p = tab.ptr[k];
        const unsigned long long n = tab.n[k];
        const T beta = tab.beta[k];
        const bool modify = !(beta < T(0));
        const T beta_eff = modify ? beta : T(0);  // soft threshold with beta = 0 is the identity: one code path, stores predicated
        const unsigned long long base = (unsigned long long)(chunk - tab.chunk0[k]) * kChunk;
        if constexpr (VEC) {
            using V = typename V16<T>::type;
            constexpr int NV = V16<T>::N;
            if (base + kChunk <= n) {  // full chunk: no bounds checks, every load issued before the first use
                V v[U];
#pragma unroll
                for (int u = 0; u < U; u++) v[u] = *reinterpret_cast<const V*>(p + base + ((unsigned long long)u * kUThreads + threadIdx.x) * NV);
#pragma unroll
                for (int u = 0; u < U; u++) {
                    T* e = reinterpret_cast<T*>(&v[u]);
                    double a = 0.0;
#pragma unroll
                    for (int q = 0; q < NV; q++) {
                        e[q] = ew_op<OP_SOFT, T>(e[q], beta_eff);
                        a += (double)abs_t(e[q]);
                    }
                    accs[u] += a;
                    if (modify) *reinterpret_cast<V*>(p + base + ((unsigned long long)u * kUThreads + threadIdx.x) * NV) = v[u];
                }
            } else {
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const unsigned long long i = base + ((unsigned long long)u * kUThreads + threadIdx.x) * NV;
                    if (i + NV <= n) {
                        V v = *reinterpret_cast<V*>(p + i);
                        T* e = reinterpret_cast<T*>(&v);
#pragma unroll
                        for (int q = 0; q < NV; q++) {
                            e[q] = ew_op<OP_SOFT, T>(e[q], beta_eff);
                            accs[u] += (double)abs_t(e[q]);
                        }
                        if (modify) *reinterpret_cast<V*>(p + i) = v;
                    } else {
                        for (unsigned long long j = i; j < n; j++) {
                            const T x = ew_op<OP_SOFT, T>(p[j], beta_eff);
                            if (modify) p[j] = x;
                            accs[u] += (double)abs_t(x);
                        }
                    }
                }
            }
        } else {
            for (int u = 0; u < kChunk / kUThreads; u++) {
                const unsigned long long i = base + (unsigned long long)u * kUThreads + threadIdx.x;
                if (i < n) {
                    const T x = ew_op<OP_SOFT, T>(p[i], beta_eff);
                    if (modify) p[i] = x;
                    accs[0] += (double)abs_t(x);
                }
            }
        }
    }
    double acc = 0.0;
#pragma unroll
    for (int u = 0; u < (VEC ? U : 1); u++) acc += accs[u];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        // publish the partial WITHOUT a release fence: an agent-scope release (buffer_wbl2) would write back every dirty line
        // of this XCD's L2 -- i.e. the thresholded coefficients all blocks have just stored -- once per block (measured: 306 us
        // instead of 190).  A write-through (agent-scope atomic) store, drained before the ticket is taken, orders the two.
        __hip_atomic_store(scratch + blockIdx.x, (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned int ticket = __hip_atomic_fetch_add(reinterpret_cast<unsigned int*>(scratch + kMaxBlocks + 1), 1u, __ATOMIC_RELAXED,
                                                           __HIP_MEMORY_SCOPE_AGENT);
        s_last = (ticket == gridDim.x - 1);
    }
    __syncthreads();
    if (s_last) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // this CU's L1 only; the partials are read past it (agent-scope loads)
        double a = 0.0;
        for (int i = threadIdx.x; i < (int)gridDim.x; i += kUThreads) a += __hip_atomic_load(scratch + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a = wave_sum(a);
        if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = a;
        __syncthreads();
        if (threadIdx.x == 0) {
            scratch[kMaxBlocks] = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
            __hip_atomic_store(reinterpret_cast<unsigned int*>(scratch + kMaxBlocks + 1), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // next launch
        }
    }
}

// A pure read stream: every block owns ONE contiguous run of chunks (the walk that streams best on this chip, profiles/r04_hbm_ceiling.md:
// 6.2 TB/s against 5.0-5.6 for a grid-stride walk), walks it from its END (the pass that wrote the bands -- a threshold, a forward
// level -- ran front to back, so the tail is what is still in the Infinity Cache), loads non-temporally (7.0 TB/s read-only with the
// hint; nothing here is read twice) and issues all loads of a chunk before the first use.  Fixed assignment and order: deterministic.
template <typename T, bool VEC>
__global__ __launch_bounds__(kUThreads) void k_abs_sum(BandTable<T> tab, double* __restrict__ partial)
{
    __shared__ double s_w[kUThreads / 64];
    double acc = 0.0;
    const unsigned int total = tab.chunk0[tab.nb];
    const unsigned int per = (total + gridDim.x - 1) / gridDim.x;
    // block b <-> chunks [total - (b+1)*per, total - b*per): block 0 starts with the very last chunk
    const long long hi = (long long)total - (long long)blockIdx.x * per, lo = hi - per > 0 ? hi - per : 0;
    for (long long cc = hi - 1; cc >= lo; cc--) {
        const unsigned int chunk = (unsigned int)cc;
        const int k = find_band(tab.chunk0, tab.nb, chunk);
        const T* __restrict__ p = tab.ptr[k];
        const unsigned long long n = tab.n[k];
        const bool sq = tab.beta[k] != T(0);  // per-band mode: 0 -> sum |x| (norm1), else sum x^2 (norm2sq)
        const unsigned long long base = (unsigned long long)(chunk - tab.chunk0[k]) * kChunk;
        if constexpr (VEC) {
            using V = typename V16<T>::type;
            constexpr int NV = V16<T>::N;
            constexpr int U = kChunk / (kUThreads * NV);
            if (base + kChunk <= n) {  // full chunk: no bounds checks, every load in flight before the first use
                typedef T NTV __attribute__((ext_vector_type(NV)));  // (the builtin wants a native vector type)
                NTV v[U];
#pragma unroll
                for (int u = 0; u < U; u++) v[u] = __builtin_nontemporal_load(reinterpret_cast<const NTV*>(p + base + ((unsigned long long)u * kUThreads + threadIdx.x) * NV));
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const T* e = reinterpret_cast<const T*>(&v[u]);
#pragma unroll
                    for (int q = 0; q < NV; q++) acc += sq ? (double)e[q] * (double)e[q] : (double)(e[q] < 0 ? -e[q] : e[q]);
                }
            } else {
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const unsigned long long i = base + ((unsigned long long)u * kUThreads + threadIdx.x) * NV;
                    if (i + NV <= n) {
                        const V v = *reinterpret_cast<const V*>(p + i);
                        const T* e = reinterpret_cast<const T*>(&v);
#pragma unroll
                        for (int q = 0; q < NV; q++) acc += sq ? (double)e[q] * (double)e[q] : (double)(e[q] < 0 ? -e[q] : e[q]);
                    } else {
                        for (unsigned long long j = i; j < n; j++) acc += sq ? (double)p[j] * (double)p[j] : (double)(p[j] < 0 ? -p[j] : p[j]);
                    }
                }
            }
        } else {
            for (int u = 0; u < kChunk / kUThreads; u++) {
                const unsigned long long i = base + (unsigned long long)u * kUThreads + threadIdx.x;
                if (i < n) acc += sq ? (double)p[i] * (double)p[i] : (double)(p[i] < 0 ? -p[i] : p[i]);
            }
        }
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = acc;
    __syncthreads();
    // the LAST block to arrive adds the partials, in the fixed order the separate one-block launch used (k_abs_sum_final: 8 us of launch and
    // boundary for 2048 doubles): publish the partial with a write-through store, drain it, take a ticket (cf. k_soft_thresh_sum)
    __shared__ int s_last;
    if (threadIdx.x == 0) {
        __hip_atomic_store(partial + blockIdx.x, (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned int ticket = __hip_atomic_fetch_add(reinterpret_cast<unsigned int*>(partial + kMaxBlocks + 1), 1u, __ATOMIC_RELAXED,
                                                           __HIP_MEMORY_SCOPE_AGENT);
        s_last = (ticket == gridDim.x - 1);
    }
    __syncthreads();
    if (s_last) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        double a = 0.0;
        for (int i = threadIdx.x; i < (int)gridDim.x; i += kUThreads) a += __hip_atomic_load(partial + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a = wave_sum(a);
        if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = a;
        __syncthreads();
        if (threadIdx.x == 0) {
            partial[kMaxBlocks] = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
            __hip_atomic_store(reinterpret_cast<unsigned int*>(partial + kMaxBlocks + 1), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // next launch
        }
    }
}


// group soft threshold (src/common.cu:134-198): one scale factor per position from the l2 norm of the
// detail coefficients there (and of the approximation at the last scale when asked), applied to all of them.
template <typename T>
struct GroupTable {
    T* h[32];
    T* v[32];
    T* d[32];
    T* a[32];  // NULL except at the last scale with do_thresh_appcoeffs
    unsigned long long n[32];
    unsigned int chunk0[33];
    T beta[32];
    int nb;
};
__device__ __forceinline__ float sqrt_t(float x) { return sqrtf(x); }
__device__ __forceinline__ double sqrt_t(double x) { return sqrt(x); }  // (the reference calls sqrtf in the double build too, B-3)

template <typename T>
__global__ __launch_bounds__(kUThreads) void k_group_soft_thresh(GroupTable<T> tab)
{
    const unsigned int total = tab.chunk0[tab.nb];
    for (unsigned int chunk = blockIdx.x; chunk < total; chunk += gridDim.x) {
        const int k = find_band(tab.chunk0, tab.nb, chunk);
        T* __restrict__ ph = tab.h[k];
        T* __restrict__ pv = tab.v[k];
        T* __restrict__ pd = tab.d[k];
        T* __restrict__ pa = tab.a[k];
        const unsigned long long n = tab.n[k];
        const T beta = tab.beta[k];
        const unsigned long long base = (unsigned long long)(chunk - tab.chunk0[k]) * kChunk;
        for (int u = 0; u < kChunk / kUThreads; u++) {
            const unsigned long long i = base + (unsigned long long)u * kUThreads + threadIdx.x;
            if (i < n) {
                const T vd = pd[i];
                const T vh = ph ? ph[i] : T(0), vv = pv ? pv[i] : T(0);
                // same association as the reference: h*h + v*v + d*d (+ a*a), plain products (no contraction across
                // the sum in the reference's nvcc build would change nothing beyond 1 ulp of the norm)
                T nrm = ph ? vh * vh + vv * vv + vd * vd : vd * vd;
                T va = T(0);
                if (pa) {
                    va = pa[i];
                    nrm += va * va;
                }
                nrm = sqrt_t(nrm);
                T res = T(0);
                if (nrm != T(0)) {
                    res = T(1) - beta / nrm;
                    if (!(res > T(0))) res = T(0);
                }
                if (ph) {
                    ph[i] = vh * res;
                    pv[i] = vv * res;
                }
                pd[i] = vd * res;
                if (pa) pa[i] = va * res;
            }
        }
    }
}

// dst += alpha * src over a table of band pairs (cublas axpy per band in the reference)
template <typename T>
struct PairTable {
    T* dst[kMaxBands];
    const T* src[kMaxBands];
    unsigned long long n[kMaxBands];
    unsigned int chunk0[kMaxBands + 1];
    int nb;
};
template <typename T>
__global__ __launch_bounds__(kUThreads) void k_axpy_bands(PairTable<T> tab, T alpha)
{
    const unsigned int total = tab.chunk0[tab.nb];
    for (unsigned int chunk = blockIdx.x; chunk < total; chunk += gridDim.x) {
        const int k = find_band(tab.chunk0, tab.nb, chunk);
        T* __restrict__ d = tab.dst[k];
        const T* __restrict__ sp = tab.src[k];
        const unsigned long long n = tab.n[k];
        const unsigned long long base = (unsigned long long)(chunk - tab.chunk0[k]) * kChunk;
        for (int u = 0; u < kChunk / kUThreads; u++) {
            const unsigned long long i = base + (unsigned long long)u * kUThreads + threadIdx.x;
            if (i < n) d[i] = fma_t<T>(alpha, sp[i], d[i]);
        }
    }
}

// out[y][x] = in[(y - sr) mod Nr][(x - sc) mod Nc], 0 <= sr < Nr, 0 <= sc < Nc  (src/common.cu:202-211)
template <typename T>
__global__ __launch_bounds__(kUThreads) void k_circshift(const T* __restrict__ in, T* __restrict__ out, int Nr, int Nc, int sr, int sc)
{
    const int x = blockIdx.x * kUThreads + threadIdx.x;
    if (x >= Nc) return;
    int c = x - sc;
    if (c < 0) c += Nc;
    for (int y = blockIdx.y; y < Nr; y += gridDim.y) {
        int r = y - sr;
        if (r < 0) r += Nr;
        out[(size_t)y * Nc + x] = in[(size_t)r * Nc + c];
    }
}

#define PDWT_CHECK_LAUNCH() PDWT_HIP_TRY(hipGetLastError())

// per-device scratch for the reduction partials (+1 slot for the result).  One buffer per device shared by every
// host thread: a reduction holds g_red_mu[dev] from its first launch until its result has been copied out, so two
// threads reducing on one device cannot interleave their partials.
static std::mutex g_mu;
static std::mutex g_red_mu[64];
static double* g_partials[64] = {};
// A CALLER's scratch (pdwt_norm1_enqueue_*, pdwt_soft_thresh_sum_*): the last-block ticket of the reduction kernels lives at element
// kMaxBlocks + 1 and must be 0 when a launch starts.  The kernels leave it at 0, but nothing says what a caller's buffer held before its
// first use (a non-zero ticket never matches gridDim.x - 1: the result slot would keep its old value, with PDWT_OK) -- so it is zeroed
// here, stream-ordered in front of every launch on such a buffer (8 bytes; the library's own per-device partials skip it).
static int reset_ticket(double* scratch)
{
    PDWT_HIP_TRY(hipMemsetAsync(scratch + kMaxBlocks + 1, 0, sizeof(double), stream()));
    return PDWT_OK;
}
static double* partials(int* dev_out)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    *dev_out = dev;
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_partials[dev]) {
        if (hipMalloc(&g_partials[dev], (kMaxBlocks + 8) * sizeof(double)) != hipSuccess) g_partials[dev] = nullptr;
        // (the arrival counter of k_abs_sum starts at 0; zeroed ON THE LIBRARY STREAM: a NULL-stream memset is not ordered with a
        // non-blocking library stream, PDWT_STREAM_NONBLOCKING=1 / pdwt_set_stream)
        else if (hipMemsetAsync(g_partials[dev], 0, (kMaxBlocks + 8) * sizeof(double), stream()) != hipSuccess) {
            (void)hipFree(g_partials[dev]);
            g_partials[dev] = nullptr;
        }
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

// w_call_soft_thresh / w_call_hard_thresh / w_call_proj_linf / w_shrink, src/common.cu:219-315, 346-371: the same
// band walk with a different elementwise operator.  normalize: beta / sqrt(2) per level (soft, hard);
// the approximation band takes beta / sqrt(2)^nlevels (soft) -- or, for the HARD threshold, the un-normalised
// beta: the reference computes beta2 but passes beta (src/common.cu:262-270, SURVEY B-4), reproduced.
template <int OP, typename T>
static int ew_bands(T** c, T beta, pdwt_info w, int do_thresh_appcoeffs, int normalize, int kernel_id)
{
    if (!c) return PDWT_EINVAL;
    BandGeom g;
    if (band_geometry(w, &g) != PDWT_OK) return PDWT_EINVAL;
    BandTable<T> tab;
    tab.nb = 0;
    tab.chunk0[0] = 0;
    bool vec = true;
    const int per = (w.ndims == 2) ? 3 : 1;
    if (OP == OP_SCALE) beta = T(1) / (T(1) + beta);  // w_shrink: scal by 1/(1+beta), src/common.cu:355
    if (do_thresh_appcoeffs) {
        T beta2 = beta;
        if (normalize > 0 && OP == OP_SOFT) {  // beta / sqrt(2)^nlevels, src/common.cu:231-235
            const int nl2 = w.nlevels / 2;
            beta2 /= (T)(1 << nl2);
            if (nl2 * 2 != w.nlevels) beta2 = (T)(beta2 / 1.4142135623730951);
        }
        // the reference sweeps the whole level-1-sized allocation of band 0; only its first
        // Nr_L*Nc_L elements are coefficients, the rest is scratch -> process the coefficients only
        if (!table_push<T>(tab, c[0], (size_t)g.Nr[0] * g.Nc[0], beta2, vec)) return PDWT_EINVAL;
    }
    for (int lev = 0; lev < w.nlevels; lev++) {
        if (normalize > 0 && (OP == OP_SOFT || OP == OP_HARD)) beta = (T)(beta / 1.4142135623730951);  // src/common.cu:244,277
        for (int b = 0; b < per; b++) {
            const int k = per * lev + 1 + b;
            if (!table_push<T>(tab, c[k], (size_t)g.Nr[k] * g.Nc[k], beta, vec)) return PDWT_EINVAL;
        }
    }
    const unsigned int total = tab.chunk0[tab.nb];
    if (total == 0) return PDWT_OK;
    const int blocks = (int)(total < (unsigned)kMaxBlocks ? total : (unsigned)kMaxBlocks);
    KTimer kt(kernel_id);
    if (vec) hipLaunchKernelGGL((k_soft_thresh<T, true, OP>), dim3(blocks), dim3(kUThreads), 0, stream(), tab);
    else hipLaunchKernelGGL((k_soft_thresh<T, false, OP>), dim3(blocks), dim3(kUThreads), 0, stream(), tab);
    PDWT_CHECK_LAUNCH();
    return PDWT_OK;
}

// soft threshold + sum|c| of the result over all bands in `scratch` (device, pdwt_sum_scratch_doubles() doubles, any
// contents: reset_ticket); nothing is copied to the host and nothing synchronises
template <typename T>
static int soft_thresh_sum(T** c, T beta, pdwt_info w, int do_thresh_appcoeffs, int normalize, double* scratch)
{
    if (!c || !scratch) return PDWT_EINVAL;
    BandGeom g;
    if (band_geometry(w, &g) != PDWT_OK) return PDWT_EINVAL;
    BandTable<T> tab;
    tab.nb = 0;
    tab.chunk0[0] = 0;
    bool vec = true;
    const int per = (w.ndims == 2) ? 3 : 1;
    {
        T beta2 = T(-1);  // band 0: summed only, unless the approximation is thresholded too
        if (do_thresh_appcoeffs) {
            beta2 = beta;
            if (normalize > 0) {  // beta / sqrt(2)^nlevels, src/common.cu:231-235
                const int nl2 = w.nlevels / 2;
                beta2 /= (T)(1 << nl2);
                if (nl2 * 2 != w.nlevels) beta2 = (T)(beta2 / 1.4142135623730951);
            }
            if (beta2 < T(0)) beta2 = T(0);
        }
        if (!table_push<T>(tab, c[0], (size_t)g.Nr[0] * g.Nc[0], beta2, vec)) return PDWT_EINVAL;
    }
    if (beta < T(0)) return PDWT_EINVAL;  // (a negative threshold means "sum only" inside the kernel)
    for (int lev = 0; lev < w.nlevels; lev++) {
        if (normalize > 0) beta = (T)(beta / 1.4142135623730951);  // src/common.cu:244
        for (int b = 0; b < per; b++) {
            const int k = per * lev + 1 + b;
            if (!table_push<T>(tab, c[k], (size_t)g.Nr[k] * g.Nc[k], beta, vec)) return PDWT_EINVAL;
        }
    }
    const unsigned int total = tab.chunk0[tab.nb];
    const int blocks = (int)(total < (unsigned)kMaxBlocks ? (total ? total : 1) : (unsigned)kMaxBlocks);
    if (const int rc = reset_ticket(scratch); rc != PDWT_OK) return rc;
    KTimer kt(K_THRESH_SUM);
    if (vec) hipLaunchKernelGGL((k_soft_thresh_sum<T, true>), dim3(blocks), dim3(kUThreads), 0, stream(), tab, scratch);
    else hipLaunchKernelGGL((k_soft_thresh_sum<T, false>), dim3(blocks), dim3(kUThreads), 0, stream(), tab, scratch);
    PDWT_CHECK_LAUNCH();
    return PDWT_OK;
}

// w_call_group_soft_thresh, src/common.cu:318-343
template <typename T>
static int group_soft_thresh(T** c, T beta, pdwt_info w, int do_thresh_appcoeffs, int normalize)
{
    if (!c) return PDWT_EINVAL;
    BandGeom g;
    if (band_geometry(w, &g) != PDWT_OK || w.nlevels > 32) return PDWT_EINVAL;
    GroupTable<T> tab;
    tab.nb = 0;
    tab.chunk0[0] = 0;
    const int per = (w.ndims == 2) ? 3 : 1;
    for (int lev = 0; lev < w.nlevels; lev++) {
        if (normalize > 0) beta = (T)(beta / 1.4142135623730951);
        const int k0 = per * lev + 1;
        const size_t n = (size_t)g.Nr[k0] * g.Nc[k0];
        const int k = tab.nb++;
        tab.h[k] = (per == 3) ? c[k0] : nullptr;
        tab.v[k] = (per == 3) ? c[k0 + 1] : nullptr;
        tab.d[k] = (per == 3) ? c[k0 + 2] : c[k0];
        // the approximation joins the group at the last scale only, where it has the size of the details
        tab.a[k] = (do_thresh_appcoeffs && lev == w.nlevels - 1) ? c[0] : nullptr;
        tab.n[k] = n;
        tab.beta[k] = beta;
        tab.chunk0[k + 1] = tab.chunk0[k] + (unsigned int)((n + kChunk - 1) / kChunk);
    }
    const unsigned int total = tab.chunk0[tab.nb];
    if (total == 0) return PDWT_OK;
    const int blocks = (int)(total < (unsigned)kMaxBlocks ? total : (unsigned)kMaxBlocks);
    KTimer kt(K_SOFT_THRESH);
    hipLaunchKernelGGL(k_group_soft_thresh<T>, dim3(blocks), dim3(kUThreads), 0, stream(), tab);
    PDWT_CHECK_LAUNCH();
    return PDWT_OK;
}

// sum over bands of |c| (mode 0) or c^2 (mode 1), in double.
//   norm1   Wavelets::norm1, src/wt.cu:398-418: |c| over all bands including band 0
//   norm2sq Wavelets::norm2sq, src/wt.cu:370-395: c^2 over all bands.  The reference's 1-D branch adds cublas_asum
//           (sum |c|) of the detail bands (src/wt.cu:389, SURVEY B-4): a bug, FIXED here (the squared l2 norm is
//           returned); knob norm2sq_ref1d = 1 reproduces the reference value.
template <typename T>
static int band_sum_double(T** c, pdwt_info w, double* out, int squares, int ref_quirk_1d, double* scratch = nullptr)
{
    // scratch != NULL: enqueue only -- the partials and the result land in the CALLER's device buffer (pdwt_sum_scratch_doubles()
    // doubles; read later with pdwt_sum_scratch_read), nothing synchronises.  That is how the shards of a batch overlap their
    // reductions (include/wt_batch.h).
    if (!c || (!out && !scratch)) return PDWT_EINVAL;
    BandGeom g;
    if (band_geometry(w, &g) != PDWT_OK) return PDWT_EINVAL;
    BandTable<T> tab;
    tab.nb = 0;
    tab.chunk0[0] = 0;
    bool vec = true;
    for (int k = 0; k < g.nbands; k++) {
        const bool sq = squares && !(ref_quirk_1d && knob(KN_NORM2SQ_REF1D) == 1 && w.ndims == 1 && k > 0);
        if (!table_push<T>(tab, c[k], (size_t)g.Nr[k] * g.Nc[k], sq ? T(1) : T(0), vec)) return PDWT_EINVAL;
    }
    int dev = 0;
    double* part = scratch ? scratch : partials(&dev);
    if (!part) return PDWT_ENOMEM;
    std::unique_lock<std::mutex> red_lock(g_red_mu[scratch ? 0 : dev], std::defer_lock);
    if (!scratch) red_lock.lock();  // the per-device partials are shared by every instance on that device
    const unsigned int total = tab.chunk0[tab.nb];
    const int blocks = (int)(total < (unsigned)kMaxBlocks ? (total ? total : 1) : (unsigned)kMaxBlocks);
    if (scratch) {
        if (const int rc = reset_ticket(scratch); rc != PDWT_OK) return rc;
    }
    {
        KTimer kt(K_ABS_SUM);
        if (vec) hipLaunchKernelGGL((k_abs_sum<T, true>), dim3(blocks), dim3(kUThreads), 0, stream(), tab, part);
        else hipLaunchKernelGGL((k_abs_sum<T, false>), dim3(blocks), dim3(kUThreads), 0, stream(), tab, part);
        PDWT_CHECK_LAUNCH();
    }
    if (scratch) return PDWT_OK;
    return pdwt_memcpy_d2h(out, part + kMaxBlocks, sizeof(double));
}

// w_add_coeffs / w_add_coeffs_1d, src/common.cu:499-526: dst += alpha*src on every band.  (The reference's 1-D
// variant sizes the bands with Nc/2 instead of the ceil-half rule, i.e. it skips part of each band for odd
// sizes, SURVEY B-4: fixed here, the whole band is added.)
template <typename T>
static int add_coeffs(T** dst, T** src, pdwt_info w, T alpha)
{
    if (!dst || !src) return PDWT_EINVAL;
    BandGeom g;
    if (band_geometry(w, &g) != PDWT_OK) return PDWT_EINVAL;
    PairTable<T> tab;
    tab.nb = 0;
    tab.chunk0[0] = 0;
    for (int k = 0; k < g.nbands; k++) {
        if (!dst[k] || !src[k]) return PDWT_EINVAL;
        const size_t n = (size_t)g.Nr[k] * g.Nc[k];
        tab.dst[k] = dst[k];
        tab.src[k] = src[k];
        tab.n[k] = n;
        tab.chunk0[k + 1] = tab.chunk0[k] + (unsigned int)((n + kChunk - 1) / kChunk);
        tab.nb++;
    }
    const unsigned int total = tab.chunk0[tab.nb];
    if (total == 0) return PDWT_OK;
    const int blocks = (int)(total < (unsigned)kMaxBlocks ? total : (unsigned)kMaxBlocks);
    KTimer kt(K_SOFT_THRESH);
    hipLaunchKernelGGL(k_axpy_bands<T>, dim3(blocks), dim3(kUThreads), 0, stream(), tab, alpha);
    PDWT_CHECK_LAUNCH();
    return PDWT_OK;
}

// w_call_circshift, src/common.cu:378-396.  inplace: result in d_image (through a copy in d_image2), else in d_image2.
template <typename T>
static int circshift(T* d_image, T* d_image2, pdwt_info w, int sr, int sc, int inplace)
{
    if (!d_image || !d_image2 || w.Nr < 1 || w.Nc < 1) return PDWT_EINVAL;
    const int Nr = w.Nr, Nc = w.Nc;
    if (sr < 0) sr += Nr;  // (the reference normalises the same way: one add, then %; src/common.cu:381-385)
    if (sc < 0) sc += Nc;
    sr %= Nr;
    sc %= Nc;
    if (sr < 0) sr += Nr;
    if (sc < 0) sc += Nc;
    if (w.ndims == 1) sr = 0;
    const T* in = d_image;
    T* out = d_image2;
    if (inplace) {
        const int rc = pdwt_memcpy_d2d(d_image2, d_image, (size_t)Nr * Nc * sizeof(T));
        if (rc != PDWT_OK) return rc;
        in = d_image2;
        out = d_image;
    }
    dim3 grid(idiv_up(Nc, kUThreads), Nr < 1024 ? Nr : 1024);
    KTimer kt(K_SOFT_THRESH);
    hipLaunchKernelGGL(k_circshift<T>, grid, dim3(kUThreads), 0, stream(), in, out, Nr, Nc, sr, sc);
    PDWT_CHECK_LAUNCH();
    return PDWT_OK;
}

}  // namespace pdwt

using namespace pdwt;

#define PDWT_UTILS_API(SFX, T)                                                                                                       \
    int pdwt_soft_thresh_##SFX(T** c, T beta, pdwt_info w, int app, int norm) { return ew_bands<OP_SOFT, T>(c, beta, w, app, norm, K_SOFT_THRESH); } \
    int pdwt_hard_thresh_##SFX(T** c, T beta, pdwt_info w, int app, int norm) { return ew_bands<OP_HARD, T>(c, beta, w, app, norm, K_SOFT_THRESH); } \
    int pdwt_proj_linf_##SFX(T** c, T beta, pdwt_info w, int app) { return ew_bands<OP_PROJ, T>(c, beta, w, app, 0, K_SOFT_THRESH); }         \
    int pdwt_shrink_##SFX(T** c, T beta, pdwt_info w, int app) { return ew_bands<OP_SCALE, T>(c, beta, w, app, 0, K_SOFT_THRESH); }           \
    int pdwt_group_soft_thresh_##SFX(T** c, T beta, pdwt_info w, int app, int norm) { return group_soft_thresh<T>(c, beta, w, app, norm); }   \
    int pdwt_norm1_as_double_##SFX(T** c, pdwt_info w, double* out) { return band_sum_double<T>(c, w, out, 0, 0); }                          \
    int pdwt_norm1_enqueue_##SFX(T** c, pdwt_info w, double* scratch) { return band_sum_double<T>(c, w, nullptr, 0, 0, scratch); }                          \
    int pdwt_norm2sq_as_double_##SFX(T** c, pdwt_info w, double* out) { return band_sum_double<T>(c, w, out, 1, 1); }                        \
    int pdwt_norm1_##SFX(T** c, pdwt_info w, T* out)                                                                                         \
    {                                                                                                                                        \
        double d = 0;                                                                                                                        \
        if (!out) return PDWT_EINVAL;                                                                                                        \
        const int rc = band_sum_double<T>(c, w, &d, 0, 0);                                                                                   \
        if (rc == PDWT_OK) *out = (T)d;                                                                                                      \
        return rc;                                                                                                                           \
    }                                                                                                                                        \
    int pdwt_norm2sq_##SFX(T** c, pdwt_info w, T* out)                                                                                       \
    {                                                                                                                                        \
        double d = 0;                                                                                                                        \
        if (!out) return PDWT_EINVAL;                                                                                                        \
        const int rc = band_sum_double<T>(c, w, &d, 1, 1);                                                                                   \
        if (rc == PDWT_OK) *out = (T)d;                                                                                                      \
        return rc;                                                                                                                           \
    }                                                                                                                                        \
    int pdwt_soft_thresh_sum_##SFX(T** c, T beta, pdwt_info w, int app, int norm, double* scratch)                                         \
    {                                                                                                                                        \
        return soft_thresh_sum<T>(c, beta, w, app, norm, scratch);                                                                           \
    }                                                                                                                                        \
    int pdwt_add_coeffs_##SFX(T** dst, T** src, pdwt_info w, T alpha) { return add_coeffs<T>(dst, src, w, alpha); }                         \
    int pdwt_circshift_##SFX(T* img, T* img2, pdwt_info w, int sr, int sc, int inplace) { return circshift<T>(img, img2, w, sr, sc, inplace); }

extern "C" {
size_t pdwt_sum_scratch_doubles(void) { return (size_t)kMaxBlocks + 8; }
size_t pdwt_sum_result_index(void) { return (size_t)kMaxBlocks; }
size_t pdwt_sum_spare_index(void) { return (size_t)kMaxBlocks + 2; }
int pdwt_sum_scratch_read(const double* scratch, double* out)
{
    if (!scratch || !out) return PDWT_EINVAL;
    return pdwt_memcpy_d2h(out, scratch + kMaxBlocks, sizeof(double));
}
PDWT_UTILS_API(f32, float)
PDWT_UTILS_API(f64, double)
}
