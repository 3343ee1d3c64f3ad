// swt.hip -- stationary (undecimated, a-trous) separable transform: kernels + level drivers.
//
// Path replaced: reference src/separable.cu:409-672 (w_kern_forward_swt_pass1/2,
// w_kern_inverse_swt_pass1/2 and their four drivers).  Math: SURVEY.md Appendix A-3 / A-4.
// Every level works at full resolution with tap spacing f = 2^(level-1); a fused row+column tile
// would need an (hlen-1)*f halo in BOTH directions (208 samples at level 5 of db7), so each level is
// two streaming passes whose dilated taps are served by L1/L2: x stays the lane axis, so every tap
// is a fully coalesced 256-byte row segment; the column pass never strides a wave across rows.
#include <type_traits>

#include "cols_ring.hpp"
#include <new>
#include <vector>

#include "common.hpp"
#include "swt_fused.hpp"

namespace pdwt {

constexpr int kSwtThreads = 256;  // 64 columns x 4 rows
constexpr int kSwtRows = 4;       // output rows per thread

// rows: lo/hi[y][g] = sum_j in[y][(g - c + f*j) mod Nc] * L/H[hlen-1-j],  c = (hlen/2-1)*f  (A-3)
template <typename T>
__global__ __launch_bounds__(kSwtThreads) void k_swt_ana_rows(const T* __restrict__ in, T* __restrict__ lo, T* __restrict__ hi, int Nr,
                                                               int Nc, int hlen, int fct, Taps2<T> f)
{
    const int gx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int gy0 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * kSwtRows;
    if (gx >= Nc) return;
    const int c = ((hlen & 1) ? hlen / 2 : hlen / 2 - 1) * fct;
    for (int r = 0; r < kSwtRows; r++) {
        const int gy = gy0 + r;
        if (gy >= Nr) return;
        const T* row = in + (size_t)gy * Nc;
        T l = 0, h = 0;
        int src = gx - c;
        for (int j = 0; j < hlen; j++, src += fct) {
            const T v = row[wrap_per(src, Nc)];
            l = fma_t(v, f.a[hlen - 1 - j], l);
            h = fma_t(v, f.b[hlen - 1 - j], h);
        }
        lo[(size_t)gy * Nc + gx] = l;
        hi[(size_t)gy * Nc + gx] = h;
    }
}

// cols: A,H from t1 and V,D from t2 (reference pass2, src/separable.cu:452-493)
template <typename T>
__global__ __launch_bounds__(kSwtThreads) void k_swt_ana_cols(const T* __restrict__ t1, const T* __restrict__ t2, T* __restrict__ cA,
                                                               T* __restrict__ cH, T* __restrict__ cV, T* __restrict__ cD, int Nr, int Nc,
                                                               int hlen, int fct, Taps2<T> f)
{
    const int gx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int gy0 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * kSwtRows;
    if (gx >= Nc) return;
    const int c = ((hlen & 1) ? hlen / 2 : hlen / 2 - 1) * fct;
    for (int r = 0; r < kSwtRows; r++) {
        const int gy = gy0 + r;
        if (gy >= Nr) return;
        T a = 0, h = 0, v = 0, d = 0;
        int src = gy - c;
        for (int j = 0; j < hlen; j++, src += fct) {
            const size_t o = (size_t)wrap_per(src, Nr) * Nc + gx;
            const T l = t1[o], g = t2[o];
            const T fl = f.a[hlen - 1 - j], fh = f.b[hlen - 1 - j];
            a = fma_t(l, fl, a);
            h = fma_t(l, fh, h);
            v = fma_t(g, fl, v);
            d = fma_t(g, fh, d);
        }
        const size_t o = (size_t)gy * Nc + gx;
        cA[o] = a;
        cH[o] = h;
        cV[o] = v;
        cD[o] = d;
    }
}

// inverse cols: t1 = A*IL/2 + H*IH/2, t2 = V*IL/2 + D*IH/2,  c = (hlen/2)*f  (A-4; taps pre-halved)
template <typename T>
__global__ __launch_bounds__(kSwtThreads) void k_swt_syn_cols(const T* __restrict__ cA, const T* __restrict__ cH, const T* __restrict__ cV,
                                                               const T* __restrict__ cD, T* __restrict__ t1, T* __restrict__ t2, int Nr,
                                                               int Nc, int hlen, int fct, Taps2<T> f)
{
    const int gx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int gy0 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * kSwtRows;
    if (gx >= Nc) return;
    const int c = (hlen / 2) * fct;
    for (int r = 0; r < kSwtRows; r++) {
        const int gy = gy0 + r;
        if (gy >= Nr) return;
        T sa = 0, sh = 0, sv = 0, sd = 0;
        int src = gy - c;
        for (int j = 0; j < hlen; j++, src += fct) {
            const size_t o = (size_t)wrap_per(src, Nr) * Nc + gx;
            const T fl = f.a[hlen - 1 - j], fh = f.b[hlen - 1 - j];
            sa = fma_t(cA[o], fl, sa);
            sh = fma_t(cH[o], fh, sh);
            sv = fma_t(cV[o], fl, sv);
            sd = fma_t(cD[o], fh, sd);
        }
        t1[(size_t)gy * Nc + gx] = sa + sh;
        t2[(size_t)gy * Nc + gx] = sv + sd;
    }
}

// inverse rows: out = a*IL/2 + d*IH/2
template <typename T>
__global__ __launch_bounds__(kSwtThreads) void k_swt_syn_rows(const T* __restrict__ a, const T* __restrict__ d, T* __restrict__ out, int Nr,
                                                               int Nc, int hlen, int fct, Taps2<T> f)
{
    const int gx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int gy0 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * kSwtRows;
    if (gx >= Nc) return;
    const int c = (hlen / 2) * fct;
    for (int r = 0; r < kSwtRows; r++) {
        const int gy = gy0 + r;
        if (gy >= Nr) return;
        const T* pa = a + (size_t)gy * Nc;
        const T* pd = d + (size_t)gy * Nc;
        T s1 = 0, s2 = 0;
        int src = gx - c;
        for (int j = 0; j < hlen; j++, src += fct) {
            const int sx = wrap_per(src, Nc);
            s1 = fma_t(pa[sx], f.a[hlen - 1 - j], s1);
            s2 = fma_t(pd[sx], f.b[hlen - 1 - j], s2);
        }
        out[(size_t)gy * Nc + gx] = s1 + s2;
    }
}

// -------------------------------------------------------------------------------------------------
// Row passes with the whole row resident in LDS (explicit periodic halo), persistent workgroups.
// The direct kernels above read every dilated tap from global memory (hlen L1/L2 round trips per output);
// here a row is staged once with 16-byte loads and each tap of a 4-output group is ONE aligned ds_read_b128
// (tap spacing f >= 4 keeps x - c*f + f*j 16-byte aligned; lanes are 16 bytes apart: conflict-free).
// Levels with f < 4 use one output per lane-iteration (conflict-free 4-byte reads).
//   SYN = false : in -> lo, hi                 (src/separable.cu:409-448, centre (hlen/2-1)*f)
//   SYN = true  : a, d -> out = a*IL/2 + d*IH/2 (src/separable.cu:593-626, centre (hlen/2)*f; taps pre-halved)
// -------------------------------------------------------------------------------------------------
template <int I, int N, typename F>
__device__ __forceinline__ void sw_for_impl(F&& fn)
{
    if constexpr (I < N) {
        fn(std::integral_constant<int, I>{});
        sw_for_impl<I + 1, N>(fn);
    }
}
template <int N, typename F>
__device__ __forceinline__ void sw_for(F&& fn) { sw_for_impl<0, N>(fn); }

__device__ __forceinline__ void swt_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <typename T> struct SwV;
template <> struct SwV<float> { typedef float type __attribute__((ext_vector_type(4))); static constexpr int N = 4; };
template <> struct SwV<double> { typedef double type __attribute__((ext_vector_type(2))); static constexpr int N = 2; };

template <typename T, int HLEN, bool SYN>
__global__ __launch_bounds__(256) void k_swt_rows_lds(const T* __restrict__ a, const T* __restrict__ d, T* __restrict__ o1, T* __restrict__ o2,
                                                       int Nr, int Nc, int fct, int HL, int HR, Taps2<T> f)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using V = typename SwV<T>::type;
    constexpr int NV = SwV<T>::N;
    constexpr int C = SYN ? HLEN / 2 : HLEN / 2 - 1;
    const int co = C * fct;
    const int pitch = HL + Nc + HR;  // elements; HL, Nc, HR are multiples of NV
    T* ra = reinterpret_cast<T*>(smem) + HL;
    T* rd = ra + pitch;  // SYN only
    const int nchunks = Nc / NV;
    for (int row = blockIdx.x; row < Nr; row += gridDim.x) {
        const V* ga = reinterpret_cast<const V*>(a + (size_t)row * Nc);
        const V* gd = reinterpret_cast<const V*>(SYN ? d + (size_t)row * Nc : a);
        for (int i = threadIdx.x; i < nchunks; i += 256) {
            reinterpret_cast<V*>(ra)[i] = ga[i];
            if (SYN) reinterpret_cast<V*>(rd)[i] = gd[i];
        }
        swt_lds_barrier();
        for (int k = threadIdx.x; k < HL + HR; k += 256) {
            const int idx = k < HL ? k - HL : Nc + (k - HL);
            const int src = wrap_per(idx, Nc);
            ra[idx] = ra[src];
            if (SYN) rd[idx] = rd[src];
        }
        swt_lds_barrier();
        if (fct >= NV) {
            for (int g = threadIdx.x; g < nchunks; g += 256) {
                const T* pa = ra + g * NV - co;
                const T* pd = rd + g * NV - co;
                T s1[NV], s2[NV];
#pragma unroll
                for (int q = 0; q < NV; q++) s1[q] = s2[q] = T(0);
                sw_for<HLEN>([&](auto J) {
                    constexpr int j = decltype(J)::value;
                    const T fa = f.a[HLEN - 1 - j], fb = f.b[HLEN - 1 - j];
                    const V va = *reinterpret_cast<const V*>(pa + fct * j);
                    if (SYN) {
                        const V vd = *reinterpret_cast<const V*>(pd + fct * j);
#pragma unroll
                        for (int q = 0; q < NV; q++) {
                            s1[q] = fma_t(va[q], fa, s1[q]);
                            s2[q] = fma_t(vd[q], fb, s2[q]);
                        }
                    } else {
#pragma unroll
                        for (int q = 0; q < NV; q++) {
                            s1[q] = fma_t(va[q], fa, s1[q]);
                            s2[q] = fma_t(va[q], fb, s2[q]);
                        }
                    }
                });
                V r1, r2;
#pragma unroll
                for (int q = 0; q < NV; q++) {
                    r1[q] = SYN ? s1[q] + s2[q] : s1[q];
                    r2[q] = s2[q];
                }
                reinterpret_cast<V*>(o1 + (size_t)row * Nc)[g] = r1;
                if (!SYN) reinterpret_cast<V*>(o2 + (size_t)row * Nc)[g] = r2;
            }
        } else {
            for (int x = threadIdx.x; x < Nc; x += 256) {
                const T* pa = ra + x - co;
                const T* pd = rd + x - co;
                T s1 = 0, s2 = 0;
                sw_for<HLEN>([&](auto J) {
                    constexpr int j = decltype(J)::value;
                    const T va = pa[fct * j];
                    s1 = fma_t(va, f.a[HLEN - 1 - j], s1);
                    s2 = fma_t(SYN ? pd[fct * j] : va, f.b[HLEN - 1 - j], s2);
                });
                o1[(size_t)row * Nc + x] = SYN ? s1 + s2 : s1;
                if (!SYN) o2[(size_t)row * Nc + x] = s2;
            }
        }
        swt_lds_barrier();  // the row buffers are rewritten by the next iteration
    }
}

#define PDWT_SWT_ROWS_HLENS(X) X(2) X(4) X(6) X(8) X(10) X(12) X(14) X(16) X(18) X(20)

// returns PDWT_OK when launched, 1 when the geometry is outside this path
template <typename T, bool SYN>
static int swt_rows_lds(const T* a, const T* d, T* o1, T* o2, int Nr, int Nc, int hlen, int fct, const Taps2<T>& f, int ktimer_id)
{
    constexpr int NV = SwV<T>::N;
    if (Nc % NV != 0 || (((uintptr_t)a | (uintptr_t)d | (uintptr_t)o1 | (uintptr_t)o2) & 15) != 0) return 1;
    const int C = SYN ? hlen / 2 : hlen / 2 - 1;
    auto up = [](int v) { return ((v + NV - 1) / NV) * NV; };
    const int HL = up(C * fct), HR = up((hlen - 1 - C) * fct + NV);
    if (HL > Nc || HR > Nc) return 1;
    const size_t lds = (size_t)(SYN ? 2 : 1) * (HL + Nc + HR) * sizeof(T);
    if (lds > 64 * 1024) return 1;
    void (*k)(const T*, const T*, T*, T*, int, int, int, int, int, Taps2<T>) = nullptr;
    switch (hlen) {
#define X(H) case H: k = k_swt_rows_lds<T, H, SYN>; break;
        PDWT_SWT_ROWS_HLENS(X)
#undef X
        default: return 1;
    }
    const int per_cu = (int)((160 * 1024) / (lds + 256)) > 8 ? 8 : (int)((160 * 1024) / (lds + 256));
    const int grid = Nr < 256 * per_cu ? Nr : 256 * per_cu;
    KTimer kt(ktimer_id);
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, stream(), a, d, o1, o2, Nr, Nc, fct, HL, HR, f);
    PDWT_HIP_TRY(hipGetLastError());
    return PDWT_OK;
}

#define PDWT_CHECK_LAUNCH() PDWT_HIP_TRY(hipGetLastError())

static dim3 swt_grid(int Nr, int Nc) { return dim3(idiv_up(Nc, 64), idiv_up(Nr, 4 * kSwtRows)); }

static int check(const void* img, const void* c, const void* tmp, const pdwt_info& w, int ndims, const void* f, int fhlen)
{
    if (!img || !c || !tmp || !f) return PDWT_EINVAL;
    if (w.Nr < 1 || w.Nc < 1 || w.nlevels < 1 || w.nlevels > 30 || w.ndims != ndims) return PDWT_EINVAL;
    if (w.hlen < 2 || w.hlen > PDWT_MAX_FILTER_WIDTH || fhlen != w.hlen) return PDWT_EINVAL;
    return PDWT_OK;
}

// w_forward_swt_separable, src/separable.cu:496-516
template <typename T>
static int forward_swt(T* d_image, T** c, T* d_tmp, pdwt_info w, const typename FiltersOf<T>::type* filt)
{
    int rc = check(d_image, c, d_tmp, w, 2, filt, filt ? filt->hlen : 0);
    if (rc != PDWT_OK) return rc;
    const Taps2<T> f = taps_fwd<T>(filt);
    T* t1 = d_tmp;
    T* t2 = d_tmp + (size_t)w.Nr * w.Nc;
    const T* in = d_image;
    const dim3 grid = swt_grid(w.Nr, w.Nc);
    for (int lev = 0; lev < w.nlevels; lev++) {
        {
            // row pass + column pass in one launch (swt_fused.inc, swt_fused_f64.inc).  The approximation ping-pongs between the
            // two halves of d_tmp (free on this path) because a single launch cannot read band 0 while it overwrites it.
            T* aout = (lev == w.nlevels - 1) ? c[0] : ((lev & 1) ? t2 : t1);
            int rr;
            if constexpr (sizeof(T) == 4) rr = swt_fwd_fused_f32(in, aout, c[3 * lev + 1], c[3 * lev + 2], c[3 * lev + 3], w.Nr, w.Nc, w.hlen, 1 << lev, f);
            else rr = swt_fwd_fused_f64(in, aout, c[3 * lev + 1], c[3 * lev + 2], c[3 * lev + 3], w.Nr, w.Nc, w.hlen, 1 << lev, f);
            if (rr < 0) return rr;
            if (rr == PDWT_OK) {
                in = aout;
                continue;
            }
            if (in != d_image && in != c[0]) {  // fell off the fused path mid-way: the two-pass kernels expect the approximation in band 0
                rc = pdwt_memcpy_d2d(c[0], in, (size_t)w.Nr * w.Nc * sizeof(T));
                if (rc != PDWT_OK) return rc;
                in = c[0];
            }
        }
        {
            int rr = swt_rows_lds<T, false>(in, in, t1, t2, w.Nr, w.Nc, w.hlen, 1 << lev, f, K_SWT_ANA_ROWS);
            if (rr < 0) return rr;
            if (rr > 0) {
                KTimer kt(K_SWT_ANA_ROWS);
                hipLaunchKernelGGL(k_swt_ana_rows<T>, grid, dim3(kSwtThreads), 0, stream(), in, t1, t2, w.Nr, w.Nc, w.hlen, 1 << lev, f);
                PDWT_CHECK_LAUNCH();
            }
        }
        {
            KTimer kt(K_SWT_ANA_COLS);
            // register-ring form per residue class (cols_ring.inc); direct dilated-tap kernel when f does not divide Nr
            int rr = swt_ana_cols_ring<T>(t1, c[0], c[3 * lev + 1], w.Nr, w.Nc, w.hlen, 1 << lev, f);
            if (rr == PDWT_OK) rr = swt_ana_cols_ring<T>(t2, c[3 * lev + 2], c[3 * lev + 3], w.Nr, w.Nc, w.hlen, 1 << lev, f);
            if (rr < 0) return rr;
            if (rr > 0) {
                hipLaunchKernelGGL(k_swt_ana_cols<T>, grid, dim3(kSwtThreads), 0, stream(), (const T*)t1, (const T*)t2, c[0], c[3 * lev + 1],
                                   c[3 * lev + 2], c[3 * lev + 3], w.Nr, w.Nc, w.hlen, 1 << lev, f);
                PDWT_CHECK_LAUNCH();
            }
        }
        in = c[0];  // stream order makes the read-then-overwrite of band 0 safe (two separate launches)
    }
    return PDWT_OK;
}

// w_inverse_swt_separable, src/separable.cu:629-650
template <typename T>
static int inverse_swt(T* d_image, T** c, T* d_tmp, pdwt_info w, const typename FiltersOf<T>::type* filt)
{
    int rc = check(d_image, c, d_tmp, w, 2, filt, filt ? filt->hlen : 0);
    if (rc != PDWT_OK) return rc;
    const Taps2<T> f = taps_inv<T>(filt, T(0.5));
    T* t1 = d_tmp;
    T* t2 = d_tmp + (size_t)w.Nr * w.Nc;
    const dim3 grid = swt_grid(w.Nr, w.Nc);
    const T* a = c[0];  // approximation feeding level i (band 0, or a half of d_tmp after a fused level)
    for (int i = w.nlevels - 1; i >= 0; i--) {
        {
            // row + column synthesis in one launch (swt_fused.inc, swt_fused_f64.inc); the approximation ping-pongs through d_tmp
            T* out = (i == 0) ? d_image : ((i & 1) ? t2 : t1);
            int rr;
            if constexpr (sizeof(T) == 4) rr = swt_inv_fused_f32(a, c[3 * i + 1], c[3 * i + 2], c[3 * i + 3], out, w.Nr, w.Nc, w.hlen, 1 << i, f);
            else rr = swt_inv_fused_f64(a, c[3 * i + 1], c[3 * i + 2], c[3 * i + 3], out, w.Nr, w.Nc, w.hlen, 1 << i, f);
            if (rr < 0) return rr;
            if (rr == PDWT_OK) {
                a = out;
                continue;
            }
            if (a != c[0]) {  // fell off the fused path mid-way: the two-pass kernels read the approximation from band 0
                rc = pdwt_memcpy_d2d(c[0], a, (size_t)w.Nr * w.Nc * sizeof(T));
                if (rc != PDWT_OK) return rc;
                a = c[0];
            }
        }
        {
            KTimer kt(K_SWT_SYN_COLS);
            int rr = swt_syn_cols_ring<T>(c[0], c[3 * i + 1], t1, w.Nr, w.Nc, w.hlen, 1 << i, f);
            if (rr == PDWT_OK) rr = swt_syn_cols_ring<T>(c[3 * i + 2], c[3 * i + 3], t2, w.Nr, w.Nc, w.hlen, 1 << i, f);
            if (rr < 0) return rr;
            if (rr > 0) {
                hipLaunchKernelGGL(k_swt_syn_cols<T>, grid, dim3(kSwtThreads), 0, stream(), (const T*)c[0], (const T*)c[3 * i + 1],
                                   (const T*)c[3 * i + 2], (const T*)c[3 * i + 3], t1, t2, w.Nr, w.Nc, w.hlen, 1 << i, f);
                PDWT_CHECK_LAUNCH();
            }
        }
        {
            T* out = (i == 0) ? d_image : c[0];
            int rr = swt_rows_lds<T, true>(t1, t2, out, out, w.Nr, w.Nc, w.hlen, 1 << i, f, K_SWT_SYN_ROWS);
            if (rr < 0) return rr;
            if (rr > 0) {
                KTimer kt(K_SWT_SYN_ROWS);
                hipLaunchKernelGGL(k_swt_syn_rows<T>, grid, dim3(kSwtThreads), 0, stream(), (const T*)t1, (const T*)t2, out, w.Nr, w.Nc, w.hlen,
                                   1 << i, f);
                PDWT_CHECK_LAUNCH();
            }
        }
    }
    return PDWT_OK;
}

// w_forward_swt_separable_1d, src/separable.cu:520-537 (approximation ping-pongs in d_tmp, lands in band 0)
template <typename T>
static int forward_swt_1d(T* d_image, T** c, T* d_tmp, pdwt_info w, const typename FiltersOf<T>::type* filt)
{
    int rc = check(d_image, c, d_tmp, w, 1, filt, filt ? filt->hlen : 0);
    if (rc != PDWT_OK) return rc;
    const Taps2<T> f = taps_fwd<T>(filt);
    T* ping[2] = {d_tmp, d_tmp + (size_t)w.Nr * w.Nc};
    const T* in = d_image;
    const dim3 grid = swt_grid(w.Nr, w.Nc);
    for (int lev = 0; lev < w.nlevels; lev++) {
        T* aout = (lev == w.nlevels - 1) ? c[0] : ping[lev & 1];
        int rr = swt_rows_lds<T, false>(in, in, aout, c[lev + 1], w.Nr, w.Nc, w.hlen, 1 << lev, f, K_SWT_ANA_ROWS);
        if (rr < 0) return rr;
        if (rr > 0) {
            KTimer kt(K_SWT_ANA_ROWS);
            hipLaunchKernelGGL(k_swt_ana_rows<T>, grid, dim3(kSwtThreads), 0, stream(), in, aout, c[lev + 1], w.Nr, w.Nc, w.hlen, 1 << lev, f);
            PDWT_CHECK_LAUNCH();
        }
        in = aout;
    }
    return PDWT_OK;
}

// w_inverse_swt_separable_1d, src/separable.cu:654-672
template <typename T>
static int inverse_swt_1d(T* d_image, T** c, T* d_tmp, pdwt_info w, const typename FiltersOf<T>::type* filt)
{
    int rc = check(d_image, c, d_tmp, w, 1, filt, filt ? filt->hlen : 0);
    if (rc != PDWT_OK) return rc;
    const Taps2<T> f = taps_inv<T>(filt, T(0.5));
    T* ping[2] = {d_tmp, d_tmp + (size_t)w.Nr * w.Nc};
    const T* a = c[0];
    const dim3 grid = swt_grid(w.Nr, w.Nc);
    for (int i = w.nlevels - 1; i >= 0; i--) {
        T* out = (i == 0) ? d_image : ping[i & 1];
        int rr = swt_rows_lds<T, true>(a, (const T*)c[i + 1], out, out, w.Nr, w.Nc, w.hlen, 1 << i, f, K_SWT_SYN_ROWS);
        if (rr < 0) return rr;
        if (rr > 0) {
            KTimer kt(K_SWT_SYN_ROWS);
            hipLaunchKernelGGL(k_swt_syn_rows<T>, grid, dim3(kSwtThreads), 0, stream(), a, (const T*)c[i + 1], out, w.Nr, w.Nc, w.hlen, 1 << i, f);
            PDWT_CHECK_LAUNCH();
        }
        a = out;
    }
    return PDWT_OK;
}

// -------------------------------------------------------------------------------------------------
// a batch of equally sized float32 images through the fused SWT level kernels (pdwt_batch2d_* on instances created with do_swt = 1,
// dwt.hip): every level of ALL images in one launch (gridDim.z = image), the per-image pointers of forward_swt / inverse_swt -- the
// approximation ping-pongs through the two halves of each image's d_tmp -- in device-side tables built once.  NULL when a level of the
// geometry is outside the fused kernels (the caller then transforms image after image).
// -------------------------------------------------------------------------------------------------
struct SwtBatch {
    int nimg, L, Nr, Nc, hlen;
    unsigned long long* d_fwd;  // [level][image][5]: in, cA, cH, cV, cD
    unsigned long long* d_inv;  // [level][image][5]: cA, cH, cV, cD, out
};
static bool swt_batch_level_ok(int Nr, int Nc, int hlen, int fct)
{
    // the geometry rules of swt_fwd_fused_f32 / swt_inv_fused_f32 and of their launchers (swt_fused.inc, swt_fused_l2.inc)
    if ((hlen & 1) || hlen < 2 || hlen > 40) return false;
    const bool l2 = hlen > 20;
    const int H = l2 ? (hlen <= 24 ? 24 : (hlen <= 32 ? 32 : 40)) : hlen, tile = l2 ? 512 : 1024;
    if ((Nc & 3) || Nc < 64 || (Nr % fct) != 0 || Nr / fct < 2 * H) return false;
    for (int inv = 0; inv < 2; inv++) {
        const int C = inv ? H / 2 : H / 2 - 1;
        const int HLc = ((C * fct + 3) >> 2) << 2, HRc = (((H - 1 - C) * fct + 3) >> 2) << 2;
        const int PW = HLc + tile + HRc;
        if (PW / 4 > 512 || PW - tile > Nc) return false;
        if (inv && 2 * 4 * (size_t)PW * sizeof(float) + 64 > 64 * 1024) return false;
    }
    return true;
}
void* swt_batch_create_f32(int nimg, float* const* d_images, float** const* d_coeffs, float* const* d_tmps, pdwt_info w)
{
    if (nimg < 1 || nimg > 65535 || !d_images || !d_coeffs || !d_tmps || w.ndims != 2 || !w.do_swt || w.nlevels < 1 || w.nlevels > 30) return nullptr;
    if (knob(KN_SWTF) != 1 || (w.hlen > 20 && knob(KN_SWTF_LONG) != 1)) return nullptr;
    for (int lev = 0; lev < w.nlevels; lev++)
        if (!swt_batch_level_ok(w.Nr, w.Nc, w.hlen, 1 << lev)) return nullptr;
    SwtBatch* B = new (std::nothrow) SwtBatch();
    if (!B) return nullptr;
    B->nimg = nimg;
    B->L = w.nlevels;
    B->Nr = w.Nr;
    B->Nc = w.Nc;
    B->hlen = w.hlen;
    B->d_fwd = B->d_inv = nullptr;
    const int L = B->L;
    std::vector<unsigned long long> hf((size_t)L * nimg * 5), hi((size_t)L * nimg * 5);
    auto al = [](const void* p) { return p && ((uintptr_t)p & 15) == 0; };
    for (int b = 0; b < nimg; b++) {
        float* const* c = d_coeffs[b];
        if (!d_images[b] || !c || !d_tmps[b]) {
            delete B;
            return nullptr;
        }
        float* t1 = d_tmps[b];
        float* t2 = d_tmps[b] + (size_t)w.Nr * w.Nc;
        const float* in = d_images[b];
        for (int lev = 0; lev < L; lev++) {  // forward_swt's level loop
            float* aout = (lev == L - 1) ? c[0] : ((lev & 1) ? t2 : t1);
            const void* e5[5] = {in, aout, c[3 * lev + 1], c[3 * lev + 2], c[3 * lev + 3]};
            for (int k = 0; k < 5; k++) {
                if (!al(e5[k]) || (k > 0 && e5[k] == e5[0])) {
                    delete B;
                    return nullptr;
                }
                hf[((size_t)lev * nimg + b) * 5 + k] = (unsigned long long)(uintptr_t)e5[k];
            }
            in = aout;
        }
        const float* a = c[0];
        for (int i = L - 1; i >= 0; i--) {  // inverse_swt's level loop
            float* out = (i == 0) ? d_images[b] : ((i & 1) ? t2 : t1);
            const void* e5[5] = {a, c[3 * i + 1], c[3 * i + 2], c[3 * i + 3], out};
            for (int k = 0; k < 5; k++) {
                if (!al(e5[k]) || (k < 4 && e5[k] == e5[4])) {
                    delete B;
                    return nullptr;
                }
                hi[((size_t)i * nimg + b) * 5 + k] = (unsigned long long)(uintptr_t)e5[k];
            }
            a = out;
        }
    }
    const size_t bytes = hf.size() * sizeof(unsigned long long);
    B->d_fwd = (unsigned long long*)pdwt_malloc(bytes);
    B->d_inv = (unsigned long long*)pdwt_malloc(bytes);
    if (!B->d_fwd || !B->d_inv || pdwt_memcpy_h2d(B->d_fwd, hf.data(), bytes) != PDWT_OK || pdwt_memcpy_h2d(B->d_inv, hi.data(), bytes) != PDWT_OK) {
        pdwt_free(B->d_fwd);
        pdwt_free(B->d_inv);
        delete B;
        return nullptr;
    }
    return B;
}
int swt_batch_forward_f32(void* batch, const pdwt_filters_f32* filt)
{
    SwtBatch* B = (SwtBatch*)batch;
    if (!B || !filt || filt->hlen != B->hlen) return PDWT_EINVAL;
    const Taps2<float> f = taps_fwd<float>(filt);
    for (int lev = 0; lev < B->L; lev++) {
        const int rc = swt_fwd_fused_f32(nullptr, nullptr, nullptr, nullptr, nullptr, B->Nr, B->Nc, B->hlen, 1 << lev, f, B->d_fwd + (size_t)lev * B->nimg * 5, B->nimg);
        if (rc != PDWT_OK) return rc < 0 ? rc : PDWT_EINVAL;  // (forward reads the images, which are intact: the caller may redo the batch image by image)
    }
    return PDWT_OK;
}
int swt_batch_inverse_f32(void* batch, const pdwt_filters_f32* filt)
{
    SwtBatch* B = (SwtBatch*)batch;
    if (!B || !filt || filt->hlen != B->hlen) return PDWT_EINVAL;
    const Taps2<float> f = taps_inv<float>(filt, 0.5f);
    for (int i = B->L - 1; i >= 0; i--) {
        const int rc = swt_inv_fused_f32(nullptr, nullptr, nullptr, nullptr, nullptr, B->Nr, B->Nc, B->hlen, 1 << i, f, B->d_inv + (size_t)i * B->nimg * 5, B->nimg);
        if (rc != PDWT_OK) return rc < 0 ? rc : PDWT_EINVAL;  // (the fused inverse never writes a band: the caller may redo the batch image by image)
    }
    return PDWT_OK;
}
void swt_batch_destroy_f32(void* batch)
{
    SwtBatch* B = (SwtBatch*)batch;
    if (!B) return;
    pdwt_free(B->d_fwd);
    pdwt_free(B->d_inv);
    delete B;
}

}  // namespace pdwt

using namespace pdwt;

extern "C" {
int pdwt_forward_swt_separable_f32(float* i, float** c, float* t, pdwt_info w, const pdwt_filters_f32* f) { return forward_swt<float>(i, c, t, w, f); }
int pdwt_forward_swt_separable_f64(double* i, double** c, double* t, pdwt_info w, const pdwt_filters_f64* f) { return forward_swt<double>(i, c, t, w, f); }
int pdwt_inverse_swt_separable_f32(float* i, float** c, float* t, pdwt_info w, const pdwt_filters_f32* f) { return inverse_swt<float>(i, c, t, w, f); }
int pdwt_inverse_swt_separable_f64(double* i, double** c, double* t, pdwt_info w, const pdwt_filters_f64* f) { return inverse_swt<double>(i, c, t, w, f); }
int pdwt_forward_swt_separable_1d_f32(float* i, float** c, float* t, pdwt_info w, const pdwt_filters_f32* f) { return forward_swt_1d<float>(i, c, t, w, f); }
int pdwt_forward_swt_separable_1d_f64(double* i, double** c, double* t, pdwt_info w, const pdwt_filters_f64* f) { return forward_swt_1d<double>(i, c, t, w, f); }
int pdwt_inverse_swt_separable_1d_f32(float* i, float** c, float* t, pdwt_info w, const pdwt_filters_f32* f) { return inverse_swt_1d<float>(i, c, t, w, f); }
int pdwt_inverse_swt_separable_1d_f64(double* i, double** c, double* t, pdwt_info w, const pdwt_filters_f64* f) { return inverse_swt_1d<double>(i, c, t, w, f); }
}
