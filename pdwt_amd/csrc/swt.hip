// swt.hip -- stationary (undecimated, a-trous) separable transform: kernels + level drivers.
//
// Path replaced: reference src/separable.cu:409-672 (w_kern_forward_swt_pass1/2,
// w_kern_inverse_swt_pass1/2 and their four drivers).  Math: SURVEY.md Appendix A-3 / A-4.
// Every level works at full resolution with tap spacing f = 2^(level-1); a fused row+column tile
// would need an (hlen-1)*f halo in BOTH directions (208 samples at level 5 of db7), so each level is
// two streaming passes whose dilated taps are served by L1/L2: x stays the lane axis, so every tap
// is a fully coalesced 256-byte row segment; the column pass never strides a wave across rows.
#include "common.hpp"

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
            KTimer kt(K_SWT_ANA_ROWS);
            hipLaunchKernelGGL(k_swt_ana_rows<T>, grid, dim3(kSwtThreads), 0, stream(), in, t1, t2, w.Nr, w.Nc, w.hlen, 1 << lev, f);
            PDWT_CHECK_LAUNCH();
        }
        {
            KTimer kt(K_SWT_ANA_COLS);
            hipLaunchKernelGGL(k_swt_ana_cols<T>, grid, dim3(kSwtThreads), 0, stream(), (const T*)t1, (const T*)t2, c[0], c[3 * lev + 1],
                               c[3 * lev + 2], c[3 * lev + 3], w.Nr, w.Nc, w.hlen, 1 << lev, f);
            PDWT_CHECK_LAUNCH();
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
    for (int i = w.nlevels - 1; i >= 0; i--) {
        {
            KTimer kt(K_SWT_SYN_COLS);
            hipLaunchKernelGGL(k_swt_syn_cols<T>, grid, dim3(kSwtThreads), 0, stream(), (const T*)c[0], (const T*)c[3 * i + 1],
                               (const T*)c[3 * i + 2], (const T*)c[3 * i + 3], t1, t2, w.Nr, w.Nc, w.hlen, 1 << i, f);
            PDWT_CHECK_LAUNCH();
        }
        {
            KTimer kt(K_SWT_SYN_ROWS);
            T* out = (i == 0) ? d_image : c[0];
            hipLaunchKernelGGL(k_swt_syn_rows<T>, grid, dim3(kSwtThreads), 0, stream(), (const T*)t1, (const T*)t2, out, w.Nr, w.Nc, w.hlen,
                               1 << i, f);
            PDWT_CHECK_LAUNCH();
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
        KTimer kt(K_SWT_ANA_ROWS);
        hipLaunchKernelGGL(k_swt_ana_rows<T>, grid, dim3(kSwtThreads), 0, stream(), in, aout, c[lev + 1], w.Nr, w.Nc, w.hlen, 1 << lev, f);
        PDWT_CHECK_LAUNCH();
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
        KTimer kt(K_SWT_SYN_ROWS);
        hipLaunchKernelGGL(k_swt_syn_rows<T>, grid, dim3(kSwtThreads), 0, stream(), a, (const T*)c[i + 1], out, w.Nr, w.Nc, w.hlen, 1 << i, f);
        PDWT_CHECK_LAUNCH();
        a = out;
    }
    return PDWT_OK;
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
