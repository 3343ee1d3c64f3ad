// nonsep.hip -- 2-D transform with four genuinely NON-SEPARABLE hlen x hlen kernels (custom filter banks only).
//
// Path replaced: reference src/nonseparable.cu (w_kern_forward / w_kern_inverse / w_kern_forward_swt /
// w_kern_inverse_swt and their drivers, :114-452).  For the named wavelets the reference's four kernels are outer
// products of the 1-D bank; that case never reaches this file -- the class runs the separable kernels on a band table
// with H and V exchanged (wt.cpp: nonsep_table), O(hlen) instead of O(hlen^2) per sample.  What is left is the
// `set_filters_forward(name, len, f1, f2, f3, f4)` case with arbitrary kernels (src/wt.cu:560-602): a plain, coalesced
// one-thread-per-output convolution, taps read through the scalar cache (uniform addresses).  Not a tuned path: the work is
// O(hlen^2) per sample by definition.  Accumulation order (jy outer, jx inner, one FMA per tap) is the reference's, so the
// results are bit-identical to the oracle restatement.
#include "common.hpp"

namespace pdwt {

constexpr int kNsTX = 64, kNsTY = 4;  // 64 consecutive outputs along x per wave

template <typename T>
__global__ __launch_bounds__(kNsTX* kNsTY) void k_ns_forward(const T* __restrict__ img, T* __restrict__ cA, T* __restrict__ cH,
                                                             T* __restrict__ cV, T* __restrict__ cD, int Nr, int Nc, int hlen,
                                                             const T* __restrict__ K)
{  // src/nonseparable.cu:114-170
    const int Nr2 = div2(Nr), Nc2 = div2(Nc);
    const int gx = blockIdx.x * kNsTX + threadIdx.x, gy = blockIdx.y * kNsTY + threadIdx.y;
    if (gx >= Nc2 || gy >= Nr2) return;
    const int c = (hlen & 1) ? hlen / 2 : hlen / 2 - 1;
    const int hh = hlen * hlen;
    T ra = 0, rh = 0, rv = 0, rd = 0;
    for (int jy = 0; jy < hlen; jy++) {
        const T* row = img + (size_t)wrap_ext(2 * gy - c + jy, Nr) * Nc;
        for (int jx = 0; jx < hlen; jx++) {
            const T v = row[wrap_ext(2 * gx - c + jx, Nc)];
            const int k = (hlen - 1 - jy) * hlen + (hlen - 1 - jx);
            ra = fma_t(v, K[k], ra);
            rh = fma_t(v, K[hh + k], rh);
            rv = fma_t(v, K[2 * hh + k], rv);
            rd = fma_t(v, K[3 * hh + k], rd);
        }
    }
    const size_t o = (size_t)gy * Nc2 + gx;
    cA[o] = ra;
    cH[o] = rh;
    cV[o] = rv;
    cD[o] = rd;
}

template <typename T>
__global__ __launch_bounds__(kNsTX* kNsTY) void k_ns_inverse(T* __restrict__ out, const T* __restrict__ cA, const T* __restrict__ cH,
                                                             const T* __restrict__ cV, const T* __restrict__ cD, int Nr, int Nc, int Nro,
                                                             int Nco, int hlen, const T* __restrict__ K)
{  // src/nonseparable.cu:176-226
    const int ox = blockIdx.x * kNsTX + threadIdx.x, oy = blockIdx.y * kNsTY + threadIdx.y;
    if (ox >= Nco || oy >= Nro) return;
    const int h2 = hlen / 2, c = h2 / 2, shift = (h2 & 1) ? 0 : 1, hh = hlen * hlen;
    const int gx = ox + shift, gy = oy + shift;
    const int offx = 1 - (gx & 1), offy = 1 - (gy & 1);
    T ra = 0, rh = 0, rv = 0, rd = 0;
    for (int jy = 0; jy < h2; jy++) {
        const size_t rowo = (size_t)wrap_per(gy / 2 - c + jy, Nr) * Nc;
        for (int jx = 0; jx < h2; jx++) {
            const size_t o = rowo + wrap_per(gx / 2 - c + jx, Nc);
            const int k = (hlen - 1 - (2 * jy + offy)) * hlen + (hlen - 1 - (2 * jx + offx));
            ra = fma_t(cA[o], K[k], ra);
            rh = fma_t(cH[o], K[hh + k], rh);
            rv = fma_t(cV[o], K[2 * hh + k], rv);
            rd = fma_t(cD[o], K[3 * hh + k], rd);
        }
    }
    out[(size_t)oy * Nco + ox] = ra + rh + rv + rd;
}

template <typename T, bool INV>
__global__ __launch_bounds__(kNsTX* kNsTY) void k_ns_swt(const T* __restrict__ inA, const T* __restrict__ inH, const T* __restrict__ inV,
                                                         const T* __restrict__ inD, T* __restrict__ oA, T* __restrict__ oH,
                                                         T* __restrict__ oV, T* __restrict__ oD, int Nr, int Nc, int hlen, int fac,
                                                         const T* __restrict__ K)
{  // src/nonseparable.cu:301-400
    const int gx = blockIdx.x * kNsTX + threadIdx.x, gy = blockIdx.y * kNsTY + threadIdx.y;
    if (gx >= Nc || gy >= Nr) return;
    const int c = (INV ? hlen / 2 : ((hlen & 1) ? hlen / 2 : hlen / 2 - 1)) * fac;
    const int hh = hlen * hlen;
    T ra = 0, rh = 0, rv = 0, rd = 0;
    for (int jy = 0; jy < hlen; jy++) {
        const size_t rowo = (size_t)wrap_per(gy - c + fac * jy, Nr) * Nc;
        for (int jx = 0; jx < hlen; jx++) {
            const size_t o = rowo + wrap_per(gx - c + fac * jx, Nc);
            const int k = (hlen - 1 - jy) * hlen + (hlen - 1 - jx);
            if constexpr (!INV) {
                const T v = inA[o];
                ra = fma_t(v, K[k], ra);
                rh = fma_t(v, K[hh + k], rh);
                rv = fma_t(v, K[2 * hh + k], rv);
                rd = fma_t(v, K[3 * hh + k], rd);
            } else {  // res += c * k / 4: the product is rounded before it is quartered and added (:386-389)
                ra += inA[o] * K[k] / T(4);
                rh += inH[o] * K[hh + k] / T(4);
                rv += inV[o] * K[2 * hh + k] / T(4);
                rd += inD[o] * K[3 * hh + k] / T(4);
            }
        }
    }
    const size_t o = (size_t)gy * Nc + gx;
    if constexpr (!INV) {
        oA[o] = ra;
        oH[o] = rh;
        oV[o] = rv;
        oD[o] = rd;
    } else {
        oA[o] = ra + rh + rv + rd;
    }
}

#define PDWT_CHECK_LAUNCH() PDWT_HIP_TRY(hipGetLastError())

static int ns_check(const void* img, const void* c, const void* tmp, const pdwt_info& w, const void* K)
{
    if (!img || !c || !tmp || !K) return PDWT_EINVAL;
    if (w.ndims != 2 || w.Nr < 1 || w.Nc < 1 || w.nlevels < 1 || w.nlevels > 32) return PDWT_EINVAL;
    if (w.hlen < 1 || w.hlen > PDWT_MAX_FILTER_WIDTH) return PDWT_EINVAL;
    return PDWT_OK;
}
static dim3 ns_grid(int nx, int ny) { return dim3(idiv_up(nx, kNsTX), idiv_up(ny, kNsTY)); }

// w_forward, src/nonseparable.cu:233-258
template <typename T>
static int ns_forward(T* img, T** c, T* tmp, pdwt_info w, const T* K)
{
    int rc = ns_check(img, c, tmp, w, K);
    if (rc != PDWT_OK) return rc;
    T* bufs[2] = {tmp, tmp + (((size_t)div2(w.Nr) * div2(w.Nc) + 63) & ~(size_t)63)};
    const T* in = img;
    int nr = w.Nr, nc = w.Nc;
    for (int i = 0; i < w.nlevels; i++) {
        T* aout = (i == w.nlevels - 1) ? c[0] : bufs[i & 1];
        KTimer kt(K_FWD2D_FUSED);
        hipLaunchKernelGGL(k_ns_forward<T>, ns_grid(div2(nc), div2(nr)), dim3(kNsTX, kNsTY), 0, stream(), in, aout, c[3 * i + 1], c[3 * i + 2],
                           c[3 * i + 3], nr, nc, w.hlen, K);
        PDWT_CHECK_LAUNCH();
        in = aout;
        nr = div2(nr);
        nc = div2(nc);
    }
    return PDWT_OK;
}

// w_inverse, src/nonseparable.cu:261-292 (K = the inverse kernels)
template <typename T>
static int ns_inverse(T* img, T** c, T* tmp, pdwt_info w, const T* K)
{
    int rc = ns_check(img, c, tmp, w, K);
    if (rc != PDWT_OK) return rc;
    int tNr[34], tNc[34];
    tNr[0] = w.Nr;
    tNc[0] = w.Nc;
    for (int i = 1; i <= w.nlevels; i++) {
        tNr[i] = div2(tNr[i - 1]);
        tNc[i] = div2(tNc[i - 1]);
    }
    T* bufs[2] = {tmp, tmp + (((size_t)tNr[1] * tNc[1] + 63) & ~(size_t)63)};
    const T* a = c[0];
    for (int i = w.nlevels - 1; i >= 0; i--) {
        T* out = (i == 0) ? img : bufs[i & 1];
        KTimer kt(K_INV2D_FUSED);
        hipLaunchKernelGGL(k_ns_inverse<T>, ns_grid(tNc[i], tNr[i]), dim3(kNsTX, kNsTY), 0, stream(), out, a, (const T*)c[3 * i + 1],
                           (const T*)c[3 * i + 2], (const T*)c[3 * i + 3], tNr[i + 1], tNc[i + 1], tNr[i], tNc[i], w.hlen, K);
        PDWT_CHECK_LAUNCH();
        a = out;
    }
    return PDWT_OK;
}

// w_forward_swt / w_inverse_swt, src/nonseparable.cu:408-452
template <typename T>
static int ns_forward_swt(T* img, T** c, T* tmp, pdwt_info w, const T* K)
{
    int rc = ns_check(img, c, tmp, w, K);
    if (rc != PDWT_OK) return rc;
    const size_t n = ((size_t)w.Nr * w.Nc + 63) & ~(size_t)63;
    T* bufs[2] = {tmp, tmp + n};
    const T* in = img;
    for (int i = 0; i < w.nlevels; i++) {
        T* aout = (i == w.nlevels - 1) ? c[0] : bufs[i & 1];
        KTimer kt(K_SWT_ANA_COLS);
        hipLaunchKernelGGL((k_ns_swt<T, false>), ns_grid(w.Nc, w.Nr), dim3(kNsTX, kNsTY), 0, stream(), in, (const T*)nullptr, (const T*)nullptr,
                           (const T*)nullptr, aout, c[3 * i + 1], c[3 * i + 2], c[3 * i + 3], w.Nr, w.Nc, w.hlen, 1 << i, K);
        PDWT_CHECK_LAUNCH();
        in = aout;
    }
    return PDWT_OK;
}
template <typename T>
static int ns_inverse_swt(T* img, T** c, T* tmp, pdwt_info w, const T* K)
{
    int rc = ns_check(img, c, tmp, w, K);
    if (rc != PDWT_OK) return rc;
    const size_t n = ((size_t)w.Nr * w.Nc + 63) & ~(size_t)63;
    T* bufs[2] = {tmp, tmp + n};
    const T* a = c[0];
    for (int i = w.nlevels - 1; i >= 0; i--) {
        T* out = (i == 0) ? img : bufs[i & 1];
        KTimer kt(K_SWT_SYN_COLS);
        hipLaunchKernelGGL((k_ns_swt<T, true>), ns_grid(w.Nc, w.Nr), dim3(kNsTX, kNsTY), 0, stream(), a, (const T*)c[3 * i + 1],
                           (const T*)c[3 * i + 2], (const T*)c[3 * i + 3], out, (T*)nullptr, (T*)nullptr, (T*)nullptr, w.Nr, w.Nc, w.hlen, 1 << i, K);
        PDWT_CHECK_LAUNCH();
        a = out;
    }
    return PDWT_OK;
}

}  // namespace pdwt

using namespace pdwt;

extern "C" {
#define PDWT_NS_API(S, T)                                                                                                              \
    int pdwt_forward_nonseparable_##S(T* img, T** c, T* tmp, pdwt_info w, const T* d_k) { return ns_forward<T>(img, c, tmp, w, d_k); }     \
    int pdwt_inverse_nonseparable_##S(T* img, T** c, T* tmp, pdwt_info w, const T* d_k) { return ns_inverse<T>(img, c, tmp, w, d_k); }     \
    int pdwt_forward_swt_nonseparable_##S(T* img, T** c, T* tmp, pdwt_info w, const T* d_k) { return ns_forward_swt<T>(img, c, tmp, w, d_k); } \
    int pdwt_inverse_swt_nonseparable_##S(T* img, T** c, T* tmp, pdwt_info w, const T* d_k) { return ns_inverse_swt<T>(img, c, tmp, w, d_k); }
PDWT_NS_API(f32, float)
PDWT_NS_API(f64, double)
}
