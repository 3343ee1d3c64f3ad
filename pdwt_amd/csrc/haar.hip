// haar.hip -- Haar fast path (2-tap filters, no halo, no wrap): pure streaming kernels + drivers.
//
// Path replaced: reference src/haar.cu:10-221 (kern_haar2d_fwd/inv, kern_haar1d_fwd/inv and their
// drivers).  Math: SURVEY.md Appendix A-5.  The arithmetic forms are kept exactly
// (0.5*((a+c)+(b+d)) etc.; 1-D: product with the DOUBLE literal 0.70710678118654746 rounded once,
// src/haar.cu:128,143-144) so results are bit-identical to the CPU oracle for any finite input.
//
// MI355X design: one thread per 2x2 input quad in BOTH directions (the reference's inverse uses one
// thread per output sample and loads each coefficient four times, src/haar.cu:45-48); lanes run
// along x; in the aligned even-size case each thread moves 2 quads with 16-byte loads / 8-byte
// stores (forward) or 8-byte loads / 16-byte stores (inverse).
#include <new>
#include <vector>

#include "common.hpp"

namespace pdwt {

constexpr int kHaarThreads = 256;
#define ONE_SQRT2 0.70710678118654746 /* src/haar.cu:128 */

template <typename T> struct Vec2;
template <> struct Vec2<float> { using type = float2; };
template <> struct Vec2<double> { using type = double2; };
template <typename T> struct Vec4;
template <> struct Vec4<float> { using type = float4; };
template <> struct Vec4<double> { using type = double4; };

template <typename T>
__device__ __forceinline__ void butterfly(T a, T b, T c, T d, T& A, T& H, T& V, T& D)
{
    // src/haar.cu:32-35 (HAAR_AVG = a+b, HAAR_DIF = a-b; 0.5 is exact in either precision)
    A = T(0.5) * ((a + c) + (b + d));
    V = T(0.5) * ((a + c) - (b + d));
    H = T(0.5) * ((a - c) + (b - d));
    D = T(0.5) * ((a - c) - (b - d));
}

// forward 2D: in Nr x Nc -> 4 bands Nr2 x Nc2.  VEC: Nc % 4 == 0 and Nr even -> 2 quads / thread.
template <typename T, bool VEC>
__global__ __launch_bounds__(kHaarThreads) void k_haar2d_fwd(const T* __restrict__ in, T* __restrict__ cA, T* __restrict__ cH,
                                                             T* __restrict__ cV, T* __restrict__ cD, int Nr, int Nc, const void* tbl)
{
    if (tbl) {  // batched launch (pdwt_batch2d_*): gridDim.z = image, five pointers per image (in, cA, cH, cV, cD)
        typedef const unsigned long long __attribute__((address_space(4))) * tbl_t;
        const tbl_t q = (tbl_t)(const unsigned long long*)tbl + 5 * (size_t)blockIdx.z;
        in = (const T*)q[0];
        cA = (T*)q[1];
        cH = (T*)q[2];
        cV = (T*)q[3];
        cD = (T*)q[4];
    }
    const int Nr2 = div2(Nr), Nc2 = div2(Nc);
    const int gy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (gy >= Nr2) return;
    if constexpr (VEC) {
        using V4 = typename Vec4<T>::type;
        using V2 = typename Vec2<T>::type;
        const int q = blockIdx.x * 64 + (threadIdx.x & 63);  // pair of output columns
        if (2 * q >= Nc2) return;
        const V4 r0 = *reinterpret_cast<const V4*>(in + (size_t)(2 * gy) * Nc + 4 * q);
        const V4 r1 = *reinterpret_cast<const V4*>(in + (size_t)(2 * gy + 1) * Nc + 4 * q);
        V2 A, H, V, D;
        butterfly<T>(r0.x, r0.y, r1.x, r1.y, A.x, H.x, V.x, D.x);
        butterfly<T>(r0.z, r0.w, r1.z, r1.w, A.y, H.y, V.y, D.y);
        const size_t o = (size_t)gy * Nc2 + 2 * q;
        *reinterpret_cast<V2*>(cA + o) = A;
        *reinterpret_cast<V2*>(cH + o) = H;
        *reinterpret_cast<V2*>(cV + o) = V;
        *reinterpret_cast<V2*>(cD + o) = D;
    } else {
        const int gx = blockIdx.x * 64 + (threadIdx.x & 63);
        if (gx >= Nc2) return;
        // odd sizes: clamp the +1 neighbour (virtual repeat of the last sample), src/haar.cu:20-25
        const int x0 = 2 * gx, x1 = (2 * gx + 1 == Nc) ? Nc - 1 : 2 * gx + 1;
        const int y0 = 2 * gy, y1 = (2 * gy + 1 == Nr) ? Nr - 1 : 2 * gy + 1;
        const T a = in[(size_t)y0 * Nc + x0], b = in[(size_t)y0 * Nc + x1];
        const T c = in[(size_t)y1 * Nc + x0], d = in[(size_t)y1 * Nc + x1];
        T A, H, V, D;
        butterfly<T>(a, b, c, d, A, H, V, D);
        const size_t o = (size_t)gy * Nc2 + gx;
        cA[o] = A;
        cH[o] = H;
        cV[o] = V;
        cD[o] = D;
    }
}

// inverse 2D: bands Nri x Nci -> out Nro x Nco (Nro in {2Nri, 2Nri-1}).  One thread per coefficient
// quad (src/haar.cu:45-54: a=A, b=V, c=H, d=D through the same butterfly).
template <typename T, bool VEC>
__global__ __launch_bounds__(kHaarThreads) void k_haar2d_inv(T* __restrict__ out, const T* __restrict__ cA, const T* __restrict__ cH,
                                                             const T* __restrict__ cV, const T* __restrict__ cD, int Nri, int Nci, int Nro,
                                                             int Nco, const void* tbl)
{
    if (tbl) {  // batched launch: (out, cA, cH, cV, cD) of image blockIdx.z
        typedef const unsigned long long __attribute__((address_space(4))) * tbl_t;
        const tbl_t q = (tbl_t)(const unsigned long long*)tbl + 5 * (size_t)blockIdx.z;
        out = (T*)q[0];
        cA = (const T*)q[1];
        cH = (const T*)q[2];
        cV = (const T*)q[3];
        cD = (const T*)q[4];
    }
    const int gy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (gy >= Nri) return;
    if constexpr (VEC) {  // Nci even, Nco == 2*Nci, Nro == 2*Nri
        using V4 = typename Vec4<T>::type;
        using V2 = typename Vec2<T>::type;
        const int q = blockIdx.x * 64 + (threadIdx.x & 63);
        if (2 * q >= Nci) return;
        const size_t o = (size_t)gy * Nci + 2 * q;
        const V2 A = *reinterpret_cast<const V2*>(cA + o), H = *reinterpret_cast<const V2*>(cH + o);
        const V2 V = *reinterpret_cast<const V2*>(cV + o), D = *reinterpret_cast<const V2*>(cD + o);
        V4 r0, r1;
        butterfly<T>(A.x, V.x, H.x, D.x, r0.x, r1.x, r0.y, r1.y);
        butterfly<T>(A.y, V.y, H.y, D.y, r0.z, r1.z, r0.w, r1.w);
        *reinterpret_cast<V4*>(out + (size_t)(2 * gy) * Nco + 4 * q) = r0;
        *reinterpret_cast<V4*>(out + (size_t)(2 * gy + 1) * Nco + 4 * q) = r1;
    } else {
        const int gx = blockIdx.x * 64 + (threadIdx.x & 63);
        if (gx >= Nci) return;
        const size_t o = (size_t)gy * Nci + gx;
        T ee, oe, eo, oo;  // (even x,even y) (odd x,even y) (even x,odd y) (odd x,odd y)
        // butterfly(a,b,c,d) returns (A: ++, H: (a-c)+(b-d), V: (a+c)-(b+d), D); with a=A b=V c=H d=D
        // the reference's four cases (src/haar.cu:50-53) are: ee = "A" form, oe = "V" form, eo = "H" form, oo = "D" form
        butterfly<T>(cA[o], cV[o], cH[o], cD[o], ee, eo, oe, oo);
        const int y0 = 2 * gy, x0 = 2 * gx;
        out[(size_t)y0 * Nco + x0] = ee;
        if (x0 + 1 < Nco) out[(size_t)y0 * Nco + x0 + 1] = oe;
        if (y0 + 1 < Nro) {
            out[(size_t)(y0 + 1) * Nco + x0] = eo;
            if (x0 + 1 < Nco) out[(size_t)(y0 + 1) * Nco + x0 + 1] = oo;
        }
    }
}

// 1-D along rows: A = s*(x0+x1), D = s*(x0-x1), s = double literal (src/haar.cu:141-144)
template <typename T>
__global__ __launch_bounds__(kHaarThreads) void k_haar1d_fwd(const T* __restrict__ in, T* __restrict__ cA, T* __restrict__ cD, int Nr, int Nc)
{
    const int Nc2 = div2(Nc);
    const int gy = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int gx = blockIdx.x * 64 + (threadIdx.x & 63);
    if (gy >= Nr || gx >= Nc2) return;
    const int x1 = (2 * gx + 1 == Nc) ? Nc - 1 : 2 * gx + 1;
    const T a = in[(size_t)gy * Nc + 2 * gx], b = in[(size_t)gy * Nc + x1];
    cA[(size_t)gy * Nc2 + gx] = (T)(ONE_SQRT2 * (double)(a + b));
    cD[(size_t)gy * Nc2 + gx] = (T)(ONE_SQRT2 * (double)(a - b));
}

template <typename T>
__global__ __launch_bounds__(kHaarThreads) void k_haar1d_inv(T* __restrict__ out, const T* __restrict__ cA, const T* __restrict__ cD, int Nr,
                                                             int Nci, int Nco)
{
    const int gy = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int gx = blockIdx.x * 64 + (threadIdx.x & 63);  // coefficient column
    if (gy >= Nr || gx >= Nci) return;
    const T a = cA[(size_t)gy * Nci + gx], b = cD[(size_t)gy * Nci + gx];
    out[(size_t)gy * Nco + 2 * gx] = (T)(ONE_SQRT2 * (double)(a + b));
    if (2 * gx + 1 < Nco) out[(size_t)gy * Nco + 2 * gx + 1] = (T)(ONE_SQRT2 * (double)(a - b));
}

#define PDWT_CHECK_LAUNCH() PDWT_HIP_TRY(hipGetLastError())

static int check(const void* img, const void* c, const void* tmp, const pdwt_info& w, int ndims)
{
    if (!img || !c || !tmp) return PDWT_EINVAL;
    if (w.Nr < 1 || w.Nc < 1 || w.nlevels < 1 || w.nlevels > 32 || w.ndims != ndims) return PDWT_EINVAL;
    return PDWT_OK;
}

template <typename T>
static bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

template <typename T>
static int haar2d_fwd_level(const T* in, T* cA, T* cH, T* cV, T* cD, int nr, int nc, const void* d_tbl = nullptr, int nimg = 1, bool tbl_vec = false)
{
    const int nr2 = div2(nr), nc2 = div2(nc);
    // (a batch: the caller has checked the alignment of every image's pointers)
    const bool vec = d_tbl ? tbl_vec
                           : (nc % 4 == 0) && (nr % 2 == 0) && aligned16<T>(in) && aligned16<T>(cA) && aligned16<T>(cH) && aligned16<T>(cV) &&
                                 aligned16<T>(cD) && sizeof(T) == 4;
    KTimer kt(K_HAAR2D_FWD);
    if (vec) {
        dim3 grid(idiv_up(nc2 / 2, 64), idiv_up(nr2, 4), nimg);
        hipLaunchKernelGGL((k_haar2d_fwd<T, true>), grid, dim3(kHaarThreads), 0, stream(), in, cA, cH, cV, cD, nr, nc, d_tbl);
    } else {
        dim3 grid(idiv_up(nc2, 64), idiv_up(nr2, 4), nimg);
        hipLaunchKernelGGL((k_haar2d_fwd<T, false>), grid, dim3(kHaarThreads), 0, stream(), in, cA, cH, cV, cD, nr, nc, d_tbl);
    }
    PDWT_CHECK_LAUNCH();
    return PDWT_OK;
}

template <typename T>
static int haar2d_inv_level(T* out, const T* cA, const T* cH, const T* cV, const T* cD, int nri, int nci, int nro, int nco, const void* d_tbl = nullptr,
                            int nimg = 1, bool tbl_vec = false)
{
    const bool vec = d_tbl ? tbl_vec
                           : (nci % 2 == 0) && (nco == 2 * nci) && (nro == 2 * nri) && aligned16<T>(out) && aligned16<T>(cA) && aligned16<T>(cH) &&
                                 aligned16<T>(cV) && aligned16<T>(cD) && sizeof(T) == 4;
    KTimer kt(K_HAAR2D_INV);
    if (vec) {
        dim3 grid(idiv_up(nci / 2, 64), idiv_up(nri, 4), nimg);
        hipLaunchKernelGGL((k_haar2d_inv<T, true>), grid, dim3(kHaarThreads), 0, stream(), out, cA, cH, cV, cD, nri, nci, nro, nco, d_tbl);
    } else {
        dim3 grid(idiv_up(nci, 64), idiv_up(nri, 4), nimg);
        hipLaunchKernelGGL((k_haar2d_inv<T, false>), grid, dim3(kHaarThreads), 0, stream(), out, cA, cH, cV, cD, nri, nci, nro, nco, d_tbl);
    }
    PDWT_CHECK_LAUNCH();
    return PDWT_OK;
}

static size_t up64(size_t n) { return (n + 63) & ~(size_t)63; }

// haar_forward2d, src/haar.cu:61-86
template <typename T>
static int haar_forward2d(T* d_image, T** c, T* d_tmp, pdwt_info w)
{
    int rc = check(d_image, c, d_tmp, w, 2);
    if (rc != PDWT_OK) return rc;
    T* ping[2] = {d_tmp, d_tmp + up64((size_t)div2(w.Nr) * div2(w.Nc))};
    const T* in = d_image;
    int nr = w.Nr, nc = w.Nc;
    for (int lev = 0; lev < w.nlevels; lev++) {
        T* aout = (lev == w.nlevels - 1) ? c[0] : ping[lev & 1];
        rc = haar2d_fwd_level<T>(in, aout, c[3 * lev + 1], c[3 * lev + 2], c[3 * lev + 3], nr, nc);
        if (rc != PDWT_OK) return rc;
        in = aout;
        nr = div2(nr);
        nc = div2(nc);
    }
    return PDWT_OK;
}

// haar_inverse2d, src/haar.cu:88-119
template <typename T>
static int haar_inverse2d(T* d_image, T** c, T* d_tmp, pdwt_info w)
{
    int rc = check(d_image, c, d_tmp, w, 2);
    if (rc != PDWT_OK) return rc;
    int tNr[34], tNc[34];
    tNr[0] = w.Nr;
    tNc[0] = w.Nc;
    for (int i = 1; i <= w.nlevels; i++) {
        tNr[i] = div2(tNr[i - 1]);
        tNc[i] = div2(tNc[i - 1]);
    }
    T* ping[2] = {d_tmp, d_tmp + up64((size_t)tNr[1] * tNc[1])};
    const T* a = c[0];
    for (int i = w.nlevels - 1; i >= 0; i--) {
        T* out = (i == 0) ? d_image : ping[i & 1];
        rc = haar2d_inv_level<T>(out, a, c[3 * i + 1], c[3 * i + 2], c[3 * i + 3], tNr[i + 1], tNc[i + 1], tNr[i], tNc[i]);
        if (rc != PDWT_OK) return rc;
        a = out;
    }
    return PDWT_OK;
}

// -------------------------------------------------------------------------------------------------
// a batch of equally sized images (pdwt_batch2d_*, dwt.hip, when the bank is Haar): every level of all images in one launch
// (gridDim.z = image), the per-image pointers of haar_forward2d / haar_inverse2d in device-side tables built once
// -------------------------------------------------------------------------------------------------
template <typename T>
struct HaarBatch {
    int nimg, L;
    int nr[34], nc[34];
    bool fvec[33], ivec[33];       // 16-byte form of the level (every image's pointers aligned)
    unsigned long long* d_fwd;     // [level][image][5]: in, cA, cH, cV, cD
    unsigned long long* d_inv;     // [level][image][5]: out, cA, cH, cV, cD
};
template <typename T>
void* haar_batch_create(int nimg, T* const* d_images, T** const* d_coeffs, T* const* d_tmps, pdwt_info w)
{
    if (nimg < 1 || nimg > 65535 || !d_images || !d_coeffs || !d_tmps || w.ndims != 2 || w.do_swt || w.nlevels < 1 || w.nlevels > 32) return nullptr;
    HaarBatch<T>* B = new (std::nothrow) HaarBatch<T>();
    if (!B) return nullptr;
    B->nimg = nimg;
    B->L = w.nlevels;
    B->d_fwd = B->d_inv = nullptr;
    B->nr[0] = w.Nr;
    B->nc[0] = w.Nc;
    for (int i = 1; i <= B->L; i++) {
        B->nr[i] = div2(B->nr[i - 1]);
        B->nc[i] = div2(B->nc[i - 1]);
    }
    const int L = B->L;
    std::vector<unsigned long long> hf((size_t)L * nimg * 5), hi((size_t)L * nimg * 5);
    for (int lev = 0; lev < L; lev++) {
        B->fvec[lev] = sizeof(T) == 4 && (B->nc[lev] % 4 == 0) && (B->nr[lev] % 2 == 0);
        B->ivec[lev] = sizeof(T) == 4 && (B->nc[lev + 1] % 2 == 0) && (B->nc[lev] == 2 * B->nc[lev + 1]) && (B->nr[lev] == 2 * B->nr[lev + 1]);
    }
    for (int b = 0; b < nimg; b++) {
        T* const* c = d_coeffs[b];
        if (!d_images[b] || !c || !d_tmps[b]) {
            delete B;
            return nullptr;
        }
        T* ping[2] = {d_tmps[b], d_tmps[b] + up64((size_t)div2(w.Nr) * div2(w.Nc))};
        const T* in = d_images[b];
        for (int lev = 0; lev < L; lev++) {  // haar_forward2d
            T* aout = (lev == L - 1) ? c[0] : ping[lev & 1];
            const void* e5[5] = {in, aout, c[3 * lev + 1], c[3 * lev + 2], c[3 * lev + 3]};
            for (int k = 0; k < 5; k++) {
                if (!e5[k]) {
                    delete B;
                    return nullptr;
                }
                hf[((size_t)lev * nimg + b) * 5 + k] = (unsigned long long)(uintptr_t)e5[k];
                B->fvec[lev] = B->fvec[lev] && aligned16<T>(e5[k]);
            }
            in = aout;
        }
        const T* a = c[0];
        for (int i = L - 1; i >= 0; i--) {  // haar_inverse2d
            T* out = (i == 0) ? d_images[b] : ping[i & 1];
            const void* e5[5] = {out, a, c[3 * i + 1], c[3 * i + 2], c[3 * i + 3]};
            for (int k = 0; k < 5; k++) {
                hi[((size_t)i * nimg + b) * 5 + k] = (unsigned long long)(uintptr_t)e5[k];
                B->ivec[i] = B->ivec[i] && aligned16<T>(e5[k]);
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
template <typename T>
int haar_batch_forward(void* batch)
{
    HaarBatch<T>* B = (HaarBatch<T>*)batch;
    if (!B) return PDWT_EINVAL;
    for (int lev = 0; lev < B->L; lev++) {
        const int rc = haar2d_fwd_level<T>(nullptr, nullptr, nullptr, nullptr, nullptr, B->nr[lev], B->nc[lev], B->d_fwd + (size_t)lev * B->nimg * 5, B->nimg, B->fvec[lev]);
        if (rc != PDWT_OK) return rc;
    }
    return PDWT_OK;
}
template <typename T>
int haar_batch_inverse(void* batch)
{
    HaarBatch<T>* B = (HaarBatch<T>*)batch;
    if (!B) return PDWT_EINVAL;
    for (int i = B->L - 1; i >= 0; i--) {
        const int rc = haar2d_inv_level<T>(nullptr, nullptr, nullptr, nullptr, nullptr, B->nr[i + 1], B->nc[i + 1], B->nr[i], B->nc[i], B->d_inv + (size_t)i * B->nimg * 5, B->nimg,
                                           B->ivec[i]);
        if (rc != PDWT_OK) return rc;
    }
    return PDWT_OK;
}
template <typename T>
void haar_batch_destroy(void* batch)
{
    HaarBatch<T>* B = (HaarBatch<T>*)batch;
    if (!B) return;
    pdwt_free(B->d_fwd);
    pdwt_free(B->d_inv);
    delete B;
}
template void* haar_batch_create<float>(int, float* const*, float** const*, float* const*, pdwt_info);
template void* haar_batch_create<double>(int, double* const*, double** const*, double* const*, pdwt_info);
template int haar_batch_forward<float>(void*);
template int haar_batch_forward<double>(void*);
template int haar_batch_inverse<float>(void*);
template int haar_batch_inverse<double>(void*);
template void haar_batch_destroy<float>(void*);
template void haar_batch_destroy<double>(void*);

// haar_forward1d, src/haar.cu:163-186
template <typename T>
static int haar_forward1d(T* d_image, T** c, T* d_tmp, pdwt_info w)
{
    int rc = check(d_image, c, d_tmp, w, 1);
    if (rc != PDWT_OK) return rc;
    T* ping[2] = {d_tmp, d_tmp + up64((size_t)w.Nr * div2(w.Nc))};
    const T* in = d_image;
    int nc = w.Nc;
    for (int lev = 0; lev < w.nlevels; lev++) {
        T* aout = (lev == w.nlevels - 1) ? c[0] : ping[lev & 1];
        dim3 grid(idiv_up(div2(nc), 64), idiv_up(w.Nr, 4));
        KTimer kt(K_HAAR1D_FWD);
        hipLaunchKernelGGL(k_haar1d_fwd<T>, grid, dim3(kHaarThreads), 0, stream(), in, aout, c[lev + 1], w.Nr, nc);
        PDWT_CHECK_LAUNCH();
        in = aout;
        nc = div2(nc);
    }
    return PDWT_OK;
}

// haar_inverse1d, src/haar.cu:193-221
template <typename T>
static int haar_inverse1d(T* d_image, T** c, T* d_tmp, pdwt_info w)
{
    int rc = check(d_image, c, d_tmp, w, 1);
    if (rc != PDWT_OK) return rc;
    int tNc[34];
    tNc[0] = w.Nc;
    for (int i = 1; i <= w.nlevels; i++) tNc[i] = div2(tNc[i - 1]);
    T* ping[2] = {d_tmp, d_tmp + up64((size_t)w.Nr * tNc[1])};
    const T* a = c[0];
    for (int i = w.nlevels - 1; i >= 0; i--) {
        T* out = (i == 0) ? d_image : ping[i & 1];
        dim3 grid(idiv_up(tNc[i + 1], 64), idiv_up(w.Nr, 4));
        KTimer kt(K_HAAR1D_INV);
        hipLaunchKernelGGL(k_haar1d_inv<T>, grid, dim3(kHaarThreads), 0, stream(), out, a, (const T*)c[i + 1], w.Nr, tNc[i + 1], tNc[i]);
        PDWT_CHECK_LAUNCH();
        a = out;
    }
    return PDWT_OK;
}

}  // namespace pdwt

using namespace pdwt;

extern "C" {
int pdwt_haar_forward2d_f32(float* i, float** c, float* t, pdwt_info w) { return haar_forward2d<float>(i, c, t, w); }
int pdwt_haar_forward2d_f64(double* i, double** c, double* t, pdwt_info w) { return haar_forward2d<double>(i, c, t, w); }
int pdwt_haar_inverse2d_f32(float* i, float** c, float* t, pdwt_info w) { return haar_inverse2d<float>(i, c, t, w); }
int pdwt_haar_inverse2d_f64(double* i, double** c, double* t, pdwt_info w) { return haar_inverse2d<double>(i, c, t, w); }
int pdwt_haar_forward1d_f32(float* i, float** c, float* t, pdwt_info w) { return haar_forward1d<float>(i, c, t, w); }
int pdwt_haar_forward1d_f64(double* i, double** c, double* t, pdwt_info w) { return haar_forward1d<double>(i, c, t, w); }
int pdwt_haar_inverse1d_f32(float* i, float** c, float* t, pdwt_info w) { return haar_inverse1d<float>(i, c, t, w); }
int pdwt_haar_inverse1d_f64(double* i, double** c, double* t, pdwt_info w) { return haar_inverse1d<double>(i, c, t, w); }
}
