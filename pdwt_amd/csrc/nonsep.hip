// nonsep.hip -- 2-D transform with four genuinely NON-SEPARABLE hlen x hlen kernels (custom filter banks only).
//
// Path replaced: reference src/nonseparable.cu (w_kern_forward / w_kern_inverse / w_kern_forward_swt /
// w_kern_inverse_swt and their drivers, :114-452).  For the named wavelets the reference's four kernels are outer
// products of the 1-D bank; that case never reaches this file -- the class runs the separable kernels on a band table
// with H and V exchanged (wt.cpp: nonsep_table), O(hlen) instead of O(hlen^2) per sample.  What is left is the
// `set_filters_forward(name, len, f1, f2, f3, f4)` case with arbitrary kernels (src/wt.cu:560-602): O(hlen^2) per sample by definition.
//
// Two forms of every kernel:
//   * LDS-tiled (k_ns_*_t, the default): a workgroup of 32 x 8 threads stages its input tile (periodic / odd-size extension applied
//     once per tile element, not once per tap) and the four kernels in LDS -- the kernels interleaved as [tap][band], so that one
//     broadcast ds_read_b128 brings the four band taps of a tap position -- and every thread register-blocks 4 x 2 outputs (forward:
//     x = tx + 32 q, so lanes read consecutive LDS words and store consecutive outputs; the decimated forward keeps even and odd input
//     columns in separate LDS rows for the same reason) or 2 x 2 quads of the four output parities (decimated inverse: one sample of the
//     four bands feeds 16 FMAs).  One LDS read per 4-16 FMAs instead of one global load + four scalar loads per 4 FMAs.
//   * plain (k_ns_forward / k_ns_inverse / k_ns_swt): one thread per output, taps through the scalar cache; any hlen, any dilation --
//     the fallback when a tile does not fit LDS (long kernels at high SWT levels), knob `nonsep_tiled` = 0, and the bit-for-bit
//     comparison of tests/.
// Accumulation order (jy outer, jx inner, one FMA per tap and band) is the reference's in both forms, so both are bit-identical to
// each other and to the oracle restatement.
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


// -------------------------------------------------------------------------------------------------
// LDS-tiled forms
// -------------------------------------------------------------------------------------------------
constexpr int kNtTX = 32, kNtTY = 8;  // threads of a workgroup: 32 along x, 8 along y (a wave = 2 rows of 32 lanes)
template <typename T> struct NsV4 { typedef T type __attribute__((ext_vector_type(4))); };

template <typename T>
__device__ __forceinline__ void ns_stage_taps(T* tapL, const T* __restrict__ K, int hh)
{  // K is band-major (4 kernels of hh taps); LDS copy is [tap][band]
    for (int i = threadIdx.y * kNtTX + threadIdx.x; i < 4 * hh; i += kNtTX * kNtTY) {
        const int band = i / hh, k = i - band * hh;
        tapL[k * 4 + band] = K[i];
    }
}
__host__ __device__ constexpr int ns_taps_elems(int hlen) { return (4 * hlen * hlen + 3) & ~3; }

// decimated forward: 128 x (8 QY) outputs per workgroup; tile rows hold the even input columns, then the odd ones
__host__ __device__ constexpr int ns_fwd_pw(int hlen) { return kNtTX * 4 + (hlen + 1) / 2; }
__host__ __device__ constexpr int ns_fwd_rows(int hlen, int qy) { return 2 * (kNtTY * qy - 1) + hlen; }
template <typename T, int QY>
__global__ __launch_bounds__(kNtTX* kNtTY) void k_ns_forward_t(const T* __restrict__ img, T* __restrict__ cA, T* __restrict__ cH,
                                                               T* __restrict__ cV, T* __restrict__ cD, int Nr, int Nc, int hlen,
                                                               const T* __restrict__ K)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using V4 = typename NsV4<T>::type;
    constexpr int QX = 4, TXO = kNtTX * QX, TYO = kNtTY * QY;
    const int Nr2 = div2(Nr), Nc2 = div2(Nc);
    const int c = (hlen & 1) ? hlen / 2 : hlen / 2 - 1;
    const int PW = ns_fwd_pw(hlen), R = ns_fwd_rows(hlen, QY);
    T* const tapL = reinterpret_cast<T*>(smem);
    T* const tile = tapL + ns_taps_elems(hlen);
    const int X0 = blockIdx.x * TXO, Y0 = blockIdx.y * TYO;
    const int tid = threadIdx.y * kNtTX + threadIdx.x;
    ns_stage_taps(tapL, K, hlen * hlen);
    for (int rr = tid >> 6; rr < R; rr += 4) {
        const T* srow = img + (size_t)wrap_ext(2 * Y0 - c + rr, Nr) * Nc;
        T* trow = tile + (size_t)rr * 2 * PW;
        for (int cc = tid & 63; cc < 2 * PW; cc += 64) trow[(cc & 1) * PW + (cc >> 1)] = srow[wrap_ext(2 * X0 - c + cc, Nc)];
    }
    __syncthreads();
    V4 acc[QY][QX];
#pragma unroll
    for (int r = 0; r < QY; r++)
#pragma unroll
        for (int q = 0; q < QX; q++) acc[r][q] = V4{0, 0, 0, 0};
    for (int jy = 0; jy < hlen; jy++) {
        const T* tk = tapL + (size_t)((hlen - 1 - jy) * hlen + (hlen - 1)) * 4;  // tap (jy, jx = 0); jx + 1 is the tap before it
        const T* trow0 = tile + (size_t)(2 * threadIdx.y + jy) * 2 * PW + threadIdx.x;
        for (int jx = 0; jx < hlen; jx++) {
            const V4 t = *reinterpret_cast<const V4*>(tk - 4 * jx);
            const T* tp = trow0 + (jx & 1) * PW + (jx >> 1);
#pragma unroll
            for (int r = 0; r < QY; r++)
#pragma unroll
                for (int q = 0; q < QX; q++) {
                    const T v = tp[(size_t)r * (2 * kNtTY) * 2 * PW + kNtTX * q];
                    acc[r][q] = __builtin_elementwise_fma(V4{v, v, v, v}, t, acc[r][q]);
                }
        }
    }
#pragma unroll
    for (int r = 0; r < QY; r++) {
        const int gy = Y0 + threadIdx.y + kNtTY * r;
        if (gy >= Nr2) continue;
#pragma unroll
        for (int q = 0; q < QX; q++) {
            const int gx = X0 + threadIdx.x + kNtTX * q;
            if (gx >= Nc2) continue;
            const size_t o = (size_t)gy * Nc2 + gx;
            cA[o] = acc[r][q][0];
            cH[o] = acc[r][q][1];
            cV[o] = acc[r][q][2];
            cD[o] = acc[r][q][3];
        }
    }
}

// decimated inverse: a thread owns QX x QY "quads" -- the four outputs (g = 2 m + p, p in {0,1}^2) that read the same band samples
// m - c + j; 64 x (8 QY) quads per workgroup; the four bands interleaved in LDS ([sample][band])
__host__ __device__ constexpr int ns_inv_tw(int hlen) { return kNtTX * 2 + hlen / 2 - 1; }
__host__ __device__ constexpr int ns_inv_th(int hlen, int qy) { return kNtTY * qy + hlen / 2 - 1; }
template <typename T, int QY>
__global__ __launch_bounds__(kNtTX* kNtTY) void k_ns_inverse_t(T* __restrict__ out, const T* __restrict__ cA, const T* __restrict__ cH,
                                                               const T* __restrict__ cV, const T* __restrict__ cD, int Nr, int Nc, int Nro,
                                                               int Nco, int hlen, const T* __restrict__ K)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using V4 = typename NsV4<T>::type;
    constexpr int QX = 2, TXQ = kNtTX * QX, TYQ = kNtTY * QY;
    const int h2 = hlen / 2, c = h2 / 2, shift = (h2 & 1) ? 0 : 1;
    const int TW = ns_inv_tw(hlen), TH = ns_inv_th(hlen, QY);
    T* const tapL = reinterpret_cast<T*>(smem);
    V4* const tile = reinterpret_cast<V4*>(tapL + ns_taps_elems(hlen));
    const int MX0 = blockIdx.x * TXQ, MY0 = blockIdx.y * TYQ;
    const int tid = threadIdx.y * kNtTX + threadIdx.x;
    ns_stage_taps(tapL, K, hlen * hlen);
    for (int rr = tid >> 6; rr < TH; rr += 4) {
        const size_t rowo = (size_t)wrap_per(MY0 - c + rr, Nr) * Nc;
        for (int cc = tid & 63; cc < TW; cc += 64) {
            const size_t o = rowo + wrap_per(MX0 - c + cc, Nc);
            tile[rr * TW + cc] = V4{cA[o], cH[o], cV[o], cD[o]};
        }
    }
    __syncthreads();
    V4 acc[QY][QX][2][2];
#pragma unroll
    for (int r = 0; r < QY; r++)
#pragma unroll
        for (int q = 0; q < QX; q++)
#pragma unroll
            for (int p = 0; p < 4; p++) acc[r][q][p >> 1][p & 1] = V4{0, 0, 0, 0};
    const V4* const tapV = reinterpret_cast<const V4*>(tapL);
    for (int jy = 0; jy < h2; jy++) {
        for (int jx = 0; jx < h2; jx++) {
            V4 t[2][2];  // [py][px]: off = 1 - p
#pragma unroll
            for (int py = 0; py < 2; py++)
#pragma unroll
                for (int px = 0; px < 2; px++) t[py][px] = tapV[(hlen - 1 - (2 * jy + 1 - py)) * hlen + (hlen - 1 - (2 * jx + 1 - px))];
#pragma unroll
            for (int r = 0; r < QY; r++)
#pragma unroll
                for (int q = 0; q < QX; q++) {
                    const V4 s = tile[(threadIdx.y + kNtTY * r + jy) * TW + threadIdx.x + kNtTX * q + jx];
#pragma unroll
                    for (int py = 0; py < 2; py++)
#pragma unroll
                        for (int px = 0; px < 2; px++) acc[r][q][py][px] = __builtin_elementwise_fma(s, t[py][px], acc[r][q][py][px]);
                }
        }
    }
#pragma unroll
    for (int r = 0; r < QY; r++)
#pragma unroll
        for (int q = 0; q < QX; q++) {
            const int my = MY0 + threadIdx.y + kNtTY * r, mx = MX0 + threadIdx.x + kNtTX * q;
#pragma unroll
            for (int py = 0; py < 2; py++) {
                const int oy = 2 * my + py - shift;
                if (oy < 0 || oy >= Nro) continue;
#pragma unroll
                for (int px = 0; px < 2; px++) {
                    const int ox = 2 * mx + px - shift;
                    if (ox < 0 || ox >= Nco) continue;
                    const V4 a = acc[r][q][py][px];
                    out[(size_t)oy * Nco + ox] = a[0] + a[1] + a[2] + a[3];
                }
            }
        }
}

// a-trous levels: 128 x (8 QY) outputs per workgroup, tile = outputs + fac (hlen - 1) in both directions; the inverse keeps the four
// bands interleaved ([sample][band]) and adds c k / 4 per tap as the plain kernel does
__host__ __device__ constexpr int ns_swt_tw(int hlen, int fac) { return kNtTX * 4 + fac * (hlen - 1); }
__host__ __device__ constexpr int ns_swt_th(int hlen, int fac, int qy) { return kNtTY * qy + fac * (hlen - 1); }
template <typename T, bool INV, int QY>
__global__ __launch_bounds__(kNtTX* kNtTY) void k_ns_swt_t(const T* __restrict__ inA, const T* __restrict__ inH, const T* __restrict__ inV,
                                                           const T* __restrict__ inD, T* __restrict__ oA, T* __restrict__ oH,
                                                           T* __restrict__ oV, T* __restrict__ oD, int Nr, int Nc, int hlen, int fac,
                                                           const T* __restrict__ K)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using V4 = typename NsV4<T>::type;
    using S = typename std::conditional<INV, V4, T>::type;  // what a tile element is
    constexpr int QX = 4, TXO = kNtTX * QX, TYO = kNtTY * QY;
    const int c = (INV ? hlen / 2 : ((hlen & 1) ? hlen / 2 : hlen / 2 - 1)) * fac;
    const int TW = ns_swt_tw(hlen, fac), TH = ns_swt_th(hlen, fac, QY);
    T* const tapL = reinterpret_cast<T*>(smem);
    S* const tile = reinterpret_cast<S*>(tapL + ns_taps_elems(hlen));
    const int X0 = blockIdx.x * TXO, Y0 = blockIdx.y * TYO;
    const int tid = threadIdx.y * kNtTX + threadIdx.x;
    ns_stage_taps(tapL, K, hlen * hlen);
    for (int rr = tid >> 6; rr < TH; rr += 4) {
        const size_t rowo = (size_t)wrap_per(Y0 - c + rr, Nr) * Nc;
        for (int cc = tid & 63; cc < TW; cc += 64) {
            const size_t o = rowo + wrap_per(X0 - c + cc, Nc);
            if constexpr (INV)
                tile[rr * TW + cc] = V4{inA[o], inH[o], inV[o], inD[o]};
            else
                tile[rr * TW + cc] = inA[o];
        }
    }
    __syncthreads();
    V4 acc[QY][QX];
#pragma unroll
    for (int r = 0; r < QY; r++)
#pragma unroll
        for (int q = 0; q < QX; q++) acc[r][q] = V4{0, 0, 0, 0};
    const V4* const tapV = reinterpret_cast<const V4*>(tapL);
    for (int jy = 0; jy < hlen; jy++) {
        const S* trow0 = tile + (size_t)(threadIdx.y + fac * jy) * TW + threadIdx.x;
        for (int jx = 0; jx < hlen; jx++) {
            const V4 t = tapV[(hlen - 1 - jy) * hlen + (hlen - 1 - jx)];
            const S* tp = trow0 + fac * jx;
#pragma unroll
            for (int r = 0; r < QY; r++)
#pragma unroll
                for (int q = 0; q < QX; q++) {
                    const S v = tp[(size_t)r * kNtTY * TW + kNtTX * q];
                    if constexpr (INV)
                        acc[r][q] += v * t / T(4);  // (the product is rounded before it is quartered and added, src/nonseparable.cu:386-389)
                    else
                        acc[r][q] = __builtin_elementwise_fma(V4{v, v, v, v}, t, acc[r][q]);
                }
        }
    }
#pragma unroll
    for (int r = 0; r < QY; r++) {
        const int gy = Y0 + threadIdx.y + kNtTY * r;
        if (gy >= Nr) continue;
#pragma unroll
        for (int q = 0; q < QX; q++) {
            const int gx = X0 + threadIdx.x + kNtTX * q;
            if (gx >= Nc) continue;
            const size_t o = (size_t)gy * Nc + gx;
            const V4 a = acc[r][q];
            if constexpr (INV) {
                oA[o] = a[0] + a[1] + a[2] + a[3];
            } else {
                oA[o] = a[0];
                oH[o] = a[1];
                oV[o] = a[2];
                oD[o] = a[3];
            }
        }
    }
}

// which tiled form a level takes: two output rows (quad rows) per thread if its LDS footprint leaves two workgroups per CU, else one,
// else (up to one workgroup per CU) two, one; 0 = the plain kernel.  A level too small to give every CU a tile or two (nx x ny outputs
// -- quads for the decimated inverse -- in tiles of tw x 8 qy) keeps the plain kernel: measured, a 512^2 image is 20 % slower tiled.
static int ns_pick_qy(size_t bytes_qy2, size_t bytes_qy1, int nx, int ny, int tw)
{
    if (knob(KN_NONSEP_TILED) == 0) return 0;
    const size_t two_wg = 80 * 1024, one_wg = 156 * 1024;
    const long long t2 = (long long)idiv_up(nx, tw) * idiv_up(ny, 2 * kNtTY), t1 = (long long)idiv_up(nx, tw) * idiv_up(ny, kNtTY);
    const bool force = knob(KN_NONSEP_TILED) == 2;  // (tests: every size through the tiled kernels)
    const bool ok2 = force || t2 >= 512, ok1 = force || t1 >= 256;
    if (ok2 && bytes_qy2 <= two_wg) return 2;
    if (ok1 && bytes_qy1 <= two_wg) return 1;
    if (ok2 && bytes_qy2 <= one_wg) return 2;
    if (ok1 && bytes_qy1 <= one_wg) return 1;
    return 0;
}
template <typename T> static size_t ns_fwd_lds(int hlen, int qy) { return sizeof(T) * ((size_t)ns_taps_elems(hlen) + (size_t)ns_fwd_rows(hlen, qy) * 2 * ns_fwd_pw(hlen)); }
template <typename T> static size_t ns_inv_lds(int hlen, int qy) { return sizeof(T) * ((size_t)ns_taps_elems(hlen) + (size_t)4 * ns_inv_th(hlen, qy) * ns_inv_tw(hlen)); }
template <typename T> static size_t ns_swt_lds(int hlen, int fac, int qy, bool inv)
{
    return sizeof(T) * ((size_t)ns_taps_elems(hlen) + (size_t)(inv ? 4 : 1) * ns_swt_th(hlen, fac, qy) * ns_swt_tw(hlen, fac));
}
#define PDWT_NS_LAUNCH_T(KERN, GRID, LDS, ...)                                              \
    do {                                                                                    \
        if ((LDS) > 64 * 1024) {                                                            \
            const int rc_ = lds_opt_in<KERN>();                                             \
            if (rc_ != PDWT_OK) return rc_;                                                 \
        }                                                                                   \
        hipLaunchKernelGGL(KERN, GRID, dim3(kNtTX, kNtTY), LDS, stream(), __VA_ARGS__);     \
    } while (0)

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
        const int qy = ns_pick_qy(ns_fwd_lds<T>(w.hlen, 2), ns_fwd_lds<T>(w.hlen, 1), div2(nc), div2(nr), kNtTX * 4);
        const dim3 gt(idiv_up(div2(nc), kNtTX * 4), idiv_up(div2(nr), kNtTY * (qy ? qy : 1)));
        if (qy == 2)
            PDWT_NS_LAUNCH_T((k_ns_forward_t<T, 2>), gt, ns_fwd_lds<T>(w.hlen, 2), in, aout, c[3 * i + 1], c[3 * i + 2], c[3 * i + 3], nr, nc, w.hlen, K);
        else if (qy == 1)
            PDWT_NS_LAUNCH_T((k_ns_forward_t<T, 1>), gt, ns_fwd_lds<T>(w.hlen, 1), in, aout, c[3 * i + 1], c[3 * i + 2], c[3 * i + 3], nr, nc, w.hlen, K);
        else
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
        const int sh = ((w.hlen / 2) & 1) ? 0 : 1;  // quads along a direction: g = o + shift runs up to N - 1 + shift
        const int mqx = (tNc[i] - 1 + sh) / 2 + 1, mqy = (tNr[i] - 1 + sh) / 2 + 1;
        const int qy = (w.hlen >= 2) ? ns_pick_qy(ns_inv_lds<T>(w.hlen, 2), ns_inv_lds<T>(w.hlen, 1), mqx, mqy, kNtTX * 2) : 0;  // (a 1-tap kernel sums over nothing: plain form)
        const dim3 gt(idiv_up(mqx, kNtTX * 2), idiv_up(mqy, kNtTY * (qy ? qy : 1)));
        if (qy == 2)
            PDWT_NS_LAUNCH_T((k_ns_inverse_t<T, 2>), gt, ns_inv_lds<T>(w.hlen, 2), out, a, (const T*)c[3 * i + 1], (const T*)c[3 * i + 2],
                             (const T*)c[3 * i + 3], tNr[i + 1], tNc[i + 1], tNr[i], tNc[i], w.hlen, K);
        else if (qy == 1)
            PDWT_NS_LAUNCH_T((k_ns_inverse_t<T, 1>), gt, ns_inv_lds<T>(w.hlen, 1), out, a, (const T*)c[3 * i + 1], (const T*)c[3 * i + 2],
                             (const T*)c[3 * i + 3], tNr[i + 1], tNc[i + 1], tNr[i], tNc[i], w.hlen, K);
        else
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
        const int fac = 1 << i;
        const int qy = ns_pick_qy(ns_swt_lds<T>(w.hlen, fac, 2, false), ns_swt_lds<T>(w.hlen, fac, 1, false), w.Nc, w.Nr, kNtTX * 4);
        const dim3 gt(idiv_up(w.Nc, kNtTX * 4), idiv_up(w.Nr, kNtTY * (qy ? qy : 1)));
        if (qy == 2)
            PDWT_NS_LAUNCH_T((k_ns_swt_t<T, false, 2>), gt, ns_swt_lds<T>(w.hlen, fac, 2, false), in, (const T*)nullptr, (const T*)nullptr, (const T*)nullptr,
                             aout, c[3 * i + 1], c[3 * i + 2], c[3 * i + 3], w.Nr, w.Nc, w.hlen, fac, K);
        else if (qy == 1)
            PDWT_NS_LAUNCH_T((k_ns_swt_t<T, false, 1>), gt, ns_swt_lds<T>(w.hlen, fac, 1, false), in, (const T*)nullptr, (const T*)nullptr, (const T*)nullptr,
                             aout, c[3 * i + 1], c[3 * i + 2], c[3 * i + 3], w.Nr, w.Nc, w.hlen, fac, K);
        else
            hipLaunchKernelGGL((k_ns_swt<T, false>), ns_grid(w.Nc, w.Nr), dim3(kNsTX, kNsTY), 0, stream(), in, (const T*)nullptr, (const T*)nullptr,
                               (const T*)nullptr, aout, c[3 * i + 1], c[3 * i + 2], c[3 * i + 3], w.Nr, w.Nc, w.hlen, fac, K);
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
        const int fac = 1 << i;
        const int qy = ns_pick_qy(ns_swt_lds<T>(w.hlen, fac, 2, true), ns_swt_lds<T>(w.hlen, fac, 1, true), w.Nc, w.Nr, kNtTX * 4);
        const dim3 gt(idiv_up(w.Nc, kNtTX * 4), idiv_up(w.Nr, kNtTY * (qy ? qy : 1)));
        if (qy == 2)
            PDWT_NS_LAUNCH_T((k_ns_swt_t<T, true, 2>), gt, ns_swt_lds<T>(w.hlen, fac, 2, true), a, (const T*)c[3 * i + 1], (const T*)c[3 * i + 2],
                             (const T*)c[3 * i + 3], out, (T*)nullptr, (T*)nullptr, (T*)nullptr, w.Nr, w.Nc, w.hlen, fac, K);
        else if (qy == 1)
            PDWT_NS_LAUNCH_T((k_ns_swt_t<T, true, 1>), gt, ns_swt_lds<T>(w.hlen, fac, 1, true), a, (const T*)c[3 * i + 1], (const T*)c[3 * i + 2],
                             (const T*)c[3 * i + 3], out, (T*)nullptr, (T*)nullptr, (T*)nullptr, w.Nr, w.Nc, w.hlen, fac, K);
        else
            hipLaunchKernelGGL((k_ns_swt<T, true>), ns_grid(w.Nc, w.Nr), dim3(kNsTX, kNsTY), 0, stream(), a, (const T*)c[3 * i + 1],
                               (const T*)c[3 * i + 2], (const T*)c[3 * i + 3], out, (T*)nullptr, (T*)nullptr, (T*)nullptr, w.Nr, w.Nc, w.hlen, fac, K);
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
