// dwt.hip -- decimating separable DWT: per-level kernels + level drivers (gfx950 / CDNA4).
//
// Path replaced: reference src/separable.cu:91-395 (w_kern_forward_pass1/2, w_kern_inverse_pass1/2
// and the four level drivers).  Math: SURVEY.md Appendix A-1 / A-2; the per-sample tap order and the
// one-FMA-per-tap accumulation are the reference's, so results are bit-identical to the CPU oracle.
//
// This file holds the GENERAL kernels (any size, any hlen <= 40, float and double) and the level drivers.  The
// drivers try the specialised float32 paths first -- two levels per launch (dwt_casc.hip), one streaming launch
// per level (dwt_stream.hip), all levels of a batched-1D row in one launch (dwt1d_fused.hip), register-ring
// column passes (cols_ring.inc) -- and fall back to the kernels below for geometries those do not take.
//
// MI355X design of the general kernels (not the reference's 16x16-thread, one-global-load-per-tap structure):
//   * one FUSED kernel per level for 2D (row pass + column pass through LDS): every input sample is
//     read from HBM once per level (+ tile halo), the four bands are written once; the reference
//     round-trips two half-width temporaries through memory per level;
//   * x is the lane axis everywhere (64 consecutive columns per wave) so global accesses coalesce
//     into full 256-byte segments; periodic halos are resolved while the tile is staged into LDS;
//   * taps travel by value in the kernarg segment (scalar loads), not in __constant__ memory;
//   * long filters whose fused tile would not leave >= 2 workgroups per CU fall back to two LDS-tiled
//     1-D passes (k_ana_rows + k_ana_cols / k_syn_cols + k_syn_rows), which are also the batched-1D
//     transform (reference: "2D separable transform without the second pass", src/separable.cu:213).
#include <stdlib.h>
#include <string.h>

#include "common.hpp"
#include "cols_ring.hpp"
#include "rows_tr.hpp"
#include "dwt1d_fused.hpp"
#include <new>
#include <vector>

#include "dwt_lds.hpp"
#include "dwt_lat.hpp"
#include "dwt_stream.hpp"
#include "dwt_casc.hpp"
#include "casc_dev.hpp"

namespace pdwt {

constexpr int kThreads = 256;  // 4 x wave64

// -------------------------------------------------------------------------------------------------
// 2D forward level, fused.  Block = TY x TX outputs of each band.
//   stage  (2TY+hlen-2) x (2TX+hlen-2) inputs -> LDS   (wrap: A-1 "virtual replicate then periodic")
//   rows   lo/hi[r][i] = sum_j in[r][2i+j] * L/H[hlen-1-j]
//   cols   A,H = L/H over lo ; V,D = L/H over hi        (reference pass2: src/separable.cu:135-176)
// -------------------------------------------------------------------------------------------------
template <typename T, int HLEN, int TX, int TY>
__global__ __launch_bounds__(kThreads) void k_fwd2d_fused(const T* __restrict__ in, T* __restrict__ cA, T* __restrict__ cH,
                                                           T* __restrict__ cV, T* __restrict__ cD, int Nr, int Nc, int hlen_rt,
                                                           Taps2<T> f)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int hlen = HLEN > 0 ? HLEN : hlen_rt;
    const int c = (hlen & 1) ? hlen / 2 : hlen / 2 - 1;
    const int RIN = 2 * TY + hlen - 2;
    const int CIN = 2 * TX + hlen - 2;
    const int CINP = CIN | 1;  // odd row pitch: the stride-2 row-pass reads spread over all banks
    T* s_in = reinterpret_cast<T*>(smem);
    T* s_lo = s_in + RIN * CINP;
    T* s_hi = s_lo + RIN * TX;

    const int Nr2 = div2(Nr), Nc2 = div2(Nc);
    const int x0 = blockIdx.x * TX, y0 = blockIdx.y * TY;
    const int xb = 2 * x0 - c, yb = 2 * y0 - c;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;

    // stage the input tile (interior tiles: no wrap taken; edge tiles: periodic / replicate rule)
    const bool interior = (xb >= 0) && (xb + CIN <= Nc) && (yb >= 0) && (yb + RIN <= Nr);
    if (interior) {
        for (int r = ty; r < RIN; r += 4) {
            const T* row = in + (size_t)(yb + r) * Nc + xb;
            for (int cc = tx; cc < CIN; cc += 64) s_in[r * CINP + cc] = row[cc];
        }
    } else {
        for (int r = ty; r < RIN; r += 4) {
            const T* row = in + (size_t)wrap_ext(yb + r, Nr) * Nc;
            for (int cc = tx; cc < CIN; cc += 64) s_in[r * CINP + cc] = row[wrap_ext(xb + cc, Nc)];
        }
    }
    __syncthreads();

    // row pass
    for (int r = ty; r < RIN; r += 4) {
        const T* p = s_in + r * CINP + 2 * tx;
        T lo = 0, hi = 0;
#pragma unroll
        for (int j = 0; j < hlen; j++) {
            const T v = p[j];
            lo = fma_t(v, f.a[hlen - 1 - j], lo);
            hi = fma_t(v, f.b[hlen - 1 - j], hi);
        }
        s_lo[r * TX + tx] = lo;
        s_hi[r * TX + tx] = hi;
    }
    __syncthreads();

    // column pass + store
    const int gx = x0 + tx;
    for (int y = ty; y < TY; y += 4) {
        T a = 0, h = 0, v = 0, d = 0;
        const T* pl = s_lo + (2 * y) * TX + tx;
        const T* ph = s_hi + (2 * y) * TX + tx;
#pragma unroll
        for (int j = 0; j < hlen; j++) {
            const T l = pl[j * TX], g = ph[j * TX];
            const T fl = f.a[hlen - 1 - j], fh = f.b[hlen - 1 - j];
            a = fma_t(l, fl, a);
            h = fma_t(l, fh, h);
            v = fma_t(g, fl, v);
            d = fma_t(g, fh, d);
        }
        const int gy = y0 + y;
        if (gy < Nr2 && gx < Nc2) {
            const size_t o = (size_t)gy * Nc2 + gx;
            cA[o] = a;
            cH[o] = h;
            cV[o] = v;
            cD[o] = d;
        }
    }
}

// -------------------------------------------------------------------------------------------------
// 2D inverse level, fused.  Block = (2TY) x (2TX) output samples from a (TY+h2) x (TX+h2) tile of
// each of the four bands.  Column synthesis first (reference pass1, src/separable.cu:246-289), row
// synthesis second (pass2, :293-328); math A-2.
// -------------------------------------------------------------------------------------------------
template <typename T, int HLEN, int TX, int TY>
__global__ __launch_bounds__(kThreads) void k_inv2d_fused(const T* __restrict__ cA, const T* __restrict__ cH, const T* __restrict__ cV,
                                                           const T* __restrict__ cD, T* __restrict__ out, int Nri, int Nci, int Nro,
                                                           int Nco, int hlen_rt, Taps2<T> f)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int hlen = HLEN > 0 ? HLEN : hlen_rt;
    const int h2 = hlen / 2;
    const int c = h2 / 2;
    const int shift = (h2 & 1) ? 0 : 1;
    const int RC = TY + h2;
    const int CC = TX + h2;
    const int CCP = CC | 1;
    T* s_A = reinterpret_cast<T*>(smem);
    T* s_H = s_A + RC * CCP;
    T* s_V = s_H + RC * CCP;
    T* s_D = s_V + RC * CCP;
    T* s_t1 = s_D + RC * CCP;  // (2TY) x CCP
    T* s_t2 = s_t1 + 2 * TY * CCP;

    const int x0 = blockIdx.x * TX, y0 = blockIdx.y * TY;  // coefficient coordinates
    const int xb = x0 - c, yb = y0 - c;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;

    const bool interior = (xb >= 0) && (xb + CC <= Nci) && (yb >= 0) && (yb + RC <= Nri);
    for (int r = ty; r < RC; r += 4) {
        const size_t ro = (size_t)(interior ? yb + r : wrap_per(yb + r, Nri)) * Nci;
        for (int cc = tx; cc < CC; cc += 64) {
            const size_t o = ro + (interior ? xb + cc : wrap_per(xb + cc, Nci));
            const int s = r * CCP + cc;
            s_A[s] = cA[o];
            s_H[s] = cH[o];
            s_V[s] = cV[o];
            s_D[s] = cD[o];
        }
    }
    __syncthreads();

    // column synthesis: t1 = A*IL + H*IH, t2 = V*IL + D*IH for the 2TY output rows of this tile
    for (int gyl = ty; gyl < 2 * TY; gyl += 4) {
        const int gp = gyl + shift;  // 2*y0 is even, so parity/halving can be done tile-locally
        const int pl = gp >> 1, off = 1 - (gp & 1);
        for (int cc = tx; cc < CC; cc += 64) {
            T sa = 0, sh = 0, sv = 0, sd = 0;
            const int base = pl * CCP + cc;
#pragma unroll
            for (int j = 0; j < h2; j++) {
                const int k = hlen - 1 - (2 * j + off);
                const T fl = f.a[k], fh = f.b[k];
                const int s = base + j * CCP;
                sa = fma_t(s_A[s], fl, sa);
                sh = fma_t(s_H[s], fh, sh);
                sv = fma_t(s_V[s], fl, sv);
                sd = fma_t(s_D[s], fh, sd);
            }
            s_t1[gyl * CCP + cc] = sa + sh;
            s_t2[gyl * CCP + cc] = sv + sd;
        }
    }
    __syncthreads();

    // row synthesis + store: out = t1*IL + t2*IH.  One lane = one PAIR of adjacent outputs (2u, 2u+1): the tap
    // parity of each output is then a compile-time/uniform quantity and the taps stay scalar operands (a
    // per-lane parity turns f.a[k] into a divergent private-memory lookup -- the reference's kernel has the
    // same divergence on its constant cache, src/separable.cu:310-326).
    for (int gyl = ty; gyl < 2 * TY; gyl += 4) {
        const int gy = 2 * y0 + gyl;
        if (gy >= Nro) break;
        for (int u = tx; u < TX; u += 64) {
            T res[2];
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const int gp = 2 * u + e + shift;
                const int pl = gp >> 1, off = 1 - ((e + shift) & 1);
                T s1 = 0, s2 = 0;
                const int base = gyl * CCP + pl;
#pragma unroll
                for (int j = 0; j < h2; j++) {
                    const int k = hlen - 1 - (2 * j + off);
                    s1 = fma_t(s_t1[base + j], f.a[k], s1);
                    s2 = fma_t(s_t2[base + j], f.b[k], s2);
                }
                res[e] = s1 + s2;
            }
            const int gx = 2 * x0 + 2 * u;
            if (gx < Nco) out[(size_t)gy * Nco + gx] = res[0];
            if (gx + 1 < Nco) out[(size_t)gy * Nco + gx + 1] = res[1];
        }
    }
}

// -------------------------------------------------------------------------------------------------
// 1-D analysis along rows (batched-1D forward level; row half of the two-pass 2D fallback).
// Block = 4 rows (one per wave) x TXO outputs.  in: Nr x Nc  ->  lo, hi: Nr x ceil(Nc/2).
// -------------------------------------------------------------------------------------------------
template <typename T, int HLEN, int TXO>
__global__ __launch_bounds__(kThreads) void k_ana_rows(const T* __restrict__ in, T* __restrict__ lo, T* __restrict__ hi, int Nr, int Nc,
                                                        int hlen_rt, Taps2<T> f)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int hlen = HLEN > 0 ? HLEN : hlen_rt;
    const int c = (hlen & 1) ? hlen / 2 : hlen / 2 - 1;
    const int CIN = 2 * TXO + hlen - 2;
    const int CINP = CIN | 1;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    T* s = reinterpret_cast<T*>(smem) + w * CINP;
    const int Nc2 = div2(Nc);
    const int x0 = blockIdx.x * TXO;
    const int xb = 2 * x0 - c;
    const int row = blockIdx.y * 4 + w;
    if (row < Nr) {
        const T* src = in + (size_t)row * Nc;
        if (xb >= 0 && xb + CIN <= Nc) {
            for (int cc = lane; cc < CIN; cc += 64) s[cc] = src[xb + cc];
        } else {
            for (int cc = lane; cc < CIN; cc += 64) s[cc] = src[wrap_ext(xb + cc, Nc)];
        }
    }
    __syncthreads();
    if (row >= Nr) return;
    for (int i = lane; i < TXO; i += 64) {
        const int gx = x0 + i;
        if (gx >= Nc2) break;
        const T* p = s + 2 * i;
        T l = 0, h = 0;
#pragma unroll
        for (int j = 0; j < hlen; j++) {
            const T v = p[j];
            l = fma_t(v, f.a[hlen - 1 - j], l);
            h = fma_t(v, f.b[hlen - 1 - j], h);
        }
        lo[(size_t)row * Nc2 + gx] = l;
        hi[(size_t)row * Nc2 + gx] = h;
    }
}

// -------------------------------------------------------------------------------------------------
// 1-D analysis along columns of two inputs (column half of the two-pass 2D fallback).
// t1,t2: Nr x Ncw -> A,H (from t1) and V,D (from t2): ceil(Nr/2) x Ncw.  Block = TYO x 64 outputs.
// -------------------------------------------------------------------------------------------------
template <typename T, int HLEN, int TYO>
__global__ __launch_bounds__(kThreads) void k_ana_cols(const T* __restrict__ t1, const T* __restrict__ t2, T* __restrict__ cA,
                                                        T* __restrict__ cH, T* __restrict__ cV, T* __restrict__ cD, int Nr, int Ncw,
                                                        int hlen_rt, Taps2<T> f)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int hlen = HLEN > 0 ? HLEN : hlen_rt;
    const int c = (hlen & 1) ? hlen / 2 : hlen / 2 - 1;
    const int RIN = 2 * TYO + hlen - 2;
    T* s1 = reinterpret_cast<T*>(smem);
    T* s2 = s1 + RIN * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int Nr2 = div2(Nr);
    const int gx = blockIdx.x * 64 + tx;
    const int y0 = blockIdx.y * TYO;
    const int yb = 2 * y0 - c;
    if (gx < Ncw) {
        for (int r = ty; r < RIN; r += 4) {
            const size_t o = (size_t)wrap_ext(yb + r, Nr) * Ncw + gx;
            s1[r * 64 + tx] = t1[o];
            s2[r * 64 + tx] = t2[o];
        }
    }
    __syncthreads();
    if (gx >= Ncw) return;
    for (int y = ty; y < TYO; y += 4) {
        const int gy = y0 + y;
        if (gy >= Nr2) break;
        T a = 0, h = 0, v = 0, d = 0;
        const T* p1 = s1 + (2 * y) * 64 + tx;
        const T* p2 = s2 + (2 * y) * 64 + tx;
#pragma unroll
        for (int j = 0; j < hlen; j++) {
            const T l = p1[j * 64], g = p2[j * 64];
            const T fl = f.a[hlen - 1 - j], fh = f.b[hlen - 1 - j];
            a = fma_t(l, fl, a);
            h = fma_t(l, fh, h);
            v = fma_t(g, fl, v);
            d = fma_t(g, fh, d);
        }
        const size_t o = (size_t)gy * Ncw + gx;
        cA[o] = a;
        cH[o] = h;
        cV[o] = v;
        cD[o] = d;
    }
}

// -------------------------------------------------------------------------------------------------
// 1-D synthesis along columns: (A,H) -> t1, (V,D) -> t2.  Bands: Nri x Nc, outputs: Nro x Nc.
// Block = 64 columns x TYC coefficient rows (2*TYC output rows).
// -------------------------------------------------------------------------------------------------
template <typename T, int HLEN, int TYC>
__global__ __launch_bounds__(kThreads) void k_syn_cols(const T* __restrict__ cA, const T* __restrict__ cH, const T* __restrict__ cV,
                                                        const T* __restrict__ cD, T* __restrict__ t1, T* __restrict__ t2, int Nri,
                                                        int Nc, int Nro, int hlen_rt, Taps2<T> f)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int hlen = HLEN > 0 ? HLEN : hlen_rt;
    const int h2 = hlen / 2;
    const int c = h2 / 2;
    const int shift = (h2 & 1) ? 0 : 1;
    const int RC = TYC + h2;
    T* s_A = reinterpret_cast<T*>(smem);
    T* s_H = s_A + RC * 64;
    T* s_V = s_H + RC * 64;
    T* s_D = s_V + RC * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int gx = blockIdx.x * 64 + tx;
    const int y0 = blockIdx.y * TYC;
    const int yb = y0 - c;
    if (gx < Nc) {
        for (int r = ty; r < RC; r += 4) {
            const size_t o = (size_t)wrap_per(yb + r, Nri) * Nc + gx;
            const int s = r * 64 + tx;
            s_A[s] = cA[o];
            s_H[s] = cH[o];
            s_V[s] = cV[o];
            s_D[s] = cD[o];
        }
    }
    __syncthreads();
    if (gx >= Nc) return;
    for (int gyl = ty; gyl < 2 * TYC; gyl += 4) {
        const int gy = 2 * y0 + gyl;
        if (gy >= Nro) break;
        const int gp = gyl + shift;
        const int pl = gp >> 1, off = 1 - (gp & 1);
        T sa = 0, sh = 0, sv = 0, sd = 0;
        const int base = pl * 64 + tx;
#pragma unroll
        for (int j = 0; j < h2; j++) {
            const int k = hlen - 1 - (2 * j + off);
            const T fl = f.a[k], fh = f.b[k];
            const int s = base + j * 64;
            sa = fma_t(s_A[s], fl, sa);
            sh = fma_t(s_H[s], fh, sh);
            sv = fma_t(s_V[s], fl, sv);
            sd = fma_t(s_D[s], fh, sd);
        }
        t1[(size_t)gy * Nc + gx] = sa + sh;
        t2[(size_t)gy * Nc + gx] = sv + sd;
    }
}

// -------------------------------------------------------------------------------------------------
// 1-D synthesis along rows: out = a*IL + d*IH.  a,d: Nr x Nci, out: Nr x Nco.
// Block = 4 rows (one per wave) x TXC coefficient columns (2*TXC outputs).
// -------------------------------------------------------------------------------------------------
template <typename T, int HLEN, int TXC>
__global__ __launch_bounds__(kThreads) void k_syn_rows(const T* __restrict__ a, const T* __restrict__ d, T* __restrict__ out, int Nr,
                                                        int Nci, int Nco, int hlen_rt, Taps2<T> f)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int hlen = HLEN > 0 ? HLEN : hlen_rt;
    const int h2 = hlen / 2;
    const int c = h2 / 2;
    const int shift = (h2 & 1) ? 0 : 1;
    const int CC = TXC + h2;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    T* sa = reinterpret_cast<T*>(smem) + w * 2 * CC;
    T* sd = sa + CC;
    const int x0 = blockIdx.x * TXC;
    const int xb = x0 - c;
    const int row = blockIdx.y * 4 + w;
    if (row < Nr) {
        const T* pa = a + (size_t)row * Nci;
        const T* pd = d + (size_t)row * Nci;
        for (int cc = lane; cc < CC; cc += 64) {
            const int sx = wrap_per(xb + cc, Nci);
            sa[cc] = pa[sx];
            sd[cc] = pd[sx];
        }
    }
    __syncthreads();
    if (row >= Nr) return;
    // one lane = one pair of adjacent outputs: uniform tap parity per accumulator (see k_inv2d_fused)
    for (int u = lane; u < TXC; u += 64) {
        const int gx = 2 * x0 + 2 * u;
        if (gx >= Nco) break;
        T res[2];
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const int gp = 2 * u + e + shift;
            const int pl = gp >> 1, off = 1 - ((e + shift) & 1);
            T s1 = 0, s2 = 0;
#pragma unroll
            for (int j = 0; j < h2; j++) {
                const int k = hlen - 1 - (2 * j + off);
                s1 = fma_t(sa[pl + j], f.a[k], s1);
                s2 = fma_t(sd[pl + j], f.b[k], s2);
            }
            res[e] = s1 + s2;
        }
        out[(size_t)row * Nco + gx] = res[0];
        if (gx + 1 < Nco) out[(size_t)row * Nco + gx + 1] = res[1];
    }
}

// =================================================================================================
// host side: launch helpers
// =================================================================================================
template <typename K>
static int set_lds(K kernel, size_t bytes)
{
    if (bytes > 64 * 1024) return lds_opt_in_ptr((const void*)kernel);  // (a driver call the first time only, never on the steady-state enqueue path)
    return PDWT_OK;
}

static bool force_twopass() { return knob(KN_FORCE_TWOPASS) == 1; }
static bool tiled_cols_forced() { return knob(KN_TILED_COLS) == 1; }

constexpr int FTX = 64, FTY = 16;           // fused tile (outputs per band / coefficient tile)
constexpr size_t kFusedLdsBudget = 64 * 1024;  // keep >= 2 workgroups per CU (160 KiB LDS)

template <typename T>
static size_t fwd_fused_lds(int hlen)
{
    const int RIN = 2 * FTY + hlen - 2, CINP = (2 * FTX + hlen - 2) | 1;
    return ((size_t)RIN * CINP + 2 * (size_t)RIN * FTX) * sizeof(T);
}
template <typename T>
static size_t inv_fused_lds(int hlen)
{
    const int h2 = hlen / 2, RC = FTY + h2, CCP = (FTX + h2) | 1;
    return (4 * (size_t)RC * CCP + 2 * (size_t)(2 * FTY) * CCP) * sizeof(T);
}

#define PDWT_CHECK_LAUNCH() PDWT_HIP_TRY(hipGetLastError())

// Filter lengths with a compile-time instantiation of the tiled kernels (tap loops fully unrolled, taps
// in SGPRs instead of one scalar load + wait per tap); any other length runs the HLEN=0 (runtime) form.
#define PDWT_TILED_HLENS(X) X(4) X(6) X(8) X(10) X(12) X(14) X(16) X(18) X(20) X(24) X(30) X(40)

// ---- level launchers --------------------------------------------------------------------------
template <typename T>
static int launch_ana_rows(const T* in, T* lo, T* hi, int Nr, int Nc, int hlen, const Taps2<T>& f)
{
    if constexpr (sizeof(T) == 8) {  // long double-precision banks: register-window + tap-register kernel (rows_tr.hip)
        const int rc = ana_rows_tr_f64(in, lo, hi, Nr, Nc, hlen, f);
        if (rc <= 0) return rc;
    }
    constexpr int TXO = 256;
    const size_t lds = 4 * (size_t)((2 * TXO + hlen - 2) | 1) * sizeof(T);
    dim3 grid(idiv_up(div2(Nc), TXO), idiv_up(Nr, 4));
    void (*k)(const T*, T*, T*, int, int, int, Taps2<T>);
    switch (hlen) {
#define X(H) case H: k = k_ana_rows<T, H, TXO>; break;
        PDWT_TILED_HLENS(X)
#undef X
        default: k = k_ana_rows<T, 0, TXO>; break;
    }
    KTimer kt(K_ANA_ROWS);
    hipLaunchKernelGGL(k, grid, dim3(kThreads), lds, stream(), in, lo, hi, Nr, Nc, hlen, f);
    PDWT_CHECK_LAUNCH();
    return PDWT_OK;
}

template <typename T>
static int launch_syn_rows(const T* a, const T* d, T* out, int Nr, int Nci, int Nco, int hlen, const Taps2<T>& f)
{
    if constexpr (sizeof(T) == 8) {
        const int rc = syn_rows_tr_f64(a, d, out, Nr, Nci, Nco, hlen, f);
        if (rc <= 0) return rc;
    }
    constexpr int TXC = 128;
    const size_t lds = 4 * 2 * (size_t)(TXC + hlen / 2) * sizeof(T);
    dim3 grid(idiv_up(Nci, TXC), idiv_up(Nr, 4));
    void (*k)(const T*, const T*, T*, int, int, int, int, Taps2<T>);
    switch (hlen) {
#define X(H) case H: k = k_syn_rows<T, H, TXC>; break;
        PDWT_TILED_HLENS(X)
#undef X
        default: k = k_syn_rows<T, 0, TXC>; break;
    }
    KTimer kt(K_SYN_ROWS);
    hipLaunchKernelGGL(k, grid, dim3(kThreads), lds, stream(), a, d, out, Nr, Nci, Nco, hlen, f);
    PDWT_CHECK_LAUNCH();
    return PDWT_OK;
}

template <typename T>
static int launch_ana_cols(const T* t1, const T* t2, T* cA, T* cH, T* cV, T* cD, int Nr, int Ncw, int hlen, const Taps2<T>& f)
{
    if (!tiled_cols_forced()) {  // register-ring kernels (cols_ring.inc): one launch per branch
        KTimer kt(K_ANA_COLS);
        const int rc = ana_cols_ring<T>(t1, cA, cH, Nr, Ncw, hlen, f, t2, cV, cD);  // both branches in one launch
        if (rc <= 0) return rc;
    }
    constexpr int TYO = 16;
    const size_t lds = 2 * (size_t)(2 * TYO + hlen - 2) * 64 * sizeof(T);
    void (*k)(const T*, const T*, T*, T*, T*, T*, int, int, int, Taps2<T>);
    switch (hlen) {
#define X(H) case H: k = k_ana_cols<T, H, TYO>; break;
        PDWT_TILED_HLENS(X)
#undef X
        default: k = k_ana_cols<T, 0, TYO>; break;
    }
    if (set_lds(k, lds) != PDWT_OK) return PDWT_EHIP;
    dim3 grid(idiv_up(Ncw, 64), idiv_up(div2(Nr), TYO));
    KTimer kt(K_ANA_COLS);
    hipLaunchKernelGGL(k, grid, dim3(kThreads), lds, stream(), t1, t2, cA, cH, cV, cD, Nr, Ncw, hlen, f);
    PDWT_CHECK_LAUNCH();
    return PDWT_OK;
}

template <typename T>
static int launch_syn_cols(const T* cA, const T* cH, const T* cV, const T* cD, T* t1, T* t2, int Nri, int Nc, int Nro, int hlen,
                           const Taps2<T>& f)
{
    if (!tiled_cols_forced()) {
        KTimer kt(K_SYN_COLS);
        const int rc = syn_cols_ring<T>(cA, cH, t1, Nri, Nc, Nro, hlen, f, cV, cD, t2);  // both branches in one launch
        if (rc <= 0) return rc;
    }
    constexpr int TYC = 16;
    const size_t lds = 4 * (size_t)(TYC + hlen / 2) * 64 * sizeof(T);
    void (*k)(const T*, const T*, const T*, const T*, T*, T*, int, int, int, int, Taps2<T>);
    switch (hlen) {
#define X(H) case H: k = k_syn_cols<T, H, TYC>; break;
        PDWT_TILED_HLENS(X)
#undef X
        default: k = k_syn_cols<T, 0, TYC>; break;
    }
    if (set_lds(k, lds) != PDWT_OK) return PDWT_EHIP;
    dim3 grid(idiv_up(Nc, 64), idiv_up(Nri, TYC));
    KTimer kt(K_SYN_COLS);
    hipLaunchKernelGGL(k, grid, dim3(kThreads), lds, stream(), cA, cH, cV, cD, t1, t2, Nri, Nc, Nro, hlen, f);
    PDWT_CHECK_LAUNCH();
    return PDWT_OK;
}

// one 2D forward level: in (nr x nc) -> A,H,V,D (nr2 x nc2); t1/t2 scratch for the two-pass form
template <typename T>
static int level_fwd2d(const T* in, T* cA, T* cH, T* cV, T* cD, T* t1, T* t2, size_t trash_floats, int nr, int nc, int hlen, const Taps2<T>& f,
                       T* taps_dev)
{
    if constexpr (sizeof(T) == 8) {  // double-precision banks: row + column pass in one launch (dwt_lds.hip)
        if (!force_twopass()) {
            int rc = fwd2d_f64_lat(in, cA, cH, cV, cD, nr, nc, hlen, f);  // orthogonal banks with a lattice table, large even levels (dwt_lat.hip)
            if (rc <= 0) return rc;
            rc = fwd2d_f64_lds(in, cA, cH, cV, cD, taps_dev, nr, nc, hlen, f);
            if (rc <= 0) return rc;
        }
    }
    if constexpr (sizeof(T) == 4) {  // float32 fast path: LDS-free streaming kernel (dwt_stream.hip)
        if (!force_twopass()) {
            // t1 (the two-pass scratch, sized for the FULL image) is unused on this path: it serves as the trash area of
            // the streaming kernels whenever it is big enough
            int rc = fwd2d_stream_f32(in, cA, cH, cV, cD, (float*)t1, trash_floats, nr, nc, hlen, f);
            if (rc <= 0) return rc;
            rc = fwd2d_f32_lds(in, cA, cH, cV, cD, nr, nc, hlen, f);  // long banks: both passes of the level in one launch (dwt_lds.hip)
            if (rc <= 0) return rc;
        }
    }
    const size_t lds = fwd_fused_lds<T>(hlen);
    if (lds <= kFusedLdsBudget && !force_twopass()) {
        dim3 grid(idiv_up(div2(nc), FTX), idiv_up(div2(nr), FTY));
        void (*k)(const T*, T*, T*, T*, T*, int, int, int, Taps2<T>);
        switch (hlen) {
#define X(H) case H: k = k_fwd2d_fused<T, H, FTX, FTY>; break;
            PDWT_TILED_HLENS(X)
#undef X
            default: k = k_fwd2d_fused<T, 0, FTX, FTY>; break;
        }
        KTimer kt(K_FWD2D_FUSED);
        hipLaunchKernelGGL(k, grid, dim3(kThreads), lds, stream(), in, cA, cH, cV, cD, nr, nc, hlen, f);
        PDWT_CHECK_LAUNCH();
        return PDWT_OK;
    }
    int rc = launch_ana_rows(in, t1, t2, nr, nc, hlen, f);
    if (rc != PDWT_OK) return rc;
    return launch_ana_cols(t1, t2, cA, cH, cV, cD, nr, div2(nc), hlen, f);
}

// one 2D inverse level: bands (nri x nci) -> out (nro x nco)
template <typename T>
static int level_inv2d(const T* cA, const T* cH, const T* cV, const T* cD, T* out, T* t1, T* t2, int nri, int nci, int nro, int nco,
                       int hlen, const Taps2<T>& f, T* taps_dev)
{
    if constexpr (sizeof(T) == 8) {  // double-precision banks: column + row synthesis in one launch (dwt_lds.hip)
        if (!force_twopass()) {
            int rc = inv2d_f64_lat(cA, cH, cV, cD, out, nri, nci, nro, nco, hlen, f);
            if (rc <= 0) return rc;
            rc = inv2d_f64_lds(cA, cH, cV, cD, out, taps_dev, nri, nci, nro, nco, hlen, f);
            if (rc <= 0) return rc;
        }
    }
    if constexpr (sizeof(T) == 4) {
        if (!force_twopass()) {
            int rc = inv2d_stream_f32(cA, cH, cV, cD, out, nri, nci, nro, nco, hlen, f);
            if (rc <= 0) return rc;
            rc = inv2d_f32_lds(cA, cH, cV, cD, out, nri, nci, nro, nco, hlen, f);
            if (rc <= 0) return rc;
        }
    }
    const size_t lds = inv_fused_lds<T>(hlen);
    if (lds <= kFusedLdsBudget && !force_twopass()) {
        dim3 grid(idiv_up(nci, FTX), idiv_up(nri, FTY));
        void (*k)(const T*, const T*, const T*, const T*, T*, int, int, int, int, int, Taps2<T>);
        switch (hlen) {
#define X(H) case H: k = k_inv2d_fused<T, H, FTX, FTY>; break;
            PDWT_TILED_HLENS(X)
#undef X
            default: k = k_inv2d_fused<T, 0, FTX, FTY>; break;
        }
        KTimer kt(K_INV2D_FUSED);
        hipLaunchKernelGGL(k, grid, dim3(kThreads), lds, stream(), cA, cH, cV, cD, out, nri, nci, nro, nco, hlen, f);
        PDWT_CHECK_LAUNCH();
        return PDWT_OK;
    }
    int rc = launch_syn_cols(cA, cH, cV, cD, t1, t2, nri, nci, nro, hlen, f);
    if (rc != PDWT_OK) return rc;
    return launch_syn_rows(t1, t2, out, nro, nci, nco, hlen, f);
}

// scratch carving inside d_tmp (>= 2*Nr*Nc + 1024 elements, see pdwt_tmp_elems):
//   [t1: Nr*Nc2][t2: Nr*Nc2][ping0: Nr2*Nc2][ping1: Nr2*Nc2], each start rounded up to 64 elements
template <typename T>
struct Scratch {
    T *t1, *t2, *ping[2];
    size_t trash_floats;  // floats of [t1, ping0): trash area of the streaming single-level kernels (dwt_stream.hpp)
    bool t1_is_trash;  // t1 holds >= kStreamTrashFloats floats (and 16 rows of the image): usable as the streaming kernels' trash area
    Scratch(T* tmp, int Nr, int Nc, int ndims)
    {
        t1_is_trash = (ndims == 2) && ((size_t)Nr * div2(Nc) * sizeof(T) >= kStreamTrashFloats * sizeof(float)) && Nr >= 32;
        auto up = [](size_t n) { return (n + 63) & ~(size_t)63; };
        const size_t half = up((size_t)Nr * div2(Nc));
        // [t1, ping0) = 2*half elements are free on the streaming path: the trash area of the single-level kernels
        trash_floats = (ndims == 2) ? 2 * half * sizeof(T) / sizeof(float) : 0;
        const size_t quarter = up((size_t)(ndims == 2 ? div2(Nr) : Nr) * div2(Nc));
        t1 = tmp;
        t2 = t1 + half;
        if (ndims == 2) {
            ping[0] = t2 + half;
            ping[1] = ping[0] + quarter;
        } else {  // 1D: no t1/t2 needed; two half-size ping buffers
            ping[0] = tmp;
            ping[1] = tmp + quarter;
        }
    }
};

static int check_args(const void* img, const void* coeffs, const void* tmp, const pdwt_info& w, int ndims, bool need_filters, const void* f,
                      int fhlen)
{
    if (!img || !coeffs || !tmp) return PDWT_EINVAL;
    if (w.Nr < 1 || w.Nc < 1 || w.nlevels < 1 || w.nlevels > 32) return PDWT_EINVAL;
    if (w.ndims != ndims) return PDWT_EINVAL;
    if (need_filters) {
        if (!f) return PDWT_EINVAL;
        if (w.hlen < 2 || w.hlen > PDWT_MAX_FILTER_WIDTH || fhlen != w.hlen) return PDWT_EINVAL;
    }
    return PDWT_OK;
}

// ---- drivers ------------------------------------------------------------------------------------
// w_forward_separable, src/separable.cu:179-209.  The approximation ping-pongs between two scratch
// buffers (the fused kernel cannot run in place) and lands in band 0 at the last level.
template <typename T>
static int forward_separable(T* d_image, T** c, T* d_tmp, pdwt_info w, const typename FiltersOf<T>::type* filt)
{
    int rc = check_args(d_image, c, d_tmp, w, 2, true, filt, filt ? filt->hlen : 0);
    if (rc != PDWT_OK) return rc;
    const Taps2<T> f = taps_fwd<T>(filt);
    Scratch<T> s(d_tmp, w.Nr, w.Nc, 2);
    const T* in = d_image;
    int nr = w.Nr, nc = w.Nc;
    int pp = 0;  // scratch buffer the next intermediate approximation goes to (never the one `in` lives in)
    for (int lev = 0; lev < w.nlevels; lev++) {
        if constexpr (sizeof(T) == 4) {
            // two levels in one launch, the approximation between them stays in registers (dwt_casc.hip)
            if (lev + 1 < w.nlevels && !force_twopass()) {
                T* a2 = (lev + 2 == w.nlevels) ? c[0] : s.ping[pp];
                float* trash = s.t1_is_trash ? (float*)s.t1 : nullptr;
                rc = fwd2d_casc_f32(in, c[3 * lev + 1], c[3 * lev + 2], c[3 * lev + 3], a2, c[3 * lev + 4], c[3 * lev + 5], c[3 * lev + 6],
                                    trash, nr, nc, w.hlen, f);
                if (rc < 0) return rc;
                if (rc == PDWT_OK) {
                    in = a2;
                    pp ^= 1;
                    nr = div2(div2(nr));
                    nc = div2(div2(nc));
                    lev++;
                    continue;
                }
            }
        }
        T* aout = (lev == w.nlevels - 1) ? c[0] : s.ping[pp];
        rc = level_fwd2d(in, aout, c[3 * lev + 1], c[3 * lev + 2], c[3 * lev + 3], s.t1, s.t2, s.trash_floats, nr, nc, w.hlen, f,
                         d_tmp + pdwt_tmp_elems(w) - 256);  // (the last 256 elements of d_tmp: tap scratch of the fused f64 kernels)
        if (rc != PDWT_OK) return rc;
        in = aout;
        pp ^= 1;
        nr = div2(nr);
        nc = div2(nc);
    }
    return PDWT_OK;
}

// w_inverse_separable, src/separable.cu:332-364
template <typename T>
static int inverse_separable(T* d_image, T** c, T* d_tmp, pdwt_info w, const typename FiltersOf<T>::type* filt)
{
    int rc = check_args(d_image, c, d_tmp, w, 2, true, filt, filt ? filt->hlen : 0);
    if (rc != PDWT_OK) return rc;
    const Taps2<T> f = taps_inv<T>(filt);
    Scratch<T> s(d_tmp, w.Nr, w.Nc, 2);
    int tNr[34], tNc[34];
    tNr[0] = w.Nr;
    tNc[0] = w.Nc;
    for (int i = 1; i <= w.nlevels; i++) {
        tNr[i] = div2(tNr[i - 1]);
        tNc[i] = div2(tNc[i - 1]);
    }
    const T* a = c[0];
    int top = w.nlevels - 1;
    int pp = (a == s.ping[0]) ? 1 : 0;  // scratch buffer the next intermediate approximation goes to (never the one `a` lives in)
    for (int i = top; i >= 0; i--) {
        if constexpr (sizeof(T) == 4) {
            // levels i and i-1 in one launch, the approximation between them stays in registers (dwt_casc.hip);
            // pairs are (1,0), (3,2), ... so that the finest -- most expensive -- level is always in a pair
            float* trash = s.t1_is_trash ? (float*)s.t1 : nullptr;
            // levels i, i-1 and i-2 in one launch (dwt_casc_invw.hip): the approximation of level i-1 is synthesised from the
            // level-i bands by the waves that need it and never goes to memory
            if (i >= 2 && !(i & 1) && !force_twopass()) {
                T* out = (i == 2) ? d_image : s.ping[pp];
                rc = inv2d_cascw_f32(nullptr, c[3 * i - 2], c[3 * i - 1], c[3 * i], c[3 * i - 5], c[3 * i - 4], c[3 * i - 3], a, c[3 * i + 1],
                                     c[3 * i + 2], c[3 * i + 3], out, trash, tNr[i - 2], tNc[i - 2], w.hlen, f);
                if (rc < 0) return rc;
                if (rc == PDWT_OK) {
                    a = out;
                    pp ^= 1;
                    i -= 2;
                    continue;
                }
            }
            if ((i & 1) && !force_twopass()) {
                T* out = (i == 1) ? d_image : s.ping[pp];
                // workgroup form first (ring hand-off through LDS instead of a recomputed halo per wave)
                rc = inv2d_cascw_f32(a, c[3 * i + 1], c[3 * i + 2], c[3 * i + 3], c[3 * i - 2], c[3 * i - 1], c[3 * i], nullptr, nullptr, nullptr,
                                     nullptr, out, trash, tNr[i - 1], tNc[i - 1], w.hlen, f);
                if (rc < 0) return rc;
                if (rc == PDWT_OK) {
                    a = out;
                    pp ^= 1;
                    i--;
                    continue;
                }
                rc = inv2d_casc_f32(a, c[3 * i + 1], c[3 * i + 2], c[3 * i + 3], c[3 * i - 2], c[3 * i - 1], c[3 * i], out, trash, tNr[i - 1],
                                    tNc[i - 1], w.hlen, f);
                if (rc < 0) return rc;
                if (rc == PDWT_OK) {
                    a = out;
                    pp ^= 1;
                    i--;
                    continue;
                }
            }
        }
        T* out = (i == 0) ? d_image : s.ping[pp];
        rc = level_inv2d(a, c[3 * i + 1], c[3 * i + 2], c[3 * i + 3], out, s.t1, s.t2, tNr[i + 1], tNc[i + 1], tNr[i], tNc[i], w.hlen, f,
                         d_tmp + pdwt_tmp_elems(w) - 256);
        if (rc != PDWT_OK) return rc;
        a = out;
        pp ^= 1;
    }
    return PDWT_OK;
}

// w_forward_separable_1d, src/separable.cu:214-236 (no trailing D2D copy: the last level writes band 0)
template <typename T>
static int forward_separable_1d(T* d_image, T** c, T* d_tmp, pdwt_info w, const typename FiltersOf<T>::type* filt)
{
    int rc = check_args(d_image, c, d_tmp, w, 1, true, filt, filt ? filt->hlen : 0);
    if (rc != PDWT_OK) return rc;
    const Taps2<T> f = taps_fwd<T>(filt);
    if (!force_twopass()) {  // all levels in one launch, row resident in LDS (dwt1d_fused.hip)
        rc = fwd1d_fused<T>(d_image, c, w, f);
        if (rc <= 0) return rc;
    }
    Scratch<T> s(d_tmp, w.Nr, w.Nc, 1);
    const T* in = d_image;
    int nc = w.Nc;
    for (int lev = 0; lev < w.nlevels; lev++) {
        T* aout = (lev == w.nlevels - 1) ? c[0] : s.ping[lev & 1];
        rc = launch_ana_rows(in, aout, c[lev + 1], w.Nr, nc, w.hlen, f);
        if (rc != PDWT_OK) return rc;
        in = aout;
        nc = div2(nc);
    }
    return PDWT_OK;
}

// w_inverse_separable_1d, src/separable.cu:368-395
template <typename T>
static int inverse_separable_1d(T* d_image, T** c, T* d_tmp, pdwt_info w, const typename FiltersOf<T>::type* filt)
{
    int rc = check_args(d_image, c, d_tmp, w, 1, true, filt, filt ? filt->hlen : 0);
    if (rc != PDWT_OK) return rc;
    const Taps2<T> f = taps_inv<T>(filt);
    if (!force_twopass()) {
        rc = inv1d_fused<T>(d_image, c, w, f);
        if (rc <= 0) return rc;
    }
    Scratch<T> s(d_tmp, w.Nr, w.Nc, 1);
    int tNc[34];
    tNc[0] = w.Nc;
    for (int i = 1; i <= w.nlevels; i++) tNc[i] = div2(tNc[i - 1]);
    const T* a = c[0];
    for (int i = w.nlevels - 1; i >= 0; i--) {
        T* out = (i == 0) ? d_image : s.ping[i & 1];
        rc = launch_syn_rows(a, (const T*)c[i + 1], out, w.Nr, tNc[i + 1], tNc[i], w.hlen, f);
        if (rc != PDWT_OK) return rc;
        a = out;
    }
    return PDWT_OK;
}


// =================================================================================================
// batched 2-D: ONE launch per level over B equally sized float32 images (pdwt_batch2d_*)
// =================================================================================================
// Below ~1024^2 a multi-level pair is launch-bound: three dependent launches per direction of a few microseconds each, 24-32 us
// per pair whatever the size.  A batch of such images (BASELINE.json: "batched 1D/2D images") runs every level of ALL images in
// one launch of the streaming level kernels (gridDim.y = image, dwt_stream.hip), the per-image pointers in device-side tables
// built once: 2 L launches per batch and direction instead of 2 L B.  Same kernels, same arithmetic: results are those of the
// per-image transforms bit for bit.  The reference has no batched entry (its TODO.txt:15 lists multi-GPU / batching as future work).
static inline bool b2_al16(const void* p) { return ((uintptr_t)p & 15) == 0; }
// (every batch object starts with its kind: 0 = float32 streaming / cascade levels, 1 = double-precision fused levels, 2 = Haar)
template <typename T> void* haar_batch_create(int nimg, T* const* d_images, T** const* d_coeffs, T* const* d_tmps, pdwt_info w);  // haar.hip
template <typename T> int haar_batch_forward(void* batch);
template <typename T> int haar_batch_inverse(void* batch);
template <typename T> void haar_batch_destroy(void* batch);
void* swt_batch_create_f32(int nimg, float* const* d_images, float** const* d_coeffs, float* const* d_tmps, pdwt_info w);  // swt.hip
int swt_batch_forward_f32(void* batch, const pdwt_filters_f32* filt);
int swt_batch_inverse_f32(void* batch, const pdwt_filters_f32* filt);
void swt_batch_destroy_f32(void* batch);
struct BatchHaar {  // (kind 2: Haar; kind 3: float32 SWT -- hb is the object of haar.hip / swt.hip; elem = sizeof of the sample type it was built for)
    int kind, dev, elem;
    void* hb;
};
static void* batch_swt_create(int nimg, float* const* d_images, float** const* d_coeffs, float* const* d_tmps, pdwt_info w)
{
    BatchHaar* B = new (std::nothrow) BatchHaar();
    if (!B) return nullptr;
    B->kind = 3;
    B->elem = 4;
    B->hb = (hipGetDevice(&B->dev) == hipSuccess) ? swt_batch_create_f32(nimg, d_images, d_coeffs, d_tmps, w) : nullptr;
    if (!B->hb) {
        delete B;
        return nullptr;
    }
    return B;
}
template <typename T>
static void* batch_haar_create(int nimg, T* const* d_images, T** const* d_coeffs, T* const* d_tmps, pdwt_info w)
{
    BatchHaar* B = new (std::nothrow) BatchHaar();
    if (!B) return nullptr;
    B->kind = 2;
    B->elem = (int)sizeof(T);
    B->hb = (hipGetDevice(&B->dev) == hipSuccess) ? haar_batch_create<T>(nimg, d_images, d_coeffs, d_tmps, w) : nullptr;
    if (!B->hb) {
        delete B;
        return nullptr;
    }
    return B;
}
struct Batch2D {
    int kind;
    int nimg, dev;
    pdwt_info w;
    size_t trash_floats;
    int lev_nr[33], lev_nc[33];
    StreamBatchF* d_fwd;  // [level][image]
    StreamBatchI* d_inv;  // [level][image]
    // the two finest levels (forward) / the three finest (inverse; two when the transform has two) through the cascade kernels, all
    // images in one launch (gridDim.y = image): per-image pointers of those launches, and image 0's for the launchers' checks
    CascBatchF* d_cf;
    CascBatchI* d_ci;
    CascBatchF cf0;
    CascBatchI ci0;
    float* trash0;
};

static Batch2D* batch2d_create(int nimg, float* const* d_images, float** const* d_coeffs, float* const* d_tmps, pdwt_info w, int hlen)
{
    if (nimg < 1 || nimg > 65535 || !d_images || !d_coeffs || !d_tmps || w.ndims != 2 || w.do_swt || w.nlevels < 1 || w.nlevels > 32) return nullptr;
    if (force_twopass()) return nullptr;
    // (images of 2048^2 and more take the cascade kernels for their finest levels, see batch2d_forward / batch2d_inverse; the tables of
    // the per-level kernels are built for every level all the same: they are the fallback when a cascade launcher declines)
    // every level must be inside the streaming path, in both directions
    int nr = w.Nr, nc = w.Nc;
    for (int lev = 0; lev < w.nlevels; lev++) {
        if (!fwd2d_stream_takes(nr, nc, hlen) || !inv2d_stream_takes(nr / 2, nc / 2, hlen)) return nullptr;
        nr /= 2;
        nc /= 2;
    }
    Scratch<float> probe(d_tmps[0], w.Nr, w.Nc, 2);
    if (probe.trash_floats < 1024) return nullptr;
    Batch2D* B = new (std::nothrow) Batch2D();
    if (!B) return nullptr;
    B->kind = 0;
    B->nimg = nimg;
    B->w = w;
    B->trash_floats = probe.trash_floats;
    B->d_fwd = nullptr;
    B->d_inv = nullptr;
    B->d_cf = nullptr;
    B->d_ci = nullptr;
    B->trash0 = nullptr;
    if (hipGetDevice(&B->dev) != hipSuccess) {
        delete B;
        return nullptr;
    }
    const int L = w.nlevels;
    std::vector<StreamBatchF> hf((size_t)L * nimg);
    std::vector<StreamBatchI> hi((size_t)L * nimg);
    for (int b = 0; b < nimg; b++) {
        float* const* c = d_coeffs[b];
        Scratch<float> s(d_tmps[b], w.Nr, w.Nc, 2);
        if (!d_images[b] || !c || !d_tmps[b] || !b2_al16(d_images[b])) {
            delete B;
            return nullptr;
        }
        // forward: the approximation ping-pongs between the two scratch buffers and lands in band 0 (forward_separable's level loop)
        const float* in = d_images[b];
        int pp = 0;
        for (int lev = 0; lev < L; lev++) {
            float* aout = (lev == L - 1) ? c[0] : s.ping[pp];
            hf[(size_t)lev * nimg + b] = StreamBatchF{in, aout, c[3 * lev + 1], c[3 * lev + 2], c[3 * lev + 3], (float*)s.t1};
            if (!b2_al16(aout) || !b2_al16(c[3 * lev + 1]) || !b2_al16(c[3 * lev + 2]) || !b2_al16(c[3 * lev + 3])) {
                delete B;
                return nullptr;
            }
            in = aout;
            pp ^= 1;
        }
        // inverse: coarse to fine (inverse_separable's level loop)
        const float* a = c[0];
        pp = 0;
        for (int i = L - 1; i >= 0; i--) {
            float* out = (i == 0) ? d_images[b] : s.ping[pp];
            hi[(size_t)i * nimg + b] = StreamBatchI{a, c[3 * i + 1], c[3 * i + 2], c[3 * i + 3], out};
            a = out;
            pp ^= 1;
        }
    }
    nr = w.Nr;
    nc = w.Nc;
    for (int lev = 0; lev <= L; lev++) {
        B->lev_nr[lev] = nr;
        B->lev_nc[lev] = nc;
        nr /= 2;
        nc /= 2;
    }
    const size_t bf = hf.size() * sizeof(StreamBatchF), bi = hi.size() * sizeof(StreamBatchI);
    B->d_fwd = (StreamBatchF*)pdwt_malloc(bf);
    B->d_inv = (StreamBatchI*)pdwt_malloc(bi);
    bool ok = B->d_fwd && B->d_inv && pdwt_memcpy_h2d(B->d_fwd, hf.data(), bf) == PDWT_OK && pdwt_memcpy_h2d(B->d_inv, hi.data(), bi) == PDWT_OK;
    // cascade tables (two levels and more): the forward pair (0, 1) writes its A2 where level 1 of the per-level chain does, the inverse
    // reads the approximation that enters level 2 (three levels in one launch) or level 1 (a two-level transform)
    if (ok && L >= 2) {
        std::vector<CascBatchF> hcf((size_t)nimg);
        std::vector<CascBatchI> hci((size_t)nimg);
        for (int b = 0; b < nimg; b++) {
            float* const* c = d_coeffs[b];
            hcf[b] = CascBatchF{d_images[b], CascBands{c[1], c[2], c[3], hf[(size_t)1 * nimg + b].cA, c[4], c[5], c[6]}};
            hci[b].b = CascInvBands{L == 2 ? hi[(size_t)1 * nimg + b].cA : nullptr, c[4], c[5], c[6], c[1], c[2], c[3]};
            hci[b].b3 = (L >= 3) ? CascInv3B{hi[(size_t)2 * nimg + b].cA, c[7], c[8], c[9]} : CascInv3B{nullptr, nullptr, nullptr, nullptr};
            hci[b].out = d_images[b];
        }
        B->cf0 = hcf[0];
        B->ci0 = hci[0];
        Scratch<float> s0(d_tmps[0], w.Nr, w.Nc, 2);
        B->trash0 = s0.t1_is_trash ? (float*)s0.t1 : nullptr;
        const size_t cf = hcf.size() * sizeof(CascBatchF), ci = hci.size() * sizeof(CascBatchI);
        B->d_cf = (CascBatchF*)pdwt_malloc(cf);
        B->d_ci = (CascBatchI*)pdwt_malloc(ci);
        ok = B->d_cf && B->d_ci && pdwt_memcpy_h2d(B->d_cf, hcf.data(), cf) == PDWT_OK && pdwt_memcpy_h2d(B->d_ci, hci.data(), ci) == PDWT_OK;
    }
    if (!ok) {
        pdwt_free(B->d_fwd);
        pdwt_free(B->d_inv);
        pdwt_free(B->d_cf);
        pdwt_free(B->d_ci);
        delete B;
        return nullptr;
    }
    return B;
}

// the launches of a batch run on the device its images live on, whatever device is current (Wavelets methods switch to their own
// device the same way; WaveletsBatch drives several devices from one thread)
struct Batch2DDev {
    int prev = -1;
    bool switched = false;
    explicit Batch2DDev(int dev)
    {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) switched = (hipSetDevice(dev) == hipSuccess);
    }
    ~Batch2DDev()
    {
        if (switched) (void)hipSetDevice(prev);
    }
};

static int batch2d_forward(Batch2D* B, const pdwt_filters_f32* filt)
{
    if (!B || !filt || filt->hlen != B->w.hlen) return PDWT_EINVAL;
    Batch2DDev on_dev(B->dev);
    const Taps2<float> f = taps_fwd<float>(filt);
    int lev0 = 0;
    if (B->d_cf && B->trash0) {
        // levels 0 and 1 of every image in one cascade launch (gridDim.y = image); 1 = geometry outside that path
        const CascBands& b = B->cf0.b;
        const int rc = fwd2d_casc_f32(B->cf0.in, b.H1, b.V1, b.D1, b.A2, b.H2, b.V2, b.D2, B->trash0, B->lev_nr[0], B->lev_nc[0], B->w.hlen, f,
                                      B->d_cf, B->nimg);
        if (rc < 0) return rc;
        if (rc == PDWT_OK) lev0 = 2;
    }
    for (int lev = lev0; lev < B->w.nlevels; lev++) {
        const int rc = fwd2d_stream_batch_f32(B->d_fwd + (size_t)lev * B->nimg, B->nimg, B->trash_floats, B->lev_nr[lev], B->lev_nc[lev], B->w.hlen, f);
        if (rc != PDWT_OK) return rc < 0 ? rc : PDWT_EINVAL;
    }
    return PDWT_OK;
}

static int batch2d_inverse(Batch2D* B, const pdwt_filters_f32* filt)
{
    if (!B || !filt || filt->hlen != B->w.hlen) return PDWT_EINVAL;
    Batch2DDev on_dev(B->dev);
    const Taps2<float> f = taps_inv<float>(filt);
    // the finest three levels (two for a two-level transform) of every image in one cascade launch, when that path takes the geometry;
    // decided BEFORE the coarser levels run (they must not be run twice), by the same test the launcher applies
    const int L = B->w.nlevels;
    const int ncasc = (L >= 3) ? 3 : 2;
    bool casc = B->d_ci && B->trash0 && L >= 2 && knob(KN_CASC) == 1 && knob(KN_CASC_IWG) != 1 && knob(KN_CASC_L3) == 1 && stream_enabled() &&
                !((long long)knob(KN_CASC_MIN) > (long long)B->lev_nr[0] * B->lev_nc[0]);
    int next = L - 1;  // next level the per-level kernels would run
    for (; next >= (casc ? ncasc : 0); next--) {
        const int rc = inv2d_stream_batch_f32(B->d_inv + (size_t)next * B->nimg, B->nimg, B->lev_nr[next + 1], B->lev_nc[next + 1], B->w.hlen, f);
        if (rc != PDWT_OK) return rc < 0 ? rc : PDWT_EINVAL;
    }
    if (casc) {
        const CascInvBands& b = B->ci0.b;
        const CascInv3B& b3 = B->ci0.b3;
        const int rc = inv2d_casc3_f32(b.A2, b.H2, b.V2, b.D2, b.H1, b.V1, b.D1, b3.A3, b3.H3, b3.V3, b3.D3, B->ci0.out, B->trash0, B->lev_nr[0],
                                       B->lev_nc[0], B->w.hlen, f, B->d_ci, B->nimg);
        if (rc < 0) return rc;
        if (rc == PDWT_OK) return PDWT_OK;
    }
    for (; next >= 0; next--) {  // (the cascade launcher declined: the per-level kernels finish the job)
        const int rc = inv2d_stream_batch_f32(B->d_inv + (size_t)next * B->nimg, B->nimg, B->lev_nr[next + 1], B->lev_nc[next + 1], B->w.hlen, f);
        if (rc != PDWT_OK) return rc < 0 ? rc : PDWT_EINVAL;
    }
    return PDWT_OK;
}

// -------------------------------------------------------------------------------------------------
// the same entry in double precision (round 5): every level of all images in one launch of the fused level kernels of dwt_lds.hip
// (gridDim.y = image; five pointers per image and level in device-side tables built once).  Any even bank of up to 40 taps, any size
// those kernels take (odd sizes included); the small levels -- which the single-image path hands to the latency-bound two-pass kernels --
// run batched as well.  Same kernels, same arithmetic: bit-identical to the per-image transforms.
// -------------------------------------------------------------------------------------------------
struct Batch2D64 {
    int kind;
    int nimg, dev;
    pdwt_info w;
    int lev_nr[34], lev_nc[34];
    unsigned long long* d_fwd;  // [level][image][5]
    unsigned long long* d_inv;  // [level][image][5]
};

static Batch2D64* batch2d_create_f64(int nimg, double* const* d_images, double** const* d_coeffs, double* const* d_tmps, pdwt_info w)
{
    if (nimg < 1 || nimg > 65535 || !d_images || !d_coeffs || !d_tmps || w.ndims != 2 || w.do_swt || w.nlevels < 1 || w.nlevels > 32) return nullptr;
    if (force_twopass() || w.hlen < 2 || w.hlen > 40 || (w.hlen & 1) || knob(KN_F64_LDS) < 1) return nullptr;
    Batch2D64* B = new (std::nothrow) Batch2D64();
    if (!B) return nullptr;
    B->kind = 1;
    B->nimg = nimg;
    B->w = w;
    B->d_fwd = B->d_inv = nullptr;
    if (hipGetDevice(&B->dev) != hipSuccess) {
        delete B;
        return nullptr;
    }
    const int L = w.nlevels;
    int nr = w.Nr, nc = w.Nc;
    for (int lev = 0; lev <= L; lev++) {
        B->lev_nr[lev] = nr;
        B->lev_nc[lev] = nc;
        // what fwd2d_lds_any asks of a level's input (nr, nc) AND what inv2d_lds_any asks of its coefficient size (div2(nr) rows >= the
        // padded bank length -- the coarsest level included --, div2(nc) >= 2): a handle whose forward runs but whose inverse is refused
        // would break the "NULL otherwise" contract of the header
        const int hp = ((w.hlen + 7) / 8) * 8;
        if (lev < L && (nr < 16 || nr < hp || nc < hp || div2(nr) < hp || div2(nc) < 2)) {
            delete B;
            return nullptr;
        }
        nr = div2(nr);
        nc = div2(nc);
    }
    std::vector<unsigned long long> hf((size_t)L * nimg * 5), hi((size_t)L * nimg * 5);
    for (int b = 0; b < nimg; b++) {
        double* const* c = d_coeffs[b];
        if (!d_images[b] || !c || !d_tmps[b]) {
            delete B;
            return nullptr;
        }
        Scratch<double> sc(d_tmps[b], w.Nr, w.Nc, 2);
        const double* in = d_images[b];
        int pp = 0;
        for (int lev = 0; lev < L; lev++) {  // forward_separable's level loop
            double* aout = (lev == L - 1) ? c[0] : sc.ping[pp];
            unsigned long long* e = &hf[((size_t)lev * nimg + b) * 5];
            e[0] = (unsigned long long)(uintptr_t)in;
            e[1] = (unsigned long long)(uintptr_t)aout;
            e[2] = (unsigned long long)(uintptr_t)c[3 * lev + 1];
            e[3] = (unsigned long long)(uintptr_t)c[3 * lev + 2];
            e[4] = (unsigned long long)(uintptr_t)c[3 * lev + 3];
            if (!aout || !c[3 * lev + 1] || !c[3 * lev + 2] || !c[3 * lev + 3]) {
                delete B;
                return nullptr;
            }
            in = aout;
            pp ^= 1;
        }
        const double* a = c[0];
        pp = 0;
        for (int i = L - 1; i >= 0; i--) {  // inverse_separable's level loop
            double* out = (i == 0) ? d_images[b] : sc.ping[pp];
            unsigned long long* e = &hi[((size_t)i * nimg + b) * 5];
            e[0] = (unsigned long long)(uintptr_t)a;
            e[1] = (unsigned long long)(uintptr_t)c[3 * i + 1];
            e[2] = (unsigned long long)(uintptr_t)c[3 * i + 2];
            e[3] = (unsigned long long)(uintptr_t)c[3 * i + 3];
            e[4] = (unsigned long long)(uintptr_t)out;
            a = out;
            pp ^= 1;
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

static int batch2d_forward_f64(Batch2D64* B, const pdwt_filters_f64* filt)
{
    if (!B || !filt || filt->hlen != B->w.hlen) return PDWT_EINVAL;
    Batch2DDev on_dev(B->dev);
    const Taps2<double> f = taps_fwd<double>(filt);
    for (int lev = 0; lev < B->w.nlevels; lev++) {
        const int rc = fwd2d_f64_lds_batch(B->d_fwd + (size_t)lev * B->nimg * 5, B->nimg, B->lev_nr[lev], B->lev_nc[lev], B->w.hlen, f);
        if (rc != PDWT_OK) return rc < 0 ? rc : PDWT_EINVAL;
    }
    return PDWT_OK;
}
static int batch2d_inverse_f64(Batch2D64* B, const pdwt_filters_f64* filt)
{
    if (!B || !filt || filt->hlen != B->w.hlen) return PDWT_EINVAL;
    Batch2DDev on_dev(B->dev);
    const Taps2<double> f = taps_inv<double>(filt);
    for (int i = B->w.nlevels - 1; i >= 0; i--) {
        const int rc = inv2d_f64_lds_batch(B->d_inv + (size_t)i * B->nimg * 5, B->nimg, B->lev_nr[i + 1], B->lev_nc[i + 1], B->lev_nr[i], B->lev_nc[i], B->w.hlen, f);
        if (rc != PDWT_OK) return rc < 0 ? rc : PDWT_EINVAL;
    }
    return PDWT_OK;
}

}  // namespace pdwt

// hlen == 2 MEANS Haar in this entry (as in the class, which sends every 2-tap transform to pdwt_haar_*, src/wt.cu:236-271): the Haar
// kernels take no bank, so a 2-tap bank that is not Haar's must not come back as PDWT_OK with Haar coefficients
template <typename F>
static bool bank_is_haar(const F* f)
{
    if (!f || f->hlen != 2) return false;
    const double r = 0.70710678118654752440, tol = 1e-6;
    auto near = [&](double a, double b) { return a - b <= tol && b - a <= tol; };
    return near(f->L[0], r) && near(f->L[1], r) && near(f->H[0], -r) && near(f->H[1], r) && near(f->IL[0], r) && near(f->IL[1], r) && near(f->IH[0], r) && near(f->IH[1], -r);
}

using namespace pdwt;

extern "C" {
// (a Haar bank -- hlen 2, the transform the class sends to pdwt_haar_* -- batches through the Haar kernels, either precision)
static bool batch_is_haar(const pdwt_info& w) { return w.hlen == 2 && w.ndims == 2 && !w.do_swt; }
void* pdwt_batch2d_create_f64(int nimg, double* const* d_images, double** const* d_coeffs, double* const* d_tmps, pdwt_info info)
{
    if (batch_is_haar(info)) return batch_haar_create<double>(nimg, d_images, d_coeffs, d_tmps, info);
    return batch2d_create_f64(nimg, d_images, d_coeffs, d_tmps, info);
}
// (a handle is only good for the precision it was created in: every entry checks the kind -- and, for the Haar object, the sample size --
//  before it casts; the other precision's handle is PDWT_EINVAL, not a reinterpretation)
static int batch_kind(const void* batch, int elem)
{
    if (!batch) return -1;
    const int k = *(const int*)batch;
    if (k == 0 || k == 3) return elem == 4 ? k : -1;
    if (k == 1) return elem == 8 ? k : -1;
    if (k == 2) return ((const BatchHaar*)batch)->elem == elem ? k : -1;
    return -1;
}
int pdwt_batch2d_forward_f64(void* batch, const pdwt_filters_f64* f)
{
    const int k = batch_kind(batch, 8);
    if (k == 2) {
        if (!bank_is_haar(f)) return PDWT_EINVAL;
        Batch2DDev on_dev(((BatchHaar*)batch)->dev);
        return haar_batch_forward<double>(((BatchHaar*)batch)->hb);
    }
    return k == 1 ? batch2d_forward_f64((Batch2D64*)batch, f) : PDWT_EINVAL;
}
int pdwt_batch2d_inverse_f64(void* batch, const pdwt_filters_f64* f)
{
    const int k = batch_kind(batch, 8);
    if (k == 2) {
        if (!bank_is_haar(f)) return PDWT_EINVAL;
        Batch2DDev on_dev(((BatchHaar*)batch)->dev);
        return haar_batch_inverse<double>(((BatchHaar*)batch)->hb);
    }
    return k == 1 ? batch2d_inverse_f64((Batch2D64*)batch, f) : PDWT_EINVAL;
}
void* pdwt_batch2d_create_f32(int nimg, float* const* d_images, float** const* d_coeffs, float* const* d_tmps, pdwt_info info)
{
    if (batch_is_haar(info)) return batch_haar_create<float>(nimg, d_images, d_coeffs, d_tmps, info);
    if (info.do_swt) return batch_swt_create(nimg, d_images, d_coeffs, d_tmps, info);
    return batch2d_create(nimg, d_images, d_coeffs, d_tmps, info, info.hlen);
}
int pdwt_batch2d_forward_f32(void* batch, const pdwt_filters_f32* f)
{
    const int k = batch_kind(batch, 4);
    if (k == 2 || k == 3) {
        if (k == 2 && !bank_is_haar(f)) return PDWT_EINVAL;
        Batch2DDev on_dev(((BatchHaar*)batch)->dev);
        return k == 2 ? haar_batch_forward<float>(((BatchHaar*)batch)->hb) : swt_batch_forward_f32(((BatchHaar*)batch)->hb, f);
    }
    return k == 0 ? batch2d_forward((Batch2D*)batch, f) : PDWT_EINVAL;
}
int pdwt_batch2d_inverse_f32(void* batch, const pdwt_filters_f32* f)
{
    const int k = batch_kind(batch, 4);
    if (k == 2 || k == 3) {
        if (k == 2 && !bank_is_haar(f)) return PDWT_EINVAL;
        Batch2DDev on_dev(((BatchHaar*)batch)->dev);
        return k == 2 ? haar_batch_inverse<float>(((BatchHaar*)batch)->hb) : swt_batch_inverse_f32(((BatchHaar*)batch)->hb, f);
    }
    return k == 0 ? batch2d_inverse((Batch2D*)batch, f) : PDWT_EINVAL;
}
// (one destructor for every kind: pdwt_batch2d_destroy and pdwt_batch2d_destroy_f64 are the same function under two names)
void pdwt_batch2d_destroy(void* batch)
{
    if (!batch) return;
    const int k = *(const int*)batch;
    if (k == 2 || k == 3) {
        BatchHaar* H = (BatchHaar*)batch;
        if (k == 3)
            swt_batch_destroy_f32(H->hb);
        else if (H->elem == 8)
            haar_batch_destroy<double>(H->hb);
        else
            haar_batch_destroy<float>(H->hb);
        delete H;
    } else if (k == 1) {
        Batch2D64* B = (Batch2D64*)batch;
        pdwt_free(B->d_fwd);
        pdwt_free(B->d_inv);
        delete B;
    } else if (k == 0) {
        Batch2D* B = (Batch2D*)batch;
        pdwt_free(B->d_fwd);
        pdwt_free(B->d_inv);
        pdwt_free(B->d_cf);
        pdwt_free(B->d_ci);
        delete B;
    }
}
void pdwt_batch2d_destroy_f64(void* batch) { pdwt_batch2d_destroy(batch); }
int pdwt_debug_set(const char* key, int value) { return pdwt::knob_set(key, value); }
int pdwt_debug_get(const char* key, int* value) { return pdwt::knob_get(key, value); }
size_t pdwt_tmp_elems(pdwt_info w) { return 2 * (size_t)(w.Nr > 0 ? w.Nr : 0) * (size_t)(w.Nc > 0 ? w.Nc : 0) + 1024; }
int pdwt_forward_separable_f32(float* i, float** c, float* t, pdwt_info w, const pdwt_filters_f32* f) { return forward_separable<float>(i, c, t, w, f); }
int pdwt_forward_separable_f64(double* i, double** c, double* t, pdwt_info w, const pdwt_filters_f64* f) { return forward_separable<double>(i, c, t, w, f); }
int pdwt_inverse_separable_f32(float* i, float** c, float* t, pdwt_info w, const pdwt_filters_f32* f) { return inverse_separable<float>(i, c, t, w, f); }
int pdwt_inverse_separable_f64(double* i, double** c, double* t, pdwt_info w, const pdwt_filters_f64* f) { return inverse_separable<double>(i, c, t, w, f); }
int pdwt_forward_separable_1d_f32(float* i, float** c, float* t, pdwt_info w, const pdwt_filters_f32* f) { return forward_separable_1d<float>(i, c, t, w, f); }
int pdwt_forward_separable_1d_f64(double* i, double** c, double* t, pdwt_info w, const pdwt_filters_f64* f) { return forward_separable_1d<double>(i, c, t, w, f); }
int pdwt_inverse_separable_1d_f32(float* i, float** c, float* t, pdwt_info w, const pdwt_filters_f32* f) { return inverse_separable_1d<float>(i, c, t, w, f); }
int pdwt_inverse_separable_1d_f64(double* i, double** c, double* t, pdwt_info w, const pdwt_filters_f64* f) { return inverse_separable_1d<double>(i, c, t, w, f); }
}
