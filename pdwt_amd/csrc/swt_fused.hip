// swt_fused.hip -- one level of the 2-D stationary transform (forward) in ONE launch, float32.
//
// The per-level form (swt.hip + cols_ring.hip) runs a row pass that writes two full-size temporaries and a column
// pass that reads them back: 3 + 6 image-sized transfers per level where 1 read + 4 writes are needed.  Here the row
// pass result never leaves the chip:
//   * rows y = rho (mod f), f = 2^(level-1), form f independent sub-images on which the dilated column filter is an
//     ordinary dense one (f divides Nr, so the periodic wrap stays inside the class);
//   * a workgroup owns a 1024-column tile x one residue class x a chunk of that class's rows and walks DOWN it: each
//     input row is staged once in LDS with its (hlen-1)*f halo (16-byte coalesced loads, prefetched one row ahead in
//     registers, two LDS buffers -> one barrier per row), every thread runs the dilated ROW pass for its 4 columns out of
//     LDS (aligned 16-byte reads at stride f) and pushes (lo,hi) into a register ring of hlen rows;
//   * as soon as the ring holds a full window the COLUMN pass emits one row of A,H,V,D (16-byte stores).
// HBM traffic per level: N (+ halo) read, 4N written -- the algorithmic minimum of a level.
// Arithmetic per sample = row pass then column pass, taps ascending, one FMA per tap: bit-identical to the two-pass
// kernels and to the oracle.  Reference code replaced: w_kern_forward_swt_pass1/2 + one iteration of
// w_forward_swt_separable (src/separable.cu:409-516).
#include "swt_fused.hpp"

#include "stream_dev.hpp"

namespace pdwt {

constexpr int kSwtTile = 1024;  // columns per workgroup: 256 threads x 4

__device__ __forceinline__ void swtf_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// FSEL: 1, 2 = tap spacing 1 or 2 (the row-pass window is read as one run of aligned 16-byte chunks),
//       0 = spacing >= 4 (a multiple of 4: every tap is its own aligned 16-byte read)
template <int HLEN, int FSEL>
__global__ __launch_bounds__(256) void k_swt_fwd_fused(const float* __restrict__ in, float* __restrict__ cA, float* __restrict__ cH,
                                                        float* __restrict__ cV, float* __restrict__ cD, int Nr, int Nc, int fct, int M,
                                                        TapsLH f)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* const smem = reinterpret_cast<float*>(smem_raw);
    constexpr int C = HLEN / 2 - 1;
    const int tid = threadIdx.x;
    const int rho = blockIdx.y % fct;
    const int m0 = (blockIdx.y / fct) * M;
    const int Mc = Nr / fct;  // rows of one residue class
    const int nout = min(M, Mc - m0);
    if (nout <= 0) return;
    const int x0 = blockIdx.x * kSwtTile;
    const int HLc = ((C * fct + 3) >> 2) << 2, HRc = (((HLEN - 1 - C) * fct + 3) >> 2) << 2;
    const int PW = HLc + kSwtTile + HRc;  // staged floats per row
    const int nst = PW >> 2;              // 16-byte chunks per staged row (<= 512: checked by the dispatcher)
    const int xs = x0 - HLc;              // global column of LDS index 0 (a multiple of 4, may be negative)
    const int xg = x0 + 4 * tid;
    const bool active = xg < Nc;
    const int nin = nout + HLEN - 1;  // class-local input rows m0-C .. m0+nout-1+(HLEN-1-C)

    // staging: chunk k of a row lives at global columns wrap(xs + 4k): never straddles the wrap (Nc % 4 == 0)
    const int k0 = min(tid, nst - 1), k1 = min(tid + 256, nst - 1);
    const int gc0 = wrapi(xs + 4 * k0, Nc), gc1 = wrapi(xs + 4 * k1, Nc);
    auto row_of = [&](int r) { return (size_t)(rho + fct * wrapi(m0 - C + min(r, nin - 1), Mc)) * Nc; };
    v4f pre0, pre1;
    {
        const float* p = in + row_of(0);
        pre0 = *reinterpret_cast<const v4f*>(p + gc0);
        pre1 = *reinterpret_cast<const v4f*>(p + gc1);
    }

    v2f ring[HLEN][4];  // (lo,hi) of the thread's 4 columns for the last HLEN rows of the class

    for (int rb = 0; rb < nin; rb += HLEN) {
        static_for<HLEN>([&](auto U) {
            constexpr int u = decltype(U)::value;
            const int r = rb + u;
            if (r < nin) {  // uniform
                float* const buf = smem + ((u & 1) ? PW : 0);  // (rb is a multiple of HLEN, which is even: r & 1 == u & 1)
                reinterpret_cast<v4f*>(buf)[k0] = pre0;
                reinterpret_cast<v4f*>(buf)[k1] = pre1;
                {  // next row's chunks fly while this one is transformed (clamped at the end: harmless re-read)
                    const float* p = in + row_of(r + 1);
                    pre0 = *reinterpret_cast<const v4f*>(p + gc0);
                    pre1 = *reinterpret_cast<const v4f*>(p + gc1);
                }
                swtf_barrier();
                // ---- row pass: (lo,hi)[q] = sum_j x[4 tid + q + (j - C) f] * (L,H)[HLEN-1-j] ----
                v2f acc[4];
#pragma unroll
                for (int q = 0; q < 4; q++) acc[q] = v2f{0.f, 0.f};
                if constexpr (FSEL == 0) {
                    int woff = 4 * tid;  // HLc == C*f here; laundered per row (see k_swt_inv_fused)
                    asm("" : "+v"(woff) : "s"(r));
                    const float* w0 = buf + woff;
                    static_for<HLEN>([&](auto J) {
                        constexpr int j = decltype(J)::value;
                        const v4f t = *reinterpret_cast<const v4f*>(w0 + j * fct);
                        const v2f tp = f.t[HLEN - 1 - j];
#pragma unroll
                        for (int q = 0; q < 4; q++) acc[q] = pk_fma(splat(t[q]), tp, acc[q]);
                    });
                } else {
                    constexpr int PAD = (((C * FSEL + 3) >> 2) << 2) - C * FSEL;      // window start inside its first chunk
                    constexpr int NCH = (PAD + 4 + (HLEN - 1) * FSEL + 3) / 4;          // aligned chunks covering the window
                    float w[NCH * 4];
#pragma unroll
                    for (int k = 0; k < NCH; k++) {
                        const v4f t = reinterpret_cast<const v4f*>(buf + 4 * tid)[k];
#pragma unroll
                        for (int q = 0; q < 4; q++) w[4 * k + q] = t[q];
                    }
                    static_for<HLEN>([&](auto J) {
                        constexpr int j = decltype(J)::value;
                        const v2f tp = f.t[HLEN - 1 - j];
#pragma unroll
                        for (int q = 0; q < 4; q++) acc[q] = pk_fma(splat(w[PAD + q + j * FSEL]), tp, acc[q]);
                    });
                }
#pragma unroll
                for (int q = 0; q < 4; q++) ring[u][q] = acc[q];
                // ---- column pass once the window r-HLEN+1 .. r is complete: output row m0 + r - (HLEN-1) of the class ----
                if (r >= HLEN - 1) {
                    v2f ah[4], vd[4];  // (A,H) from lo, (V,D) from hi
#pragma unroll
                    for (int q = 0; q < 4; q++) ah[q] = vd[q] = v2f{0.f, 0.f};
                    static_for<HLEN>([&](auto J) {
                        constexpr int j = decltype(J)::value;
                        constexpr int s = (u + 1 + j) % HLEN;  // oldest row first
                        const v2f tp = f.t[HLEN - 1 - j];
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            ah[q] = pk_fma(splat(ring[s][q].x), tp, ah[q]);
                            vd[q] = pk_fma(splat(ring[s][q].y), tp, vd[q]);
                        }
                    });
                    if (active) {
                        const size_t o = (size_t)(rho + fct * (m0 + r - (HLEN - 1))) * Nc + xg;
                        *reinterpret_cast<v4f*>(cA + o) = v4f{ah[0].x, ah[1].x, ah[2].x, ah[3].x};
                        *reinterpret_cast<v4f*>(cH + o) = v4f{ah[0].y, ah[1].y, ah[2].y, ah[3].y};
                        *reinterpret_cast<v4f*>(cV + o) = v4f{vd[0].x, vd[1].x, vd[2].x, vd[3].x};
                        *reinterpret_cast<v4f*>(cD + o) = v4f{vd[0].y, vd[1].y, vd[2].y, vd[3].y};
                    }
                }
            }
        });
    }
}

// -------------------------------------------------------------------------------------------------
// inverse level in one launch.  The reference synthesises columns first, then rows (src/separable.cu:553-626);
// that order would need the column results at the halo COLUMNS of the tile.  The two 1-D operators commute, so
// here every staged row goes through the ROW synthesis first,
//     u1 = IL_x(A) + IH_x(V),   u2 = IL_x(H) + IH_x(D)        (A,H: row-low-pass branch; V,D: row-high-pass branch)
// (u1,u2) enter the register ring and the COLUMN synthesis out = IL_y(u1) + IH_y(u2) emits one image row per
// input row -- the mirror image of the forward kernel.  Same taps, same products, a different summation
// order than the two-pass kernels: equal to them within a few ulp (tests: 1e-5 relative, like every SWT inverse
// comparison -- the reference itself halves each product where this build halves the taps once).
// -------------------------------------------------------------------------------------------------
constexpr int kSwtTileI = 512;  // inverse: 256 threads x 2 columns (four staged bands: half the registers per thread)

template <int HLEN, int FSEL>
__global__ __launch_bounds__(256) void k_swt_inv_fused(const float* __restrict__ cA, const float* __restrict__ cH, const float* __restrict__ cV,
                                                        const float* __restrict__ cD, float* __restrict__ out, int Nr, int Nc, int fct, int M,
                                                        Taps2<float> f)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* const smem = reinterpret_cast<float*>(smem_raw);
    constexpr int C = HLEN / 2;  // synthesis centre (A-4)
    const int tid = threadIdx.x;
    const int rho = blockIdx.y % fct;
    const int m0 = (blockIdx.y / fct) * M;
    const int Mc = Nr / fct;
    const int nout = min(M, Mc - m0);
    if (nout <= 0) return;
    const int x0 = blockIdx.x * kSwtTileI;
    const int HLc = ((C * fct + 1) >> 1) << 1, HRc = (((HLEN - 1 - C) * fct + 1) >> 1) << 1;  // multiples of 2 (8-byte chunks)
    const int PW = HLc + kSwtTileI + HRc;
    const int nst = PW >> 1;  // 8-byte chunks per staged row (<= 512)
    const int xs = x0 - HLc;
    const int xg = x0 + 2 * tid;
    const bool active = xg < Nc;
    const int nin = nout + HLEN - 1;
    const int k0 = min(tid, nst - 1), k1 = min(tid + 256, nst - 1);
    const int gc0 = wrapi(xs + 2 * k0, Nc), gc1 = wrapi(xs + 2 * k1, Nc);
    auto row_of = [&](int r) { return (size_t)(rho + fct * wrapi(m0 - C + min(r, nin - 1), Mc)) * Nc; };
    const float* const band[4] = {cA, cV, cH, cD};  // staging order: (A,V) feed u1, (H,D) feed u2
    v2f pre[4][2];
    {
        const size_t o = row_of(0);
#pragma unroll
        for (int b = 0; b < 4; b++) {
            pre[b][0] = *reinterpret_cast<const v2f*>(band[b] + o + gc0);
            pre[b][1] = *reinterpret_cast<const v2f*>(band[b] + o + gc1);
        }
    }
    v2f ru1[HLEN], ru2[HLEN];  // ring: (u1, u2) of the thread's 2 columns

    for (int rb = 0; rb < nin; rb += HLEN) {
        static_for<HLEN>([&](auto U) {
            constexpr int u = decltype(U)::value;
            const int r = rb + u;
            if (r < nin) {
                float* const buf = smem + ((u & 1) ? 4 * PW : 0);
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    reinterpret_cast<v2f*>(buf + b * PW)[k0] = pre[b][0];
                    reinterpret_cast<v2f*>(buf + b * PW)[k1] = pre[b][1];
                }
                {
                    const size_t o = row_of(r + 1);
#pragma unroll
                    for (int b = 0; b < 4; b++) {
                        pre[b][0] = *reinterpret_cast<const v2f*>(band[b] + o + gc0);
                        pre[b][1] = *reinterpret_cast<const v2f*>(band[b] + o + gc1);
                    }
                }
                swtf_barrier();
                // ---- row synthesis: s[b] = sum_j band_b[x + (j - C) f] * tap_b[HLEN-1-j], the column pair packed ----
                v2f sacc[4];
                static_for<4>([&](auto B) {
                    constexpr int b = decltype(B)::value;
                    sacc[b] = v2f{0.f, 0.f};
                    // opaque per row: otherwise the 4*HLEN tap addresses (w0 + j*fct, fct a run-time value) are hoisted out
                    // of the row loop as 2 x 56 loop-invariant registers (221 VGPRs).  On the OFFSET, not the pointer:
                    // laundering the pointer loses its LDS address space and the ds_reads become flat loads.
                    // Not volatile (a volatile asm is a barrier for every memory operation around it) but fed the row index, so it
                    // can be neither hoisted nor merged across rows.
                    int woff = b * PW + 2 * tid;
                    asm("" : "+v"(woff) : "s"(r));
                    const float* w0 = buf + woff;
                    if constexpr (FSEL != 1) {  // spacing even: HLc == C*f, every tap an aligned 8-byte read
                        static_for<HLEN>([&](auto J) {
                            constexpr int j = decltype(J)::value;
                            const v2f t = *reinterpret_cast<const v2f*>(w0 + j * fct);
                            sacc[b] = pk_fma(t, splat((b & 1) ? f.b[HLEN - 1 - j] : f.a[HLEN - 1 - j]), sacc[b]);  // A,H with IL; V,D with IH
                        });
                    } else {  // spacing 1: one run of aligned 8-byte chunks covers the window
                        constexpr int PAD = (((C + 1) >> 1) << 1) - C;
                        constexpr int NCH = (PAD + 2 + (HLEN - 1) + 1) / 2;
                        float w[NCH * 2];
#pragma unroll
                        for (int k = 0; k < NCH; k++) {
                            const v2f t = reinterpret_cast<const v2f*>(w0)[k];
                            w[2 * k] = t.x;
                            w[2 * k + 1] = t.y;
                        }
                        static_for<HLEN>([&](auto J) {
                            constexpr int j = decltype(J)::value;
                            sacc[b] = pk_fma(v2f{w[PAD + j], w[PAD + 1 + j]}, splat((b & 1) ? f.b[HLEN - 1 - j] : f.a[HLEN - 1 - j]), sacc[b]);
                        });
                    }
                });
                ru1[u] = sacc[0] + sacc[1];  // band[]: 0 = A (IL), 1 = V (IH), 2 = H (IL), 3 = D (IH)
                ru2[u] = sacc[2] + sacc[3];
                // ---- column synthesis: out = IL_y(u1) + IH_y(u2) over the window r-HLEN+1 .. r ----
                if (r >= HLEN - 1) {
                    v2f o1 = {0.f, 0.f}, o2 = {0.f, 0.f};
                    static_for<HLEN>([&](auto J) {
                        constexpr int j = decltype(J)::value;
                        constexpr int s = (u + 1 + j) % HLEN;
                        o1 = pk_fma(ru1[s], splat(f.a[HLEN - 1 - j]), o1);
                        o2 = pk_fma(ru2[s], splat(f.b[HLEN - 1 - j]), o2);
                    });
                    if (active) *reinterpret_cast<v2f*>(out + (size_t)(rho + fct * (m0 + r - (HLEN - 1))) * Nc + xg) = o1 + o2;
                }
            }
        });
    }
}

// 4 columns per thread (16-byte accesses): half the instructions per sample, twice the ring registers
template <int HLEN, int FSEL>
__global__ __launch_bounds__(256) void k_swt_inv_fused4(const float* __restrict__ cA, const float* __restrict__ cH, const float* __restrict__ cV,
                                                        const float* __restrict__ cD, float* __restrict__ out, int Nr, int Nc, int fct, int M,
                                                        Taps2<float> f)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* const smem = reinterpret_cast<float*>(smem_raw);
    constexpr int C = HLEN / 2;  // synthesis centre (A-4)
    const int tid = threadIdx.x;
    const int rho = blockIdx.y % fct;
    const int m0 = (blockIdx.y / fct) * M;
    const int Mc = Nr / fct;
    const int nout = min(M, Mc - m0);
    if (nout <= 0) return;
    const int x0 = blockIdx.x * kSwtTile;
    const int HLc = ((C * fct + 3) >> 2) << 2, HRc = (((HLEN - 1 - C) * fct + 3) >> 2) << 2;
    const int PW = HLc + kSwtTile + HRc;
    const int nst = PW >> 2;  // 16-byte chunks per staged row (<= 512)
    const int xs = x0 - HLc;
    const int xg = x0 + 4 * tid;
    const bool active = xg < Nc;
    const int nin = nout + HLEN - 1;
    const int k0 = min(tid, nst - 1), k1 = min(tid + 256, nst - 1);
    const int gc0 = wrapi(xs + 4 * k0, Nc), gc1 = wrapi(xs + 4 * k1, Nc);
    auto row_of = [&](int r) { return (size_t)(rho + fct * wrapi(m0 - C + min(r, nin - 1), Mc)) * Nc; };
    const float* const band[4] = {cA, cV, cH, cD};  // staging order: (A,V) feed u1, (H,D) feed u2
    v4f pre[4][2];
    {
        const size_t o = row_of(0);
#pragma unroll
        for (int b = 0; b < 4; b++) {
            pre[b][0] = *reinterpret_cast<const v4f*>(band[b] + o + gc0);
            pre[b][1] = *reinterpret_cast<const v4f*>(band[b] + o + gc1);
        }
    }
    v2f ru1[HLEN][2], ru2[HLEN][2];  // ring: (u1, u2) of the thread's 4 columns (two packed pairs each)

    for (int rb = 0; rb < nin; rb += HLEN) {
        static_for<HLEN>([&](auto U) {
            constexpr int u = decltype(U)::value;
            const int r = rb + u;
            if (r < nin) {
                float* const buf = smem + ((u & 1) ? 4 * PW : 0);
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    reinterpret_cast<v4f*>(buf + b * PW)[k0] = pre[b][0];
                    reinterpret_cast<v4f*>(buf + b * PW)[k1] = pre[b][1];
                }
                {
                    const size_t o = row_of(r + 1);
#pragma unroll
                    for (int b = 0; b < 4; b++) {
                        pre[b][0] = *reinterpret_cast<const v4f*>(band[b] + o + gc0);
                        pre[b][1] = *reinterpret_cast<const v4f*>(band[b] + o + gc1);
                    }
                }
                swtf_barrier();
                // ---- row synthesis: s[b] = sum_j band_b[x + (j - C) f] * tap_b[HLEN-1-j], the column pair packed ----
                v2f sacc[4][2];
                static_for<4>([&](auto B) {
                    constexpr int b = decltype(B)::value;
                    sacc[b][0] = sacc[b][1] = v2f{0.f, 0.f};
                    int woff = b * PW + 4 * tid;  // laundered per row: see k_swt_inv_fused
                    asm("" : "+v"(woff) : "s"(r));
                    const float* w0 = buf + woff;
                    if constexpr (FSEL == 0) {
                        static_for<HLEN>([&](auto J) {
                            constexpr int j = decltype(J)::value;
                            const v4f t = *reinterpret_cast<const v4f*>(w0 + j * fct);
                            const v2f tp = splat((b & 1) ? f.b[HLEN - 1 - j] : f.a[HLEN - 1 - j]);
                            sacc[b][0] = pk_fma(v2f{t[0], t[1]}, tp, sacc[b][0]);
                            sacc[b][1] = pk_fma(v2f{t[2], t[3]}, tp, sacc[b][1]);
                        });
                    } else {
                        constexpr int PAD = (((C * FSEL + 3) >> 2) << 2) - C * FSEL;
                        constexpr int NCH = (PAD + 4 + (HLEN - 1) * FSEL + 3) / 4;
                        float w[NCH * 4];
#pragma unroll
                        for (int k = 0; k < NCH; k++) {
                            const v4f t = reinterpret_cast<const v4f*>(w0)[k];
#pragma unroll
                            for (int q = 0; q < 4; q++) w[4 * k + q] = t[q];
                        }
                        static_for<HLEN>([&](auto J) {
                            constexpr int j = decltype(J)::value;
                            const v2f tp = splat((b & 1) ? f.b[HLEN - 1 - j] : f.a[HLEN - 1 - j]);
                            sacc[b][0] = pk_fma(v2f{w[PAD + j * FSEL], w[PAD + 1 + j * FSEL]}, tp, sacc[b][0]);
                            sacc[b][1] = pk_fma(v2f{w[PAD + 2 + j * FSEL], w[PAD + 3 + j * FSEL]}, tp, sacc[b][1]);
                        });
                    }
                });
                ru1[u][0] = sacc[0][0] + sacc[1][0];  // band[]: 0 = A (IL), 1 = V (IH), 2 = H (IL), 3 = D (IH)
                ru1[u][1] = sacc[0][1] + sacc[1][1];
                ru2[u][0] = sacc[2][0] + sacc[3][0];
                ru2[u][1] = sacc[2][1] + sacc[3][1];
                // ---- column synthesis: out = IL_y(u1) + IH_y(u2) over the window r-HLEN+1 .. r ----
                if (r >= HLEN - 1) {
                    v2f o1[2] = {v2f{0.f, 0.f}, v2f{0.f, 0.f}}, o2[2] = {v2f{0.f, 0.f}, v2f{0.f, 0.f}};
                    static_for<HLEN>([&](auto J) {
                        constexpr int j = decltype(J)::value;
                        constexpr int s = (u + 1 + j) % HLEN;
                        const v2f ta = splat(f.a[HLEN - 1 - j]), tb = splat(f.b[HLEN - 1 - j]);
                        o1[0] = pk_fma(ru1[s][0], ta, o1[0]);
                        o1[1] = pk_fma(ru1[s][1], ta, o1[1]);
                        o2[0] = pk_fma(ru2[s][0], tb, o2[0]);
                        o2[1] = pk_fma(ru2[s][1], tb, o2[1]);
                    });
                    if (active) {
                        const v2f a0 = o1[0] + o2[0], a1 = o1[1] + o2[1];
                        *reinterpret_cast<v4f*>(out + (size_t)(rho + fct * (m0 + r - (HLEN - 1))) * Nc + xg) = v4f{a0.x, a0.y, a1.x, a1.y};
                    }
                }
            }
        });
    }
}

#define PDWT_CHECK_LAUNCH() PDWT_HIP_TRY(hipGetLastError())
#define PDWT_SWTF_HLENS(X) X(2) X(4) X(6) X(8) X(10) X(12) X(14) X(16)

template <int HLEN>
static int launch_swt_fwd(const float* in, float* cA, float* cH, float* cV, float* cD, int Nr, int Nc, int fct, const Taps2<float>& f2)
{
    constexpr int C = HLEN / 2 - 1;
    TapsLH f;
    for (int k = 0; k < PDWT_MAX_FILTER_WIDTH; k++) f.t[k] = v2f{f2.a[k], f2.b[k]};
    const int HLc = ((C * fct + 3) >> 2) << 2, HRc = (((HLEN - 1 - C) * fct + 3) >> 2) << 2;
    const int PW = HLc + kSwtTile + HRc;
    if (PW / 4 > 512 || PW - kSwtTile > Nc) return 1;  // two staging chunks per thread; halo shorter than a row
    const size_t lds = 2 * (size_t)PW * sizeof(float) + 64;  // (+ slack: the last thread's window read is rounded up to 16 bytes)
    const int Mc = Nr / fct;
    // rows of a class per workgroup: ~700 workgroups (2-3 per CU, all resident) measured best at 4096^2 db7 (72-88 us per
    // level at M = 24, 80 at 32, 100 at 64, 140 at 128): shorter chunks pay HLEN-1 warm-up rows each, taller ones leave
    // too few workgroups to overlap the per-row barrier
    const int tiles = idiv_up(Nc, kSwtTile);
    int M = env_int("PDWT_SWTF_M", 0);
    if (M <= 0) {
        M = (int)(((long long)Mc * fct * tiles + 703) / 704);
        if (M < HLEN) M = HLEN;
    }
    if (M > Mc) M = Mc;
    dim3 grid(tiles, fct * idiv_up(Mc, M));
    KTimer kt(K_SWT_ANA_COLS);
    if (fct == 1) hipLaunchKernelGGL((k_swt_fwd_fused<HLEN, 1>), grid, dim3(256), lds, stream(), in, cA, cH, cV, cD, Nr, Nc, fct, M, f);
    else if (fct == 2) hipLaunchKernelGGL((k_swt_fwd_fused<HLEN, 2>), grid, dim3(256), lds, stream(), in, cA, cH, cV, cD, Nr, Nc, fct, M, f);
    else hipLaunchKernelGGL((k_swt_fwd_fused<HLEN, 0>), grid, dim3(256), lds, stream(), in, cA, cH, cV, cD, Nr, Nc, fct, M, f);
    PDWT_CHECK_LAUNCH();
    return PDWT_OK;
}

int swt_fwd_fused_f32(const float* in, float* cA, float* cH, float* cV, float* cD, int Nr, int Nc, int hlen, int fct, const Taps2<float>& f)
{
    if (env_int("PDWT_SWTF", 1) != 1) return 1;
    if ((Nc & 3) || Nc < 64 || (Nr % fct) != 0 || Nr / fct < 2 * hlen) return 1;
    if (!al16(in) || !al16(cA) || !al16(cH) || !al16(cV) || !al16(cD)) return 1;
    if (in == cA || in == cH || in == cV || in == cD) return 1;  // one launch: the input must not be a band being written
    switch (hlen) {
#define X(H) \
    case H: return launch_swt_fwd<H>(in, cA, cH, cV, cD, Nr, Nc, fct, f);
        PDWT_SWTF_HLENS(X)
#undef X
        default: return 1;
    }
}

template <int HLEN>
static int launch_swt_inv4(const float* cA, const float* cH, const float* cV, const float* cD, float* out, int Nr, int Nc, int fct,
                           const Taps2<float>& f)
{
    constexpr int C = HLEN / 2;
    const int HLc = ((C * fct + 3) >> 2) << 2, HRc = (((HLEN - 1 - C) * fct + 3) >> 2) << 2;
    const int PW = HLc + kSwtTile + HRc;
    if (PW / 4 > 512 || PW - kSwtTile > Nc) return 1;
    const size_t lds = 2 * 4 * (size_t)PW * sizeof(float) + 64;
    if (lds > 64 * 1024) return 1;
    const int Mc = Nr / fct;
    const int tiles = idiv_up(Nc, kSwtTile);
    int M = env_int("PDWT_SWTF_MI", 0);
    if (M <= 0) {
        M = (int)(((long long)Mc * fct * tiles + 511) / 512);
        if (M < HLEN) M = HLEN;
    }
    if (M > Mc) M = Mc;
    dim3 grid(tiles, fct * idiv_up(Mc, M));
    KTimer kt(K_SWT_SYN_COLS);
    if (fct == 1) hipLaunchKernelGGL((k_swt_inv_fused4<HLEN, 1>), grid, dim3(256), lds, stream(), cA, cH, cV, cD, out, Nr, Nc, fct, M, f);
    else if (fct == 2) hipLaunchKernelGGL((k_swt_inv_fused4<HLEN, 2>), grid, dim3(256), lds, stream(), cA, cH, cV, cD, out, Nr, Nc, fct, M, f);
    else hipLaunchKernelGGL((k_swt_inv_fused4<HLEN, 0>), grid, dim3(256), lds, stream(), cA, cH, cV, cD, out, Nr, Nc, fct, M, f);
    PDWT_CHECK_LAUNCH();
    return PDWT_OK;
}

template <int HLEN>
static int launch_swt_inv(const float* cA, const float* cH, const float* cV, const float* cD, float* out, int Nr, int Nc, int fct,
                          const Taps2<float>& f)
{
    constexpr int C = HLEN / 2;
    const int HLc = ((C * fct + 1) >> 1) << 1, HRc = (((HLEN - 1 - C) * fct + 1) >> 1) << 1;
    const int PW = HLc + kSwtTileI + HRc;
    if (PW / 2 > 512 || PW - kSwtTileI > Nc) return 1;
    const size_t lds = 2 * 4 * (size_t)PW * sizeof(float) + 64;
    if (lds > 64 * 1024) return 1;
    const int Mc = Nr / fct;
    const int tiles = idiv_up(Nc, kSwtTileI);
    int M = env_int("PDWT_SWTF_MI", 0);
    if (M <= 0) {
        M = (int)(((long long)Mc * fct * tiles + 703) / 704);
        if (M < HLEN) M = HLEN;
    }
    if (M > Mc) M = Mc;
    dim3 grid(tiles, fct * idiv_up(Mc, M));
    KTimer kt(K_SWT_SYN_COLS);
    if (fct == 1) hipLaunchKernelGGL((k_swt_inv_fused<HLEN, 1>), grid, dim3(256), lds, stream(), cA, cH, cV, cD, out, Nr, Nc, fct, M, f);
    else if (fct == 2) hipLaunchKernelGGL((k_swt_inv_fused<HLEN, 2>), grid, dim3(256), lds, stream(), cA, cH, cV, cD, out, Nr, Nc, fct, M, f);
    else hipLaunchKernelGGL((k_swt_inv_fused<HLEN, 0>), grid, dim3(256), lds, stream(), cA, cH, cV, cD, out, Nr, Nc, fct, M, f);
    PDWT_CHECK_LAUNCH();
    return PDWT_OK;
}

int swt_inv_fused_f32(const float* cA, const float* cH, const float* cV, const float* cD, float* out, int Nr, int Nc, int hlen, int fct,
                      const Taps2<float>& f)
{
    if (env_int("PDWT_SWTF", 1) != 1) return 1;
    if ((Nc & 3) || Nc < 64 || (Nr % fct) != 0 || Nr / fct < 2 * hlen) return 1;
    if (fct > 1 && (fct & 1)) return 1;
    if (!al16(out) || !al16(cA) || !al16(cH) || !al16(cV) || !al16(cD)) return 1;
    if (out == cA || out == cH || out == cV || out == cD) return 1;
    switch (hlen) {
#define X(H) \
    case H: {                                                                                                   \
        /* 4 columns per thread measured faster (559 vs 660 us for the 5 levels of 4096^2 db7); 2 columns where its tile does not fit */ \
        const int rc4 = env_int("PDWT_SWTF_PX", 4) == 4 ? launch_swt_inv4<H>(cA, cH, cV, cD, out, Nr, Nc, fct, f) : 1; \
        return rc4 == 1 ? launch_swt_inv<H>(cA, cH, cV, cD, out, Nr, Nc, fct, f) : rc4;                           \
    }
        PDWT_SWTF_HLENS(X)
#undef X
        default: return 1;
    }
}

}  // namespace pdwt
