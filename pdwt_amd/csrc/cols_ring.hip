// cols_ring.hip -- column passes of the two-pass 2D transform as LDS-free register-ring kernels (gfx950).
//
// Used where the fused level kernels do not apply: long filters and double precision (config C5: db20, f64).
// Reference code replaced: w_kern_forward_pass2 (src/separable.cu:135-176) and w_kern_inverse_pass1
// (src/separable.cu:246-289).
//
// One lane owns CPL adjacent columns and walks DOWN them: a ring of HLEN+PF raw input rows (analysis) or of
// HLEN/2+PF coefficient rows of each band (synthesis) lives in registers with compile-time slot indices, every
// row is loaded from HBM exactly once per chunk (coalesced: 64 lanes x CPL x sizeof(T) contiguous bytes), the
// PF extra slots are the prefetch distance (loads for rows the next outputs need are issued as soon as a slot
// dies), and each loaded value feeds HLEN (analysis: 2 filters x HLEN/2 outputs) FMAs straight from registers:
// no LDS traffic at all, where the tiled kernels spent one ds_read per FMA.
// Loads are branch-free (row index clamped) so hipcc neither predicates nor waits on them early.
// Same tap order / one FMA per tap as the tiled kernels and the CPU oracle: bit-identical results.
#include <stdlib.h>

#include <type_traits>

#include "common.hpp"
#include "cols_ring.hpp"
#include "tapreg.hpp"

namespace pdwt {

template <int I, int N, typename F>
__device__ __forceinline__ void cfor_impl(F&& fn)
{
    if constexpr (I < N) {
        fn(std::integral_constant<int, I>{});
        cfor_impl<I + 1, N>(fn);
    }
}
template <int N, typename F>
__device__ __forceinline__ void cfor(F&& fn) { cfor_impl<0, N>(fn); }

template <typename T, int N> struct VecT { typedef T type __attribute__((ext_vector_type(N))); };
template <typename T> struct VecT<T, 1> { typedef T type; };
template <typename T, int N> __device__ __forceinline__ T vget(const typename VecT<T, N>::type& v, int i)
{
    if constexpr (N == 1) return v;
    else return v[i];
}
template <typename T, int N> __device__ __forceinline__ void vset(typename VecT<T, N>::type& v, int i, T x)
{
    if constexpr (N == 1) v = x;
    else v[i] = x;
}

constexpr int kRingPF = 8;  // prefetch distance in rows (extra ring slots)

// -------------------------------------------------------------------------------------------------
// analysis along columns, decimating: t (Nr x Ncw) -> lo, hi (ceil(Nr/2) x Ncw).   Math: SURVEY A-1.
// -------------------------------------------------------------------------------------------------
template <typename T, int HLEN, int CPL>
__global__ __launch_bounds__(256) void k_ana_cols_ring(const T* __restrict__ t, T* __restrict__ lo, T* __restrict__ hi, int Nr, int Ncw,
                                                        int RO, Taps2<T> f)
{
    using V = typename VecT<T, CPL>::type;
    constexpr int C = HLEN / 2 - 1;
    constexpr int RS = HLEN + kRingPF;  // ring slots (even)
    const int lane = threadIdx.x & 63;
    const int strip = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int x0 = (strip * 64 + lane) * CPL;
    if (strip * 64 * CPL >= Ncw) return;
    const bool active = x0 < Ncw;
    const int xl = active ? x0 : 0;
    const int Nr2 = div2(Nr);
    const int y0 = blockIdx.y * RO;
    const int nout = min(RO, Nr2 - y0);
    if (nout <= 0) return;
    const int rb = 2 * y0 - C;
    const int nin = 2 * nout + HLEN - 2;

    V ring[RS];
    auto row_ptr = [&](int r) { return reinterpret_cast<const V*>(t + (size_t)wrap_ext(rb + min(r, nin - 1), Nr) * Ncw + xl); };
    cfor<RS>([&](auto S) { ring[decltype(S)::value] = *row_ptr(decltype(S)::value); });

    for (int q0 = 0; q0 < nout; q0 += RS / 2) {
        cfor<RS / 2>([&](auto U) {
            constexpr int u = decltype(U)::value;
            const int q = q0 + u;
            if (q < nout) {
                T al[CPL], ah[CPL];
#pragma unroll
                for (int p = 0; p < CPL; p++) al[p] = ah[p] = T(0);
                cfor<HLEN>([&](auto J) {
                    constexpr int j = decltype(J)::value;
                    constexpr int s = (2 * u + j) % RS;
                    const T fl = f.a[HLEN - 1 - j], fh = f.b[HLEN - 1 - j];
#pragma unroll
                    for (int p = 0; p < CPL; p++) {
                        const T v = vget<T, CPL>(ring[s], p);
                        al[p] = fma_t(v, fl, al[p]);
                        ah[p] = fma_t(v, fh, ah[p]);
                    }
                });
                if (active) {
                    V vl, vh;
#pragma unroll
                    for (int p = 0; p < CPL; p++) {
                        vset<T, CPL>(vl, p, al[p]);
                        vset<T, CPL>(vh, p, ah[p]);
                    }
                    const size_t o = (size_t)(y0 + q) * Ncw + x0;
                    *reinterpret_cast<V*>(lo + o) = vl;
                    *reinterpret_cast<V*>(hi + o) = vh;
                }
            }
            // rows 2q, 2q+1 are dead: their slots take the rows RS ahead (clamped at the chunk end: harmless re-read)
            ring[(2 * u) % RS] = *row_ptr(2 * q + RS);
            ring[(2 * u + 1) % RS] = *row_ptr(2 * q + RS + 1);
        });
    }
}

// -------------------------------------------------------------------------------------------------
// synthesis along columns: ca, cd (Nri x Nc) -> out (Nro x Nc), out = ca * IL + cd * IH.   Math: SURVEY A-2.
// Coefficient row (chunk-local) t+H2-1 completes the window t..t+H2-1 and yields the output rows
// SHIFT=1: 2t-1 (tap parity 1), 2t (parity 0);  SHIFT=0: 2t, 2t+1.
// -------------------------------------------------------------------------------------------------
template <typename T, int HLEN, int CPL>
__global__ __launch_bounds__(256) void k_syn_cols_ring(const T* __restrict__ ca, const T* __restrict__ cd, T* __restrict__ out, int Nri,
                                                        int Nc, int Nro, int RQ, Taps2<T> f)
{
    using V = typename VecT<T, CPL>::type;
    constexpr int H2 = HLEN / 2;
    constexpr int C = H2 / 2;
    constexpr int SHIFT = (H2 & 1) ? 0 : 1;
    constexpr int RS = H2 + kRingPF;
    const int lane = threadIdx.x & 63;
    const int strip = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int x0 = (strip * 64 + lane) * CPL;
    if (strip * 64 * CPL >= Nc) return;
    const bool active = x0 < Nc;
    const int xl = active ? x0 : 0;
    const int y0 = blockIdx.y * RQ;
    const int nq = min(RQ, Nri - y0);
    if (nq <= 0) return;
    const int rb = y0 - C;
    const int nrows = nq + H2 - 1 + SHIFT;
    const int nsteps = nq + SHIFT;

    V ra[RS], rd[RS];
    auto off_of = [&](int r) { return (size_t)wrap_per(rb + min(r, nrows - 1), Nri) * Nc + xl; };
    cfor<RS>([&](auto S) {
        const size_t o = off_of(decltype(S)::value);
        ra[decltype(S)::value] = *reinterpret_cast<const V*>(ca + o);
        rd[decltype(S)::value] = *reinterpret_cast<const V*>(cd + o);
    });

    auto emit = [&](auto U, auto OFF, int gy) {
        constexpr int u = decltype(U)::value, off = decltype(OFF)::value;
        T sa[CPL], sd[CPL];
#pragma unroll
        for (int p = 0; p < CPL; p++) sa[p] = sd[p] = T(0);
        cfor<H2>([&](auto J) {
            constexpr int j = decltype(J)::value;
            constexpr int s = (u + j) % RS;
            constexpr int k = HLEN - 1 - (2 * j + off);
            const T fl = f.a[k], fh = f.b[k];
#pragma unroll
            for (int p = 0; p < CPL; p++) {
                sa[p] = fma_t(vget<T, CPL>(ra[s], p), fl, sa[p]);
                sd[p] = fma_t(vget<T, CPL>(rd[s], p), fh, sd[p]);
            }
        });
        if (active) {
            V r;
#pragma unroll
            for (int p = 0; p < CPL; p++) vset<T, CPL>(r, p, sa[p] + sd[p]);
            *reinterpret_cast<V*>(out + (size_t)gy * Nc + x0) = r;
        }
    };

    for (int t0 = 0; t0 < nsteps; t0 += RS) {
        cfor<RS>([&](auto U) {
            constexpr int u = decltype(U)::value;
            const int tt = t0 + u;
            if (tt < nsteps) {
                const int g1 = 2 * tt - SHIFT, g0 = g1 + 1;
                if (g1 >= 0 && g1 < 2 * nq && 2 * y0 + g1 < Nro) emit(U, std::integral_constant<int, 1>{}, 2 * y0 + g1);
                if (g0 < 2 * nq && 2 * y0 + g0 < Nro) emit(U, std::integral_constant<int, 0>{}, 2 * y0 + g0);
            }
            const size_t o = off_of(tt + RS);  // row tt is dead: its slot takes the row RS ahead
            ra[u] = *reinterpret_cast<const V*>(ca + o);
            rd[u] = *reinterpret_cast<const V*>(cd + o);
        });
    }
}

// -------------------------------------------------------------------------------------------------
// Long filters: taps in a VGPR, broadcast by v_readlane ("_tr" kernels).
// The kernels above take the taps from the kernarg segment, i.e. from SGPRs.  2*HLEN taps of a long bank do not
// fit the ~100 SGPRs of a wave (db20 in double: 160) and, because every tap is used in every unrolled row of the
// ring period, all of them are live at once: hipcc parks them in VGPR lanes and fetches them back with TWO
// v_readlane per use (measured: 5544 v_readlane for 2240 FMAs in k_syn_cols_ring<double,40>).  Here lane k of
// one register holds tap k, a GROUP of G rows is computed tap-major (read a tap once, feed it to G rows, drop
// it), and an opaque barrier per group keeps the compiler from hoisting the reads back into one live set:
// 2 v_readlane per 2*G FMAs instead of per FMA.  Each accumulator still sums its taps in ascending order.
// -------------------------------------------------------------------------------------------------
template <typename T, int HLEN, int CPL>
__global__ __launch_bounds__(256) void k_ana_cols_ring_tr(const T* __restrict__ t, T* __restrict__ lo, T* __restrict__ hi, int Nr, int Ncw,
                                                           int RO, Taps2<T> f)
{
    using V = typename VecT<T, CPL>::type;
    constexpr int G = kTapG;
    constexpr int C = HLEN / 2 - 1;
    constexpr int RS = ((HLEN + 2 * G - 2 + 2 * G - 1) / (2 * G)) * (2 * G);  // >= HLEN + 2G - 2 (last row of a group), RS/2 % G == 0
    static_assert(HLEN <= 64, "one tap per lane");
    const int lane = threadIdx.x & 63;
    const int strip = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int x0 = (strip * 64 + lane) * CPL;
    if (strip * 64 * CPL >= Ncw) return;
    const bool active = x0 < Ncw;
    const int xl = active ? x0 : 0;
    const int Nr2 = div2(Nr);
    const int y0 = blockIdx.y * RO;
    const int nout = min(RO, Nr2 - y0);
    if (nout <= 0) return;
    const int rb = 2 * y0 - C;
    const int nin = 2 * nout + HLEN - 2;
    T tapA = f.a[min(lane, HLEN - 1)], tapB = f.b[min(lane, HLEN - 1)];  // lane k <- tap k

    V ring[RS];
    auto row_ptr = [&](int r) { return reinterpret_cast<const V*>(t + (size_t)wrap_ext(rb + min(r, nin - 1), Nr) * Ncw + xl); };
    cfor<RS>([&](auto S) { ring[decltype(S)::value] = *row_ptr(decltype(S)::value); });

    for (int q0 = 0; q0 < nout; q0 += RS / 2) {
        cfor<RS / 2 / G>([&](auto GG) {
            constexpr int u0 = decltype(GG)::value * G;
            static_assert(CPL == 1 && G == 4, "accumulator layout of tap_order");
            T acc[8];  // [2r] = lo of row r, [2r+1] = hi of row r
#pragma unroll
            for (int r = 0; r < 8; r++) acc[r] = T(0);
            cfor<HLEN>([&](auto J) {
                constexpr int j = decltype(J)::value;
                tap_order(tapA, tapB, acc);
                const T fl = lane_bcast(tapA, HLEN - 1 - j), fh = lane_bcast(tapB, HLEN - 1 - j);
                cfor<G>([&](auto Rr) {
                    constexpr int r = decltype(Rr)::value;
                    constexpr int s = (2 * (u0 + r) + j) % RS;
                    const T v = ring[s];
                    acc[2 * r] = fma_t(v, fl, acc[2 * r]);
                    acc[2 * r + 1] = fma_t(v, fh, acc[2 * r + 1]);
                });
            });
            T al[G][CPL], ah[G][CPL];
#pragma unroll
            for (int r = 0; r < G; r++) {
                al[r][0] = acc[2 * r];
                ah[r][0] = acc[2 * r + 1];
            }
            cfor<G>([&](auto Rr) {
                constexpr int r = decltype(Rr)::value;
                const int q = q0 + u0 + r;
                if (q < nout && active) {
                    V vl, vh;
#pragma unroll
                    for (int p = 0; p < CPL; p++) {
                        vset<T, CPL>(vl, p, al[r][p]);
                        vset<T, CPL>(vh, p, ah[r][p]);
                    }
                    const size_t o = (size_t)(y0 + q) * Ncw + x0;
                    *reinterpret_cast<V*>(lo + o) = vl;
                    *reinterpret_cast<V*>(hi + o) = vh;
                }
            });
            // the 2G rows the group consumed first are dead: their slots take the rows RS ahead (clamped)
            cfor<2 * G>([&](auto I) {
                constexpr int i = decltype(I)::value;
                ring[(2 * u0 + i) % RS] = *row_ptr(2 * (q0 + u0) + i + RS);
            });
        });
    }
}

template <typename T, int HLEN, int CPL>
__global__ __launch_bounds__(256) void k_syn_cols_ring_tr(const T* __restrict__ ca, const T* __restrict__ cd, T* __restrict__ out, int Nri,
                                                           int Nc, int Nro, int RQ, Taps2<T> f)
{
    using V = typename VecT<T, CPL>::type;
    constexpr int G = kTapG;
    constexpr int H2 = HLEN / 2;
    constexpr int C = H2 / 2;
    constexpr int SHIFT = (H2 & 1) ? 0 : 1;
    constexpr int RS = ((H2 + 7 + G - 1) / G) * G;  // >= H2 + G - 1, RS % G == 0, >= 4 rows of prefetch distance
    static_assert(HLEN <= 64, "one tap per lane");
    const int lane = threadIdx.x & 63;
    const int strip = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int x0 = (strip * 64 + lane) * CPL;
    if (strip * 64 * CPL >= Nc) return;
    const bool active = x0 < Nc;
    const int xl = active ? x0 : 0;
    const int y0 = blockIdx.y * RQ;
    const int nq = min(RQ, Nri - y0);
    if (nq <= 0) return;
    const int rb = y0 - C;
    const int nrows = nq + H2 - 1 + SHIFT;
    const int nsteps = nq + SHIFT;
    T tapA = f.a[min(lane, HLEN - 1)], tapB = f.b[min(lane, HLEN - 1)];

    V ra[RS], rd[RS];
    auto off_of = [&](int r) { return (size_t)wrap_per(rb + min(r, nrows - 1), Nri) * Nc + xl; };
    cfor<RS>([&](auto S) {
        const size_t o = off_of(decltype(S)::value);
        ra[decltype(S)::value] = *reinterpret_cast<const V*>(ca + o);
        rd[decltype(S)::value] = *reinterpret_cast<const V*>(cd + o);
    });

    for (int t0 = 0; t0 < nsteps; t0 += RS) {
        cfor<RS / G>([&](auto GG) {
            constexpr int u0 = decltype(GG)::value * G;
            static_assert(CPL == 1 && G == 4, "accumulator layout of tap_order");
            T acc[2][8];  // [e][2r] = a-branch, [e][2r+1] = d-branch of step r; e = 0: tap parity 1 (row g1), 1: parity 0 (row g0)
#pragma unroll
            for (int e = 0; e < 2; e++)
#pragma unroll
                for (int r = 0; r < 8; r++) acc[e][r] = T(0);
            cfor<H2>([&](auto J) {
                constexpr int j = decltype(J)::value;
                cfor<2>([&](auto E) {
                    constexpr int e = decltype(E)::value;
                    constexpr int k = HLEN - 1 - (2 * j + (1 - e));
                    tap_order(tapA, tapB, acc[e]);
                    const T fl = lane_bcast(tapA, k), fh = lane_bcast(tapB, k);
                    cfor<G>([&](auto Rr) {
                        constexpr int r = decltype(Rr)::value;
                        constexpr int s = (u0 + r + j) % RS;
                        acc[e][2 * r] = fma_t(ra[s], fl, acc[e][2 * r]);
                        acc[e][2 * r + 1] = fma_t(rd[s], fh, acc[e][2 * r + 1]);
                    });
                });
            });
            T sa[G][2][CPL], sd[G][2][CPL];
#pragma unroll
            for (int r = 0; r < G; r++)
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    sa[r][e][0] = acc[e][2 * r];
                    sd[r][e][0] = acc[e][2 * r + 1];
                }
            cfor<G>([&](auto Rr) {
                constexpr int r = decltype(Rr)::value;
                const int tt = t0 + u0 + r;
                if (tt < nsteps && active) {
                    const int g1 = 2 * tt - SHIFT, g0 = g1 + 1;
                    if (g1 >= 0 && g1 < 2 * nq && 2 * y0 + g1 < Nro) {
                        V v;
#pragma unroll
                        for (int p = 0; p < CPL; p++) vset<T, CPL>(v, p, sa[r][0][p] + sd[r][0][p]);
                        *reinterpret_cast<V*>(out + (size_t)(2 * y0 + g1) * Nc + x0) = v;
                    }
                    if (g0 < 2 * nq && 2 * y0 + g0 < Nro) {
                        V v;
#pragma unroll
                        for (int p = 0; p < CPL; p++) vset<T, CPL>(v, p, sa[r][1][p] + sd[r][1][p]);
                        *reinterpret_cast<V*>(out + (size_t)(2 * y0 + g0) * Nc + x0) = v;
                    }
                }
            });
            cfor<G>([&](auto Rr) {
                constexpr int r = decltype(Rr)::value;
                const size_t o = off_of(t0 + u0 + r + RS);
                ra[(u0 + r) % RS] = *reinterpret_cast<const V*>(ca + o);
                rd[(u0 + r) % RS] = *reinterpret_cast<const V*>(cd + o);
            });
        });
    }
}

// -------------------------------------------------------------------------------------------------
// Stationary (a-trous) column passes, level with tap spacing f = 2^(level-1).  Rows r = rho (mod f) form
// f independent sub-signals of M = Nr/f samples on which the dilated filter is an ordinary dense filter
// (the periodic wrap stays inside a residue class because f divides Nr): one wave = one column strip x one
// residue class x a chunk of the sub-signal, with the same register ring as above -- every row is read
// once per chunk instead of hlen times (reference: src/separable.cu:452-493, 553-589; math A-3 / A-4).
// -------------------------------------------------------------------------------------------------
template <typename T, int HLEN, int CPL>
__global__ __launch_bounds__(256) void k_swt_ana_cols_ring(const T* __restrict__ t, T* __restrict__ lo, T* __restrict__ hi, int Nr, int Nc,
                                                            int fct, int RO, Taps2<T> f)
{
    using V = typename VecT<T, CPL>::type;
    constexpr int C = HLEN / 2 - 1;
    constexpr int RS = HLEN + kRingPF;
    const int lane = threadIdx.x & 63;
    const int strip = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int x0 = (strip * 64 + lane) * CPL;
    if (strip * 64 * CPL >= Nc) return;
    const bool active = x0 < Nc;
    const int xl = active ? x0 : 0;
    const int M = Nr / fct;
    const int rho = blockIdx.y % fct;
    const int m0 = (blockIdx.y / fct) * RO;
    const int nout = min(RO, M - m0);
    if (nout <= 0) return;
    const int nin = nout + HLEN - 1;

    V ring[RS];
    auto row_ptr = [&](int r) {
        const int sub = wrap_per(m0 - C + min(r, nin - 1), M);
        return reinterpret_cast<const V*>(t + (size_t)(rho + fct * sub) * Nc + xl);
    };
    cfor<RS>([&](auto S) { ring[decltype(S)::value] = *row_ptr(decltype(S)::value); });

    for (int q0 = 0; q0 < nout; q0 += RS) {
        cfor<RS>([&](auto U) {
            constexpr int u = decltype(U)::value;
            const int q = q0 + u;
            if (q < nout) {
                T al[CPL], ah[CPL];
#pragma unroll
                for (int p = 0; p < CPL; p++) al[p] = ah[p] = T(0);
                cfor<HLEN>([&](auto J) {
                    constexpr int j = decltype(J)::value;
                    constexpr int s = (u + j) % RS;
                    const T fl = f.a[HLEN - 1 - j], fh = f.b[HLEN - 1 - j];
#pragma unroll
                    for (int p = 0; p < CPL; p++) {
                        const T v = vget<T, CPL>(ring[s], p);
                        al[p] = fma_t(v, fl, al[p]);
                        ah[p] = fma_t(v, fh, ah[p]);
                    }
                });
                if (active) {
                    V vl, vh;
#pragma unroll
                    for (int p = 0; p < CPL; p++) {
                        vset<T, CPL>(vl, p, al[p]);
                        vset<T, CPL>(vh, p, ah[p]);
                    }
                    const size_t o = (size_t)(rho + fct * (m0 + q)) * Nc + x0;
                    *reinterpret_cast<V*>(lo + o) = vl;
                    *reinterpret_cast<V*>(hi + o) = vh;
                }
            }
            ring[u] = *row_ptr(q + RS);  // row q of the chunk is dead
        });
    }
}

// out = a * (IL/2) + d * (IH/2) along columns (taps pre-halved by the caller), centre c = (hlen/2) * f
template <typename T, int HLEN, int CPL>
__global__ __launch_bounds__(256) void k_swt_syn_cols_ring(const T* __restrict__ ca, const T* __restrict__ cd, T* __restrict__ out, int Nr,
                                                            int Nc, int fct, int RO, Taps2<T> f)
{
    using V = typename VecT<T, CPL>::type;
    constexpr int C = HLEN / 2;
    constexpr int RS = HLEN + kRingPF;
    const int lane = threadIdx.x & 63;
    const int strip = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int x0 = (strip * 64 + lane) * CPL;
    if (strip * 64 * CPL >= Nc) return;
    const bool active = x0 < Nc;
    const int xl = active ? x0 : 0;
    const int M = Nr / fct;
    const int rho = blockIdx.y % fct;
    const int m0 = (blockIdx.y / fct) * RO;
    const int nout = min(RO, M - m0);
    if (nout <= 0) return;
    const int nin = nout + HLEN - 1;

    V ra[RS], rd[RS];
    auto off_of = [&](int r) { return (size_t)(rho + fct * wrap_per(m0 - C + min(r, nin - 1), M)) * Nc + xl; };
    cfor<RS>([&](auto S) {
        const size_t o = off_of(decltype(S)::value);
        ra[decltype(S)::value] = *reinterpret_cast<const V*>(ca + o);
        rd[decltype(S)::value] = *reinterpret_cast<const V*>(cd + o);
    });
    for (int q0 = 0; q0 < nout; q0 += RS) {
        cfor<RS>([&](auto U) {
            constexpr int u = decltype(U)::value;
            const int q = q0 + u;
            if (q < nout) {
                T sa[CPL], sd[CPL];
#pragma unroll
                for (int p = 0; p < CPL; p++) sa[p] = sd[p] = T(0);
                cfor<HLEN>([&](auto J) {
                    constexpr int j = decltype(J)::value;
                    constexpr int s = (u + j) % RS;
                    const T fl = f.a[HLEN - 1 - j], fh = f.b[HLEN - 1 - j];
#pragma unroll
                    for (int p = 0; p < CPL; p++) {
                        sa[p] = fma_t(vget<T, CPL>(ra[s], p), fl, sa[p]);
                        sd[p] = fma_t(vget<T, CPL>(rd[s], p), fh, sd[p]);
                    }
                });
                if (active) {
                    V r;
#pragma unroll
                    for (int p = 0; p < CPL; p++) vset<T, CPL>(r, p, sa[p] + sd[p]);
                    *reinterpret_cast<V*>(out + (size_t)(rho + fct * (m0 + q)) * Nc + x0) = r;
                }
            }
            const size_t o = off_of(q + RS);
            ra[u] = *reinterpret_cast<const V*>(ca + o);
            rd[u] = *reinterpret_cast<const V*>(cd + o);
        });
    }
}

// =================================================================================================
// host side
// =================================================================================================
#define PDWT_CHECK_LAUNCH() PDWT_HIP_TRY(hipGetLastError())

static int pick_chunk(int nrows, int strips, int unit)
{
    static const int env = getenv("PDWT_RING_R") ? atoi(getenv("PDWT_RING_R")) : 0;
    int R = env > 0 ? env : (int)(((long long)nrows * strips) / 4096);
    if (R < unit) R = unit;
    if (R > 256) R = 256;
    return R;
}

template <typename T, int HLEN, int CPL>
static int launch_ana(const T* t, T* lo, T* hi, int Nr, int Ncw, const Taps2<T>& f)
{
    const int strips = idiv_up(Ncw, 64 * CPL);
    const int RO = pick_chunk(div2(Nr), strips, HLEN);
    dim3 grid(idiv_up(strips, 4), idiv_up(div2(Nr), RO));
    // taps beyond the SGPR budget: VGPR tap register + v_readlane (k_*_tr above)
    if constexpr (sizeof(T) == 8 && HLEN >= 20 && CPL == 1)
        hipLaunchKernelGGL((k_ana_cols_ring_tr<T, HLEN, CPL>), grid, dim3(256), 0, stream(), t, lo, hi, Nr, Ncw, RO, f);
    else
        hipLaunchKernelGGL((k_ana_cols_ring<T, HLEN, CPL>), grid, dim3(256), 0, stream(), t, lo, hi, Nr, Ncw, RO, f);
    PDWT_CHECK_LAUNCH();
    return PDWT_OK;
}
template <typename T, int HLEN, int CPL>
static int launch_syn(const T* ca, const T* cd, T* out, int Nri, int Nc, int Nro, const Taps2<T>& f)
{
    const int strips = idiv_up(Nc, 64 * CPL);
    const int RQ = pick_chunk(Nri, strips, HLEN / 2);
    dim3 grid(idiv_up(strips, 4), idiv_up(Nri, RQ));
    if constexpr (sizeof(T) == 8 && HLEN >= 20 && CPL == 1)
        hipLaunchKernelGGL((k_syn_cols_ring_tr<T, HLEN, CPL>), grid, dim3(256), 0, stream(), ca, cd, out, Nri, Nc, Nro, RQ, f);
    else
        hipLaunchKernelGGL((k_syn_cols_ring<T, HLEN, CPL>), grid, dim3(256), 0, stream(), ca, cd, out, Nri, Nc, Nro, RQ, f);
    PDWT_CHECK_LAUNCH();
    return PDWT_OK;
}

// lengths for which the ring kernels are instantiated (long / double-precision banks that skip the fused kernels)
#define PDWT_RING_HLENS(X) X(12) X(14) X(16) X(18) X(20) X(24) X(30) X(40)

template <typename T> static bool vec_ok(const void* a, const void* b, const void* c, int n, int cpl)
{
    const uintptr_t m = (uintptr_t)(cpl * sizeof(T) - 1);
    return (n % cpl) == 0 && (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c) & m) == 0;
}

template <typename T>
int ana_cols_ring(const T* t, T* lo, T* hi, int Nr, int Ncw, int hlen, const Taps2<T>& f)
{
    constexpr int CPLV = 16 / sizeof(T) / 2;  // 8-byte accesses for float (2 cols), 8-byte for double (1 col)
    const bool v = CPLV > 1 && vec_ok<T>(t, lo, hi, Ncw, CPLV);
    switch (hlen) {
#define X(H) \
    case H: return v ? launch_ana<T, H, CPLV>(t, lo, hi, Nr, Ncw, f) : launch_ana<T, H, 1>(t, lo, hi, Nr, Ncw, f);
        PDWT_RING_HLENS(X)
#undef X
        default: return 1;
    }
}
template <typename T>
int syn_cols_ring(const T* ca, const T* cd, T* out, int Nri, int Nc, int Nro, int hlen, const Taps2<T>& f)
{
    constexpr int CPLV = 16 / sizeof(T) / 2;
    const bool v = CPLV > 1 && vec_ok<T>(ca, cd, out, Nc, CPLV);
    switch (hlen) {
#define X(H) \
    case H: return v ? launch_syn<T, H, CPLV>(ca, cd, out, Nri, Nc, Nro, f) : launch_syn<T, H, 1>(ca, cd, out, Nri, Nc, Nro, f);
        PDWT_RING_HLENS(X)
#undef X
        default: return 1;
    }
}

#define PDWT_SWT_RING_HLENS(X) X(2) X(4) X(6) X(8) X(10) X(12) X(14) X(16) X(18) X(20)

template <typename T, int HLEN, int CPL>
static int launch_swt_ana(const T* t, T* lo, T* hi, int Nr, int Nc, int fct, const Taps2<T>& f)
{
    const int strips = idiv_up(Nc, 64 * CPL);
    const int M = Nr / fct;
    int RO = pick_chunk(Nr, strips, HLEN);  // Nr rows in total = fct residue classes x M
    if (RO > M) RO = M;
    dim3 grid(idiv_up(strips, 4), fct * idiv_up(M, RO));
    hipLaunchKernelGGL((k_swt_ana_cols_ring<T, HLEN, CPL>), grid, dim3(256), 0, stream(), t, lo, hi, Nr, Nc, fct, RO, f);
    PDWT_CHECK_LAUNCH();
    return PDWT_OK;
}
template <typename T, int HLEN, int CPL>
static int launch_swt_syn(const T* ca, const T* cd, T* out, int Nr, int Nc, int fct, const Taps2<T>& f)
{
    const int strips = idiv_up(Nc, 64 * CPL);
    const int M = Nr / fct;
    int RO = pick_chunk(Nr, strips, HLEN);
    if (RO > M) RO = M;
    dim3 grid(idiv_up(strips, 4), fct * idiv_up(M, RO));
    hipLaunchKernelGGL((k_swt_syn_cols_ring<T, HLEN, CPL>), grid, dim3(256), 0, stream(), ca, cd, out, Nr, Nc, fct, RO, f);
    PDWT_CHECK_LAUNCH();
    return PDWT_OK;
}

template <typename T>
int swt_ana_cols_ring(const T* t, T* lo, T* hi, int Nr, int Nc, int hlen, int fct, const Taps2<T>& f)
{
    if (Nr % fct != 0 || Nr / fct < hlen) return 1;  // the wrap must stay inside a residue class
    constexpr int CPLV = 16 / sizeof(T) / 2;
    const bool v = CPLV > 1 && vec_ok<T>(t, lo, hi, Nc, CPLV);
    switch (hlen) {
#define X(H) \
    case H: return v ? launch_swt_ana<T, H, CPLV>(t, lo, hi, Nr, Nc, fct, f) : launch_swt_ana<T, H, 1>(t, lo, hi, Nr, Nc, fct, f);
        PDWT_SWT_RING_HLENS(X)
#undef X
        default: return 1;
    }
}
template <typename T>
int swt_syn_cols_ring(const T* ca, const T* cd, T* out, int Nr, int Nc, int hlen, int fct, const Taps2<T>& f)
{
    if (Nr % fct != 0 || Nr / fct < hlen) return 1;
    constexpr int CPLV = 16 / sizeof(T) / 2;
    const bool v = CPLV > 1 && vec_ok<T>(ca, cd, out, Nc, CPLV);
    switch (hlen) {
#define X(H) \
    case H: return v ? launch_swt_syn<T, H, CPLV>(ca, cd, out, Nr, Nc, fct, f) : launch_swt_syn<T, H, 1>(ca, cd, out, Nr, Nc, fct, f);
        PDWT_SWT_RING_HLENS(X)
#undef X
        default: return 1;
    }
}
template int swt_ana_cols_ring<float>(const float*, float*, float*, int, int, int, int, const Taps2<float>&);
template int swt_ana_cols_ring<double>(const double*, double*, double*, int, int, int, int, const Taps2<double>&);
template int swt_syn_cols_ring<float>(const float*, const float*, float*, int, int, int, int, const Taps2<float>&);
template int swt_syn_cols_ring<double>(const double*, const double*, double*, int, int, int, int, const Taps2<double>&);

template int ana_cols_ring<float>(const float*, float*, float*, int, int, int, const Taps2<float>&);
template int ana_cols_ring<double>(const double*, double*, double*, int, int, int, const Taps2<double>&);
template int syn_cols_ring<float>(const float*, const float*, float*, int, int, int, int, const Taps2<float>&);
template int syn_cols_ring<double>(const double*, const double*, double*, int, int, int, int, const Taps2<double>&);

}  // namespace pdwt
