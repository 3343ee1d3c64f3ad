// dwt1d_fused.hip -- batched-1D DWT, ALL LEVELS in one launch, one workgroup per signal (gfx950).
//
// Path replaced: reference w_forward_separable_1d / w_inverse_separable_1d (src/separable.cu:214-236,
// 368-395): L launches of the row-pass kernel, the approximation round-tripping through global memory
// between levels (+ a device-to-device copy when L is even).
//
// MI355X design: a signal of up to 16K float32 samples fits LDS many times over (160 KiB per CU), so the
// whole row is staged once (16-byte coalesced loads), every level is computed LDS -> LDS with the
// periodic / odd-size extension resolved by plain index arithmetic on the resident row, detail bands go
// straight to HBM with 16-byte stores and only the final approximation leaves the chip:
// HBM traffic = read N + write N per direction = the ALGORITHMIC minimum (the per-level form moves
// 2N(1 + 1/2 + ... ) ~ 3.75N for 4 levels).  Rows are independent, so the batch needs no halo exchange and
// shards across GPUs by rows.
//
// Each work item produces PO adjacent outputs from ONE register window of 2*PO + hlen - 2 samples
// (aligned 16-byte LDS reads), so an input sample is read from LDS ~1.4x instead of hlen/2 times.
// Tap order / one FMA per tap as in the reference kernels (src/separable.cu:112-127, 305-326): the
// results are bit-identical to the per-level kernels and to the CPU oracle.
#include <algorithm>
#include <type_traits>

#include "common.hpp"
#include "dwt1d_fused.hpp"

// This file is compiled TWICE: as itself (variant `dflt`: default cache policy) and through dwt1d_fused_nt.hip (variant `nt`: float32 only,
// PDWT_1D_NT = 5 -- non-temporal row loads in the forward and band loads in the inverse kernels).  forward_separable_1d / inverse_separable_1d
// (dwt.hip) take the `nt` variant for batches that do not fit the Infinity Cache (knob dwt1d_nt_mb): a row that is read once then leaves no line
// behind -- C4 (8192 x 8192 float32, 268 MB per image): -2.5 % over 20 interleaved pairs; a 64 MB batch that LIVES in the cache would lose 3-18 %.
#ifndef PDWT_1D_VARIANT
#define PDWT_1D_VARIANT dflt
#endif

namespace pdwt {
namespace PDWT_1D_VARIANT {

constexpr int kMaxLev1D = 32;

template <typename T>
struct Bands1D {
    T* p[kMaxLev1D + 1];   // p[0] = A_L, p[l] = D_l (l = 1 finest)
    int n[kMaxLev1D + 1];  // n[0] = Nc, n[l] = samples per row at level l
    int nlev;
};

// clang ext vectors (not HIP's float4/double2 structs): element access by index keeps them in registers;
// with the struct types the prefetch array below ended up in scratch memory.
typedef float v4f_t __attribute__((ext_vector_type(4)));
typedef double v2d_t __attribute__((ext_vector_type(2)));
typedef float pair_f32 __attribute__((ext_vector_type(2)));
template <typename T> struct Vec16;
template <> struct Vec16<float> { using type = v4f_t; static constexpr int N = 4; };
template <> struct Vec16<double> { using type = v2d_t; static constexpr int N = 2; };

// -------------------------------------------------------------------------------------------------
// LDS row buffers carry their periodic extension explicitly: [HL halo | n samples | HR halo].  After a
// level has written its n outputs, `fill_halo` copies the wrapped samples into the halo cells (<= ~2*hlen
// cells, one barrier), so EVERY work item reads its window with plain aligned 16-byte LDS loads -- no
// per-item wrap arithmetic, no divergent edge path.
// -------------------------------------------------------------------------------------------------
template <typename T, bool EXT>
__device__ __forceinline__ void fill_halo(T* data, int n, int HL, int HR)
{
    // data[-HL .. -1] and data[n .. n+HR-1] <- periodic images (EXT: after the virtual repeat of the last
    // sample when n is odd, src/separable.cu:116-121)
    for (int k = threadIdx.x; k < HL + HR; k += 256) {
        const int idx = k < HL ? k - HL : n + (k - HL);
        data[idx] = data[EXT ? wrap_ext(idx, n) : wrap_per(idx, n)];
    }
}

// -------------------------------------------------------------------------------------------------
// forward: row -> [A_L, D_1 .. D_L]
// -------------------------------------------------------------------------------------------------
template <typename T, int HLEN>
struct Fwd1DGeom {
    static constexpr int NV = Vec16<T>::N;
    static constexpr int PO = NV;  // outputs per work item (one 16-byte store per band)
    static constexpr int C = (HLEN & 1) ? HLEN / 2 : HLEN / 2 - 1;
    static constexpr int CA = ((C + NV - 1) / NV) * NV;  // left halo = window start rounded down to 16 bytes
    static constexpr int WL = ((CA + 2 * PO + (HLEN - 2 - C) + NV - 1) / NV) * NV;
    static constexpr int HL = CA;
    static constexpr int HR = WL;  // covers the last (possibly partial) item's window
    static __host__ __device__ int buf_elems(int n) { return HL + ((n + NV - 1) / NV) * NV + HR; }
};

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. every barrier
// would wait for the global STORES of the level just finished and for the PREFETCH of the next row
// (measured: 78 % of wave cycles in SQ_WAIT_ANY).  LDS hand-offs only need this wave's DS operations retired.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Cache policy of the row traffic (every row is read once and written once).  PDWT_1D_NT: bit 0 = non-temporal row loads in the forward
// kernels, bit 2 = non-temporal band loads in the inverse kernels, bit 1 = non-temporal band / row stores.  Measured on the C4 shape
// (docs/EXPERIMENTS.md, round 5); the default is what won.
#ifndef PDWT_1D_NT
#define PDWT_1D_NT 0
#endif
template <int BIT, typename V>
__device__ __forceinline__ V ld_row(const V* p)
{
    if constexpr ((PDWT_1D_NT & BIT) != 0)
        return __builtin_nontemporal_load(p);
    else
        return *p;
}
template <typename V>
__device__ __forceinline__ void st_row(V* p, const V& v)
{
#if PDWT_1D_NT & 2
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}

// LDS bank swizzle.  A work item's window is WL/NV consecutive 16-byte slots and consecutive lanes start two
// slots apart, so for one ds_read_b128 the lanes of a service group hit every slot of the 256-byte bank row
// twice (measured: SQ_LDS_BANK_CONFLICT = 80 % of SQ_LDS_IDX_ACTIVE).  XOR-ing bit 0 of the slot index with
// bit 4 sends the two colliding lanes (8 lanes = 16 slots apart) to adjacent slots: conflict-free.
__device__ __forceinline__ int swz(int slot) { return slot ^ ((slot >> 4) & 1); }

template <int I, int N, typename F>
__device__ __forceinline__ void sfor_impl(F&& fn)
{
    if constexpr (I < N) {
        fn(std::integral_constant<int, I>{});
        sfor_impl<I + 1, N>(fn);
    }
}
template <int N, typename F>
__device__ __forceinline__ void sfor(F&& fn) { sfor_impl<0, N>(fn); }

constexpr int kPre1D = 8;  // 16-byte chunks of the NEXT row each thread keeps in flight (rows <= 8*256 chunks)

template <typename T, int HLEN, bool PREFETCH>
__global__ __launch_bounds__(256) void k_fwd1d_fused(const T* __restrict__ in, Bands1D<T> b, int Nr, Taps2<T> f)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using G = Fwd1DGeom<T, HLEN>;
    using V = typename Vec16<T>::type;
    constexpr int NV = G::NV, PO = G::PO, C = G::C, CA = G::CA, WL = G::WL, HL = G::HL, HR = G::HR;
    constexpr int HLS = HL / NV;  // halo in 16-byte slots
    const int Nc = b.n[0];
    V* const B0 = reinterpret_cast<V*>(smem);   // buffer 0: level input rows of n0, n2, ...
    V* const B1 = B0 + G::buf_elems(Nc) / NV;   // buffer 1: n1, n3, ...
    auto elem = [](V* B, int e) -> T& {         // sample e (>= -HL) of the row held in buffer B
        const int le = e + HL;
        return reinterpret_cast<T*>(B + swz(le / NV))[le % NV];
    };
    auto fill_halo = [&](V* B, int n) {         // periodic images after the odd-size extension (A-1)
        for (int k = threadIdx.x; k < HL + HR; k += 256) {
            const int idx = k < HL ? k - HL : n + (k - HL);
            elem(B, idx) = elem(B, wrap_ext(idx, n));
        }
    };
    const bool rowvec = (Nc % NV) == 0 && (reinterpret_cast<uintptr_t>(in) & 15) == 0;
    const int nchunks = Nc / NV;
    constexpr bool prefetch = PREFETCH;  // host guarantees rowvec && nchunks <= kPre1D*256 when set

    V pre[kPre1D];
    // compile-time indices only (sfor): any index hipcc cannot fold sends the whole array to scratch memory
#define PDWT_ISSUE_ROW(ROW)                                                                                   \
    {                                                                                                          \
        const V* src_ = reinterpret_cast<const V*>(in + (size_t)(ROW) * (size_t)Nc);                            \
        sfor<kPre1D>([&](auto K_) {                                                                            \
            constexpr int k_ = decltype(K_)::value;                                                            \
            pre[k_] = ld_row<1>(src_ + min((int)threadIdx.x + 256 * k_, nchunks - 1));                                     \
        });                                                                                                    \
    }
    // branch-free (index clamped, not predicated): a conditional load makes hipcc wait for each one at the join;
    // clamped loads issue back to back and stay in flight until the next row is staged.
    // The prefetch is UNCONDITIONAL (row index clamped): a conditional issue makes `pre` a phi of old and new
    // values, and hipcc resolves that phi with register copies right behind the loads -- i.e. it waits for
    // them on the spot and the prefetch overlaps nothing.
    size_t row = blockIdx.x;
    if constexpr (prefetch) PDWT_ISSUE_ROW(row < (size_t)Nr ? row : (size_t)Nr - 1)

    for (; row < (size_t)Nr; row += gridDim.x) {
        // ---- stage the row (from the registers prefetched during the previous row) ----
        if constexpr (prefetch) {
            sfor<kPre1D>([&](auto K_) {
                constexpr int k = decltype(K_)::value;
                if (threadIdx.x + 256 * k < nchunks) B0[swz(HLS + threadIdx.x + 256 * k)] = pre[k];
            });
            const size_t nr = row + gridDim.x;
            PDWT_ISSUE_ROW(nr < (size_t)Nr ? nr : (size_t)Nr - 1)  // lands while this row is transformed
        } else if (rowvec) {
            const V* src = reinterpret_cast<const V*>(in + row * (size_t)Nc);
            for (int i = threadIdx.x; i < nchunks; i += 256) B0[swz(HLS + i)] = src[i];
        } else {
            const T* src = in + row * (size_t)Nc;
            for (int i = threadIdx.x; i < Nc; i += 256) elem(B0, i) = src[i];
        }
        lds_barrier();
        fill_halo(B0, Nc);
        lds_barrier();

        V* cur = B0;
        V* nxt = B1;
        int n = Nc;
        for (int lev = 1; lev <= b.nlev; lev++) {
            const int no = (n + 1) >> 1;
            T* gd = b.p[lev] + row * (size_t)no;
            T* ga = b.p[0] + row * (size_t)no;
            const bool last = lev == b.nlev;
            const bool vec_ok = ((no % NV) == 0) && ((reinterpret_cast<uintptr_t>(gd) & 15) == 0) && ((reinterpret_cast<uintptr_t>(ga) & 15) == 0);
            const int items = (no + PO - 1) / PO;
            for (int it = threadIdx.x; it < items; it += 256) {
                const int i0 = it * PO;
                T w[WL];  // window = slots 2*it .. 2*it + WL/NV - 1 (sample 2*i0 - CA onwards)
#pragma unroll
                for (int k = 0; k < WL / NV; k++) {
                    const V t = cur[swz(2 * it + k)];
#pragma unroll
                    for (int q = 0; q < NV; q++) w[k * NV + q] = t[q];
                }
                V vlo, vhi;
                if constexpr (sizeof(T) == 4) {
                    // explicit (lo, hi) pairs: one v_pk_fma_f32 per (output, tap), the sample broadcast to both
                    // halves by op_sel and the taps travelling as (L[k], H[k]) SGPR pairs.  Left to itself hipcc
                    // packs two OUTPUTS per instruction instead and spends ~0.7 v_mov per FMA pair assembling
                    // (w[k], w[k+2]) operands.  Same taps in the same order, one FMA each: bit-identical.
                    pair_f32 acc[PO];
#pragma unroll
                    for (int q = 0; q < PO; q++) acc[q] = pair_f32{0.f, 0.f};
#pragma unroll
                    for (int j = 0; j < HLEN; j++) {
                        const pair_f32 t = pair_f32{f.a[HLEN - 1 - j], f.b[HLEN - 1 - j]};
#pragma unroll
                        for (int q = 0; q < PO; q++) {
                            const float v = w[CA - C + 2 * q + j];
                            acc[q] = __builtin_elementwise_fma(pair_f32{v, v}, t, acc[q]);
                        }
                    }
#pragma unroll
                    for (int q = 0; q < PO; q++) {
                        vlo[q] = acc[q][0];
                        vhi[q] = acc[q][1];
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < PO; q++) {
                        T l = 0, h = 0;
#pragma unroll
                        for (int j = 0; j < HLEN; j++) {
                            const T v = w[CA - C + 2 * q + j];
                            l = fma_t(v, f.a[HLEN - 1 - j], l);
                            h = fma_t(v, f.b[HLEN - 1 - j], h);
                        }
                        vlo[q] = l;
                        vhi[q] = h;
                    }
                }
                if (vec_ok) {  // wave-uniform
                    st_row(reinterpret_cast<V*>(gd + i0), vhi);
                    if (last) st_row(reinterpret_cast<V*>(ga + i0), vlo);
                } else {
#pragma unroll
                    for (int q = 0; q < PO; q++)
                        if (i0 + q < no) {
                            gd[i0 + q] = vhi[q];
                            if (last) ga[i0 + q] = vlo[q];
                        }
                }
                if (!last) nxt[swz(HLS + it)] = vlo;  // a partial last item spills into halo cells, refilled below
            }
            lds_barrier();
            if (last) break;
            fill_halo(nxt, no);
            lds_barrier();
            V* t = cur;
            cur = nxt;
            nxt = t;
            n = no;
        }
    }
}

// -------------------------------------------------------------------------------------------------
// forward, ONE row buffer ("in place"): rows whose two-buffer footprint leaves room for a single workgroup per CU -- double precision at
// 8192 samples: 64 KiB row + 32 KiB level-1 approximation -- ran the per-level kernels (3.75x the traffic: 0.81 ms for the C4 shape in
// double against 0.22 in float).  Here a level's approximation outputs wait in REGISTERS (at most MAXIT items per thread) until every
// thread has read its windows, then overwrite the front of the same buffer: 66 KiB, two workgroups per CU, and the next row's 16 chunks
// per thread are prefetched during the row (KPRE).  Same item arithmetic as k_fwd1d_fused: bit-identical.
// -------------------------------------------------------------------------------------------------
template <typename T, int HLEN, int MAXIT, int KPRE>
__global__ __launch_bounds__(256, 2) void k_fwd1d_fused_ip(const T* __restrict__ in, Bands1D<T> b, int Nr, Taps2<T> f)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using G = Fwd1DGeom<T, HLEN>;
    using V = typename Vec16<T>::type;
    constexpr int NV = G::NV, PO = G::PO, C = G::C, CA = G::CA, WL = G::WL, HL = G::HL, HR = G::HR;
    constexpr int HLS = HL / NV;
    const int Nc = b.n[0];
    V* const B0 = reinterpret_cast<V*>(smem);
    auto elem = [](V* B, int e) -> T& {
        const int le = e + HL;
        return reinterpret_cast<T*>(B + swz(le / NV))[le % NV];
    };
    auto fill_halo = [&](V* B, int n) {
        for (int k = threadIdx.x; k < HL + HR; k += 256) {
            const int idx = k < HL ? k - HL : n + (k - HL);
            elem(B, idx) = elem(B, wrap_ext(idx, n));
        }
    };
    const int nchunks = Nc / NV;  // (host: Nc % NV == 0, aligned rows, nchunks <= KPRE * 256)
    V pre[KPRE];
    // (the lane index is laundered per row: otherwise hipcc hoists the KPRE clamped 64-bit addresses out of the row loop and keeps -- spills --
    // 2 x KPRE registers of them)
#define PDWT_ISSUE_ROW_IP(ROW)                                                                                 \
    {                                                                                                          \
        const V* src_ = reinterpret_cast<const V*>(in + (size_t)(ROW) * (size_t)Nc);                            \
        int tx_ = threadIdx.x;                                                                                 \
        asm volatile("" : "+v"(tx_));                                                                          \
        sfor<KPRE>([&](auto K_) {                                                                              \
            constexpr int k_ = decltype(K_)::value;                                                            \
            pre[k_] = ld_row<1>(src_ + min(tx_ + 256 * k_, nchunks - 1));                                             \
        });                                                                                                    \
    }
    size_t row = blockIdx.x;
    PDWT_ISSUE_ROW_IP(row < (size_t)Nr ? row : (size_t)Nr - 1)
    for (; row < (size_t)Nr; row += gridDim.x) {
        sfor<KPRE>([&](auto K_) {
            constexpr int k = decltype(K_)::value;
            if (threadIdx.x + 256 * k < nchunks) B0[swz(HLS + threadIdx.x + 256 * k)] = pre[k];
        });
        {
            const size_t nr = row + gridDim.x;
            PDWT_ISSUE_ROW_IP(nr < (size_t)Nr ? nr : (size_t)Nr - 1)  // lands while this row is transformed
        }
        lds_barrier();
        fill_halo(B0, Nc);
        lds_barrier();
        int n = Nc;
        for (int lev = 1; lev <= b.nlev; lev++) {
            const int no = (n + 1) >> 1;
            T* gd = b.p[lev] + row * (size_t)no;
            T* ga = b.p[0] + row * (size_t)no;
            const bool last = lev == b.nlev;
            const bool vec_ok = ((no % NV) == 0) && ((reinterpret_cast<uintptr_t>(gd) & 15) == 0) && ((reinterpret_cast<uintptr_t>(ga) & 15) == 0);
            const int items = (no + PO - 1) / PO;  // (host: items <= MAXIT * 256 at level 1, hence at every level)
            V keep[MAXIT];
            sfor<MAXIT>([&](auto Kk) {
                constexpr int kk = decltype(Kk)::value;
                const int it = threadIdx.x + 256 * kk;
                if (it < items) {
                    const int i0 = it * PO;
                    T w[WL];
#pragma unroll
                    for (int k = 0; k < WL / NV; k++) {
                        const V t = B0[swz(2 * it + k)];
#pragma unroll
                        for (int q = 0; q < NV; q++) w[k * NV + q] = t[q];
                    }
                    V vlo, vhi;
#pragma unroll
                    for (int q = 0; q < PO; q++) {
                        T l = 0, h = 0;
#pragma unroll
                        for (int j = 0; j < HLEN; j++) {
                            const T v = w[CA - C + 2 * q + j];
                            l = fma_t(v, f.a[HLEN - 1 - j], l);
                            h = fma_t(v, f.b[HLEN - 1 - j], h);
                        }
                        vlo[q] = l;
                        vhi[q] = h;
                    }
                    if (vec_ok) {
                        st_row(reinterpret_cast<V*>(gd + i0), vhi);
                        if (last) st_row(reinterpret_cast<V*>(ga + i0), vlo);
                    } else {
#pragma unroll
                        for (int q = 0; q < PO; q++)
                            if (i0 + q < no) {
                                gd[i0 + q] = vhi[q];
                                if (last) ga[i0 + q] = vlo[q];
                            }
                    }
                    keep[kk] = vlo;
                }
                // (one item at a time: interleaved, the windows of several items push the kernel past 256 registers = one workgroup per CU)
                __builtin_amdgcn_sched_barrier(0);
            });
            lds_barrier();  // every window of this level has been read
            if (last) break;
            sfor<MAXIT>([&](auto Kk) {
                constexpr int kk = decltype(Kk)::value;
                const int it = threadIdx.x + 256 * kk;
                if (it < items) B0[swz(HLS + it)] = keep[kk];  // (a partial last item spills into halo cells, refilled below)
            });
            lds_barrier();
            fill_halo(B0, no);
            lds_barrier();
            n = no;
        }
    }
#undef PDWT_ISSUE_ROW_IP
}

// -------------------------------------------------------------------------------------------------
// inverse: [A_L, D_1 .. D_L] -> row.  Per level (coarse -> fine): a (LDS) and d (staged into LDS)
// -> out (LDS); the last level writes the image row with 16-byte stores.  Math: SURVEY A-2.
// -------------------------------------------------------------------------------------------------
template <typename T, int HLEN>
struct Inv1DGeom {
    static constexpr int NV = Vec16<T>::N;
    static constexpr int PO = 2 * NV;                    // outputs per work item = NV coefficient positions:
                                                         // consecutive lanes read windows 16 bytes apart -> conflict-free ds_read_b128
    static constexpr int H2 = HLEN / 2;
    static constexpr int C = H2 / 2;
    static constexpr int SHIFT = (H2 & 1) ? 0 : 1;
    static constexpr int HL = ((C + NV - 1) / NV) * NV;  // left halo (data region stays 16-byte aligned)
    static constexpr int OFF0 = HL - C;                  // first needed coefficient inside the aligned window
    static constexpr int WL = ((OFF0 + PO / 2 + H2 + 1 + NV - 1) / NV) * NV;  // aligned window length
    static constexpr int HR = ((WL + PO + NV - 1) / NV) * NV;
    static __host__ __device__ int buf_elems(int n) { return HL + ((n + NV - 1) / NV) * NV + HR; }
};

// one work item of a synthesis level: PO outputs g0 = it*PO .. from the aligned coefficient windows of a and d
template <typename T, int HLEN>
__device__ __forceinline__ void inv1d_item(const T* a, const T* sd, int it, const Taps2<T>& f, typename Vec16<T>::type (&res)[2])
{
    using G = Inv1DGeom<T, HLEN>;
    using V = typename Vec16<T>::type;
    constexpr int NV = G::NV, PO = G::PO, H2 = G::H2, SHIFT = G::SHIFT, WL = G::WL, HL = G::HL, OFF0 = G::OFF0;
    const V* pa = reinterpret_cast<const V*>(a + it * NV - HL);
    const V* pd = reinterpret_cast<const V*>(sd + it * NV - HL);
    T wa[WL], wd[WL];
#pragma unroll
    for (int k = 0; k < WL / NV; k++) {
        const V ta = pa[k], td = pd[k];
#pragma unroll
        for (int q = 0; q < NV; q++) {
            wa[k * NV + q] = ta[q];
            wd[k * NV + q] = td[q];
        }
    }
    if constexpr (sizeof(T) == 4) {
        // The outputs gp = 2m (even phase) and gp = 2m+1 (odd phase) read the SAME coefficients a[m+j], d[m+j] with
        // the even / odd taps: one v_pk_fma_f32 per (coefficient, tap pair), the coefficient broadcast by op_sel --
        // no operand assembly (hipcc's own packing spends a v_mov per two FMAs).  An unpaired output at either end
        // of the item (SHIFT = 1) takes plain FMAs.  Same taps, same order, one FMA each, same final s1 + s2.
#pragma unroll
        for (int m = 0; 2 * m < PO + SHIFT; m++) {
            const bool has_e = 2 * m >= SHIFT, has_o = 2 * m + 1 < PO + SHIFT;  // gp = 2m / 2m+1 inside the item
            if (has_e && has_o) {
                pair_f32 s1 = {0.f, 0.f}, s2 = {0.f, 0.f};
#pragma unroll
                for (int j = 0; j < H2; j++) {
                    const int ke = HLEN - 2 - 2 * j, ko = HLEN - 1 - 2 * j;
                    const float xa = wa[OFF0 + m + j], xd = wd[OFF0 + m + j];
                    s1 = __builtin_elementwise_fma(pair_f32{xa, xa}, pair_f32{f.a[ke], f.a[ko]}, s1);
                    s2 = __builtin_elementwise_fma(pair_f32{xd, xd}, pair_f32{f.b[ke], f.b[ko]}, s2);
                }
                const pair_f32 r = s1 + s2;
                const int q = 2 * m - SHIFT;
                res[q / NV][q % NV] = r[0];
                res[(q + 1) / NV][(q + 1) % NV] = r[1];
            } else {
                const int gp = has_e ? 2 * m : 2 * m + 1, off = 1 - (gp & 1), q = gp - SHIFT;
                float s1 = 0, s2 = 0;
#pragma unroll
                for (int j = 0; j < H2; j++) {
                    const int k = HLEN - 1 - (2 * j + off);
                    s1 = fma_t(wa[OFF0 + m + j], f.a[k], s1);
                    s2 = fma_t(wd[OFF0 + m + j], f.b[k], s2);
                }
                res[q / NV][q % NV] = s1 + s2;
            }
        }
    } else {
#pragma unroll
        for (int q = 0; q < PO; q++) {
            const int gp = q + SHIFT;  // g0 = it*PO is even -> parity and halving are item-local
            const int pl = gp >> 1, off = 1 - (gp & 1);
            T s1 = 0, s2 = 0;
#pragma unroll
            for (int j = 0; j < H2; j++) {
                const int k = HLEN - 1 - (2 * j + off);
                s1 = fma_t(wa[OFF0 + pl + j], f.a[k], s1);
                s2 = fma_t(wd[OFF0 + pl + j], f.b[k], s2);
            }
            res[q / NV][q % NV] = s1 + s2;
        }
    }
}

template <typename T, int NV2, typename V>
__device__ __forceinline__ void inv1d_store(T* dst, int g0, int nout, bool vec_ok, const V (&res)[2])
{
    constexpr int NV = NV2;
    if (vec_ok && g0 + 2 * NV <= nout) {
        reinterpret_cast<V*>(dst + g0)[0] = res[0];
        reinterpret_cast<V*>(dst + g0)[1] = res[1];
    } else {
#pragma unroll
        for (int q = 0; q < 2 * NV; q++)
            if (g0 + q < nout) dst[g0 + q] = res[q / NV][q % NV];
    }
}

template <typename T, int NV2, typename V>
__device__ __forceinline__ void inv1d_store_g(T* dst, int g0, int nout, bool vec_ok, const V (&res)[2])  // (to the output row in global memory)
{
    constexpr int NV = NV2;
    if (vec_ok && g0 + 2 * NV <= nout) {
        st_row(reinterpret_cast<V*>(dst + g0), res[0]);
        st_row(reinterpret_cast<V*>(dst + g0) + 1, res[1]);
    } else {
#pragma unroll
        for (int q = 0; q < 2 * NV; q++)
            if (g0 + q < nout) dst[g0 + q] = res[q / NV][q % NV];
    }
}

template <typename T, int HLEN>
__global__ __launch_bounds__(256) void k_inv1d_fused(T* __restrict__ out_img, Bands1D<T> b, Taps2<T> f)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using G = Inv1DGeom<T, HLEN>;
    using V = typename Vec16<T>::type;
    constexpr int NV = G::NV, PO = G::PO, HL = G::HL, HR = G::HR;
    const int Nc = b.n[0];
    const size_t row = blockIdx.x;
    const int be = G::buf_elems(b.n[1]);
    // three buffers of n1 (+halo) elements: approximation in, detail in, output.  The last level writes the
    // image row straight to HBM, so no LDS buffer ever holds more than n1 samples.
    T* a = reinterpret_cast<T*>(smem) + HL;
    T* sd = a + be;
    T* o = sd + be;

    auto stage = [&](T* dst, const T* src, int n) {
        const bool vec = ((n % NV) == 0) && ((reinterpret_cast<uintptr_t>(src) & 15) == 0);
        if (vec) {
            for (int i = threadIdx.x; i < n / NV; i += 256) reinterpret_cast<V*>(dst)[i] = reinterpret_cast<const V*>(src)[i];
        } else {
            for (int i = threadIdx.x; i < n; i += 256) dst[i] = src[i];
        }
        // halo straight from global (wrapped positions): no extra barrier needed before the fill
        for (int k = threadIdx.x; k < HL + HR; k += 256) {
            const int idx = k < HL ? k - HL : n + (k - HL);
            dst[idx] = src[wrap_per(idx, n)];
        }
    };
    stage(a, b.p[0] + row * (size_t)b.n[b.nlev], b.n[b.nlev]);

    for (int lev = b.nlev; lev >= 1; lev--) {
        const int nin = b.n[lev], nout = b.n[lev - 1];
        stage(sd, b.p[lev] + row * (size_t)nin, nin);
        lds_barrier();
        const bool last = lev == 1;
        T* g = out_img + row * (size_t)Nc;
        const bool vec_ok = ((nout % NV) == 0) && ((reinterpret_cast<uintptr_t>(g) & 15) == 0);
        const int items = (nout + PO - 1) / PO;
        for (int it = threadIdx.x; it < items; it += 256) {
            V res[2];
            inv1d_item<T, HLEN>(a, sd, it, f, res);
            if (last) inv1d_store_g<T, NV, V>(g, it * PO, nout, vec_ok, res);
            else inv1d_store<T, NV, V>(o, it * PO, nout + HR - PO, true, res);  // spill of a partial item lands in halo cells (refilled)
        }
        if (last) break;
        lds_barrier();
        fill_halo<T, false>(o, nout, HL, HR);
        // (the barrier at the top of the next iteration, after staging d, publishes the halo)
        T* t = a;  // the output becomes the next level's approximation
        a = o;
        o = t;
    }
}

// -------------------------------------------------------------------------------------------------
// inverse, persistent + prefetching form (the fast path): every band of the NEXT signal is fetched into
// registers while the current one is reconstructed, so no level ever waits on HBM.  Register slots are fixed
// per level (compile-time indices): A_L: 1, D1: 4, D2: 2, D3..D6: 1 each (x 256 threads x 16 bytes), which
// covers rows up to 4096 chunks... i.e. n1 <= 4096 elements f32 and at most 6 levels; anything else runs
// k_inv1d_fused above.  The level loop is unrolled over the level number for the same reason.
// -------------------------------------------------------------------------------------------------
constexpr int kInvMaxLev = 6;
__host__ __device__ constexpr int inv_cap(int lev) { return lev == 1 ? 4 : (lev == 2 ? 2 : 1); }  // lev 0 = A_L
__host__ __device__ constexpr int inv_slot0(int lev)
{  // first register slot of level `lev`: [A | D1 D1 D1 D1 | D2 D2 | D3 | D4 | D5 | D6]
    return lev == 0 ? 0 : (lev == 1 ? 1 : (lev == 2 ? 5 : 4 + lev));
}
constexpr int kInvSlots = 11;

template <typename T, int HLEN>
__global__ __launch_bounds__(256) void k_inv1d_fused_pf(T* __restrict__ out_img, Bands1D<T> b, int Nr, Taps2<T> f)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using G = Inv1DGeom<T, HLEN>;
    using V = typename Vec16<T>::type;
    constexpr int NV = G::NV, PO = G::PO, HL = G::HL, HR = G::HR;
    const int Nc = b.n[0];
    const int L = b.nlev;
    const int be = G::buf_elems(b.n[1]);
    T* const buf0 = reinterpret_cast<T*>(smem) + HL;
    T* const sd = buf0 + be;
    T* const buf1 = sd + be;

    V pre[kInvSlots];
    // unconditional, clamped, compile-time slots (see k_fwd1d_fused): levels beyond L re-read level L (harmless)
#define PDWT_ISSUE_BANDS(ROW)                                                                          \
    sfor<kInvMaxLev + 1>([&](auto LV_) {                                                               \
        constexpr int lv_ = decltype(LV_)::value;                                                      \
        const int lc_ = lv_ == 0 ? 0 : (lv_ < L ? lv_ : L);                                            \
        const int n_ = lv_ == 0 ? b.n[L] : b.n[lc_];                                                   \
        const V* src_ = reinterpret_cast<const V*>(b.p[lc_] + (size_t)(ROW) * (size_t)n_);            \
        sfor<inv_cap(lv_)>([&](auto K_) {                                                              \
            constexpr int k_ = decltype(K_)::value;                                                    \
            pre[inv_slot0(lv_) + k_] = ld_row<4>(src_ + min((int)threadIdx.x + 256 * k_, n_ / NV - 1));            \
        });                                                                                            \
    });

    size_t row = blockIdx.x;
    PDWT_ISSUE_BANDS(row < (size_t)Nr ? row : (size_t)Nr - 1)

    for (; row < (size_t)Nr; row += gridDim.x) {
        T* a = buf0;
        T* o = buf1;
        // A_L -> a
        if ((int)threadIdx.x < b.n[L] / NV) reinterpret_cast<V*>(a)[threadIdx.x] = pre[0];
        sfor<kInvMaxLev>([&](auto LI_) {
            constexpr int lev = kInvMaxLev - decltype(LI_)::value;  // 6, 5, ..., 1
            if (lev == 1 || lev <= L) {  // (level 1 always exists: keeps its prefetch issue unconditional)
                const int nin = b.n[lev], nout = b.n[lev - 1];
                // detail band of this level: registers -> LDS
                sfor<inv_cap(lev)>([&](auto K_) {
                    constexpr int k = decltype(K_)::value;
                    if ((int)threadIdx.x + 256 * k < nin / NV) reinterpret_cast<V*>(sd)[threadIdx.x + 256 * k] = pre[inv_slot0(lev) + k];
                });
                if constexpr (lev == 1) {  // all prefetched registers consumed: fetch the next signal during level 1
                    const size_t nr = row + gridDim.x;
                    PDWT_ISSUE_BANDS(nr < (size_t)Nr ? nr : (size_t)Nr - 1)
                }
                lds_barrier();
                if (lev == L) fill_halo<T, false>(a, nin, HL, HR);
                fill_halo<T, false>(sd, nin, HL, HR);
                lds_barrier();
                const bool last = lev == 1;
                T* g = out_img + row * (size_t)Nc;
                const int items = (nout + PO - 1) / PO;
                const bool vec_ok = (nout % NV) == 0;
                for (int it = threadIdx.x; it < items; it += 256) {
                    V res[2];
                    inv1d_item<T, HLEN>(a, sd, it, f, res);
                    if (last) inv1d_store_g<T, NV, V>(g, it * PO, nout, vec_ok, res);
                    else inv1d_store<T, NV, V>(o, it * PO, nout + HR - PO, true, res);
                }
                lds_barrier();
                if (!last) {
                    fill_halo<T, false>(o, nout, HL, HR);
                    T* t = a;
                    a = o;
                    o = t;
                }
            }
        });
    }
#undef PDWT_ISSUE_BANDS
}

// -------------------------------------------------------------------------------------------------
// inverse, TWO buffers instead of three (cf. k_fwd1d_fused_ip): a level's outputs wait in registers until every thread has read its
// windows of `a` and `sd`, then overwrite `a`.  Double precision at 8192 samples: 66 KiB instead of 99 -> two workgroups per CU; the
// prefetch slots are CAP times those of k_inv1d_fused_pf (CAP = 2: D1 8, D2 4, the others 2 chunks per thread).
// -------------------------------------------------------------------------------------------------
template <int CAP> __host__ __device__ constexpr int inv_cap_x(int lev) { return CAP * inv_cap(lev); }
template <int CAP> __host__ __device__ constexpr int inv_slot0_x(int lev) { return CAP * inv_slot0(lev); }

template <typename T, int HLEN, int MAXIT, int CAP>
__global__ __launch_bounds__(256, 2) void k_inv1d_fused_ip(T* __restrict__ out_img, Bands1D<T> b, int Nr, Taps2<T> f)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using G = Inv1DGeom<T, HLEN>;
    using V = typename Vec16<T>::type;
    constexpr int NV = G::NV, PO = G::PO, HL = G::HL, HR = G::HR;
    const int Nc = b.n[0];
    const int L = b.nlev;
    const int be = G::buf_elems(b.n[1]);
    T* const a = reinterpret_cast<T*>(smem) + HL;
    T* const sd = a + be;

    V pre[CAP * kInvSlots];
#define PDWT_ISSUE_BANDS_IP(ROW)                                                                       \
    sfor<kInvMaxLev + 1>([&](auto LV_) {                                                               \
        constexpr int lv_ = decltype(LV_)::value;                                                      \
        const int lc_ = lv_ == 0 ? 0 : (lv_ < L ? lv_ : L);                                            \
        const int n_ = lv_ == 0 ? b.n[L] : b.n[lc_];                                                   \
        const V* src_ = reinterpret_cast<const V*>(b.p[lc_] + (size_t)(ROW) * (size_t)n_);            \
        sfor<inv_cap_x<CAP>(lv_)>([&](auto K_) {                                                       \
            constexpr int k_ = decltype(K_)::value;                                                    \
            pre[inv_slot0_x<CAP>(lv_) + k_] = ld_row<4>(src_ + min((int)threadIdx.x + 256 * k_, n_ / NV - 1));     \
        });                                                                                            \
    });

    size_t row = blockIdx.x;
    PDWT_ISSUE_BANDS_IP(row < (size_t)Nr ? row : (size_t)Nr - 1)

    for (; row < (size_t)Nr; row += gridDim.x) {
        sfor<inv_cap_x<CAP>(0)>([&](auto K_) {
            constexpr int k = decltype(K_)::value;
            if ((int)threadIdx.x + 256 * k < b.n[L] / NV) reinterpret_cast<V*>(a)[threadIdx.x + 256 * k] = pre[inv_slot0_x<CAP>(0) + k];
        });
        sfor<kInvMaxLev>([&](auto LI_) {
            constexpr int lev = kInvMaxLev - decltype(LI_)::value;  // 6, 5, ..., 1
            if (lev == 1 || lev <= L) {
                const int nin = b.n[lev], nout = b.n[lev - 1];
                sfor<inv_cap_x<CAP>(lev)>([&](auto K_) {
                    constexpr int k = decltype(K_)::value;
                    if ((int)threadIdx.x + 256 * k < nin / NV) reinterpret_cast<V*>(sd)[threadIdx.x + 256 * k] = pre[inv_slot0_x<CAP>(lev) + k];
                });
                if constexpr (lev == 1) {
                    const size_t nr = row + gridDim.x;
                    PDWT_ISSUE_BANDS_IP(nr < (size_t)Nr ? nr : (size_t)Nr - 1)
                }
                lds_barrier();
                if (lev == L) fill_halo<T, false>(a, nin, HL, HR);
                fill_halo<T, false>(sd, nin, HL, HR);
                lds_barrier();
                const int items = (nout + PO - 1) / PO;
                if constexpr (lev == 1) {  // the image row: straight to HBM
                    T* g = out_img + row * (size_t)Nc;
                    const bool vec_ok = (nout % NV) == 0;
                    for (int it = threadIdx.x; it < items; it += 256) {
                        V res[2];
                        inv1d_item<T, HLEN>(a, sd, it, f, res);
                        inv1d_store_g<T, NV, V>(g, it * PO, nout, vec_ok, res);
                    }
                    lds_barrier();  // (the next row's A_L overwrites `a`)
                } else {
                    V keep[MAXIT][2];  // (host: items <= MAXIT * 256 at level 2, hence at every parked level)
                    sfor<MAXIT>([&](auto Kk) {
                        constexpr int kk = decltype(Kk)::value;
                        const int it = threadIdx.x + 256 * kk;
                        if (it < items) inv1d_item<T, HLEN>(a, sd, it, f, keep[kk]);
                        __builtin_amdgcn_sched_barrier(0);
                    });
                    lds_barrier();  // every window of `a` and `sd` has been read
                    sfor<MAXIT>([&](auto Kk) {
                        constexpr int kk = decltype(Kk)::value;
                        const int it = threadIdx.x + 256 * kk;
                        if (it < items) inv1d_store<T, NV, V>(a, it * PO, nout + HR - PO, true, keep[kk]);
                    });
                    lds_barrier();
                    fill_halo<T, false>(a, nout, HL, HR);
                    // (the barrier after the next level's detail staging publishes the halo)
                }
            }
        });
    }
#undef PDWT_ISSUE_BANDS_IP
}

// =================================================================================================
// host side
// =================================================================================================
#define PDWT_CHECK_LAUNCH() PDWT_HIP_TRY(hipGetLastError())

template <typename T>
static bool fill_bands(Bands1D<T>& b, T** c, const pdwt_info& w)
{
    if (w.nlevels > kMaxLev1D) return false;
    b.nlev = w.nlevels;
    b.n[0] = w.Nc;
    for (int l = 1; l <= w.nlevels; l++) b.n[l] = div2(b.n[l - 1]);
    for (int l = 0; l <= w.nlevels; l++) {
        if (!c[l]) return false;
        b.p[l] = c[l];
    }
    return true;
}

template <typename K>
static int set_lds(K kernel, size_t bytes)
{
    if (bytes > 64 * 1024) return lds_opt_in_ptr((const void*)kernel);  // (a driver call the first time only, never on the steady-state enqueue path)
    return PDWT_OK;
}

// Rows whose buffers need up to 80 KB leave room for two workgroups per CU.  Longer rows (double precision at 8192 samples: 99 KB forward) go to
// the per-level kernels: with ONE 4-wave workgroup per CU these kernels are slower than the 3.75x traffic they save (tools/dwt1d_budget.py,
// knob dwt1d_lds_kb = 158: 8192 x 8192 float64 sym8 L4 1.03 ms against 0.85 per level, db20 L3 1.78 against 1.00; float32 rows of 16K / 32K
// samples 0.525 / 0.500 against 0.543 / 0.505 -- a wash)
static size_t lds_budget_1d() { return (size_t)knob(KN_DWT1D_LDS_KB) * 1024; }

template <typename T, int HLEN>
static int launch_fwd(const T* in, T** c, const pdwt_info& w, const Taps2<T>& f)
{
    Bands1D<T> b;
    if (!fill_bands(b, c, w)) return 1;
    using G = Fwd1DGeom<T, HLEN>;
    // a single level never writes the second buffer (its approximation goes straight to HBM)
    const size_t lds = ((size_t)G::buf_elems(w.Nc) + (w.nlevels > 1 ? (size_t)G::buf_elems(b.n[1]) : 0)) * sizeof(T);
    constexpr int NVh = Vec16<T>::N;
    if (lds > lds_budget_1d()) {
        // the one-buffer form (double precision only: float rows of that size are a wash, see lds_budget_1d) when IT fits the budget
        if constexpr (sizeof(T) == 8 && HLEN <= 20) {  // (longer banks are not instantiated: they would never be launched, see below)
            constexpr int MAXIT = 8, KPRE = 16;
            const size_t lds1 = (size_t)G::buf_elems(w.Nc) * sizeof(T);
            const int items1 = (b.n[1] + G::PO - 1) / G::PO;
            // (banks of more than 20 taps: the window of an item no longer fits next to the prefetch registers -- scratch spills -- per-level kernels)
            if (knob(KN_DWT1D_F64) == 1 && w.nlevels > 1 && lds1 <= lds_budget_1d() && items1 <= MAXIT * 256 && (w.Nc % NVh) == 0 && ((uintptr_t)in & 15) == 0 &&
                (w.Nc / NVh) <= KPRE * 256) {
                auto k = k_fwd1d_fused_ip<T, HLEN, MAXIT, KPRE>;
                if (set_lds(k, lds1) != PDWT_OK) return PDWT_EHIP;
                const int per_cu = std::max(1, std::min(8, (int)((160 * 1024) / (lds1 + 512))));
                const int grid = w.Nr < 256 * per_cu ? w.Nr : 256 * per_cu;
                KTimer kt(K_ANA_ROWS, true);
                PDWT_LAUNCH_KT(kt, k, dim3(grid), dim3(256), lds1, in, b, w.Nr, f);
                PDWT_CHECK_LAUNCH();
                return PDWT_OK;
            }
        }
        return 1;
    }
    const bool pre = (w.Nc % NVh) == 0 && ((uintptr_t)in & 15) == 0 && (w.Nc / NVh) <= kPre1D * 256;
    auto k = pre ? k_fwd1d_fused<T, HLEN, true> : k_fwd1d_fused<T, HLEN, false>;
    if (set_lds(k, lds) != PDWT_OK) return PDWT_EHIP;
    // persistent workgroups: as many as fit the chip at once (256 CUs x LDS-limited residency), each walks rows
    const int per_cu = (int)((160 * 1024) / (lds + 512)) > 8 ? 8 : (int)((160 * 1024) / (lds + 512));
    const int grid = w.Nr < 256 * per_cu ? w.Nr : 256 * per_cu;
    KTimer kt(K_ANA_ROWS, true);
    PDWT_LAUNCH_KT(kt, k, dim3(grid), dim3(256), lds, in, b, w.Nr, f);
    PDWT_CHECK_LAUNCH();
    return PDWT_OK;
}

template <typename T, int HLEN>
static int launch_inv(T* out, T** c, const pdwt_info& w, const Taps2<T>& f)
{
    Bands1D<T> b;
    if (!fill_bands(b, c, w)) return 1;
    using G = Inv1DGeom<T, HLEN>;
    // a single level writes the image row straight to HBM: no output buffer in LDS
    const size_t lds = (w.nlevels > 1 ? 3 : 2) * (size_t)G::buf_elems(b.n[1]) * sizeof(T);
    constexpr int NVh = Vec16<T>::N;
    if (lds > lds_budget_1d()) {
        if constexpr (sizeof(T) == 8) {
            constexpr int MAXIT = 4, CAP = 2;
            const size_t lds2 = 2 * (size_t)G::buf_elems(b.n[1]) * sizeof(T);
            bool ok = knob(KN_DWT1D_F64) == 1 && w.nlevels > 1 && w.nlevels <= kInvMaxLev && lds2 <= lds_budget_1d() && (w.Nc % NVh) == 0 && ((uintptr_t)out & 15) == 0 &&
                      (b.n[1] + G::PO - 1) / G::PO <= MAXIT * 256;
            for (int l = 0; l <= w.nlevels && ok; l++) {
                const int n = l == 0 ? b.n[w.nlevels] : b.n[l];
                ok = (n % NVh) == 0 && n / NVh <= 256 * inv_cap_x<CAP>(l) && ((uintptr_t)b.p[l] & 15) == 0 && n >= NVh;
            }
            if (ok) {
                auto k = k_inv1d_fused_ip<T, HLEN, MAXIT, CAP>;
                if (set_lds(k, lds2) != PDWT_OK) return PDWT_EHIP;
                const int per_cu = std::max(1, std::min(8, (int)((160 * 1024) / (lds2 + 512))));
                const int grid = w.Nr < 256 * per_cu ? w.Nr : 256 * per_cu;
                KTimer kt(K_SYN_ROWS, true);
                PDWT_LAUNCH_KT(kt, k, dim3(grid), dim3(256), lds2, out, b, w.Nr, f);
                PDWT_CHECK_LAUNCH();
                return PDWT_OK;
            }
        }
        return 1;
    }
    bool pf = w.nlevels <= kInvMaxLev && (w.Nc % NVh) == 0 && ((uintptr_t)out & 15) == 0;
    for (int l = 0; l <= w.nlevels && pf; l++) {
        const int n = l == 0 ? b.n[w.nlevels] : b.n[l];
        pf = (n % NVh) == 0 && n / NVh <= 256 * inv_cap(l) && ((uintptr_t)b.p[l] & 15) == 0 && n >= NVh;
    }
    KTimer kt(K_SYN_ROWS, true);
    if (pf) {
        auto k = k_inv1d_fused_pf<T, HLEN>;
        if (set_lds(k, lds) != PDWT_OK) return PDWT_EHIP;
        const int per_cu = (int)((160 * 1024) / (lds + 512)) > 8 ? 8 : (int)((160 * 1024) / (lds + 512));
        const int grid = w.Nr < 256 * per_cu ? w.Nr : 256 * per_cu;
        PDWT_LAUNCH_KT(kt, k, dim3(grid), dim3(256), lds, out, b, w.Nr, f);
    } else {
        auto k = k_inv1d_fused<T, HLEN>;
        if (set_lds(k, lds) != PDWT_OK) return PDWT_EHIP;
        PDWT_LAUNCH_KT(kt, k, dim3(w.Nr), dim3(256), lds, out, b, f);
    }
    PDWT_CHECK_LAUNCH();
    return PDWT_OK;
}

#ifndef PDWT_1D_HLENS  // (a diagnostic build may restrict the instantiated lengths)
#define PDWT_1D_HLENS(X) X(4) X(6) X(8) X(10) X(12) X(14) X(16) X(18) X(20) X(22) X(24) X(26) X(28) X(30) X(32) X(34) X(36) X(38) X(40)
#endif

template <typename T>
int fwd1d_fused(const T* in, T** c, const pdwt_info& w, const Taps2<T>& f)
{
    if (w.Nc < 64) return 1;
    switch (w.hlen) {
#define X(H) \
    case H: return launch_fwd<T, H>(in, c, w, f);
        PDWT_1D_HLENS(X)
#undef X
        default: return 1;
    }
}
template <typename T>
int inv1d_fused(T* out, T** c, const pdwt_info& w, const Taps2<T>& f)
{
    if (w.Nc < 64) return 1;
    switch (w.hlen) {
#define X(H) \
    case H: return launch_inv<T, H>(out, c, w, f);
        PDWT_1D_HLENS(X)
#undef X
        default: return 1;
    }
}

template int fwd1d_fused<float>(const float*, float**, const pdwt_info&, const Taps2<float>&);
template int inv1d_fused<float>(float*, float**, const pdwt_info&, const Taps2<float>&);
#ifndef PDWT_1D_FLOAT_ONLY
template int fwd1d_fused<double>(const double*, double**, const pdwt_info&, const Taps2<double>&);
template int inv1d_fused<double>(double*, double**, const pdwt_info&, const Taps2<double>&);
#endif

}  // namespace PDWT_1D_VARIANT
}  // namespace pdwt
