// dwt_f64_fused.hip -- one 2-D DWT level per launch for LONG double-precision banks (db20: 40 taps), row pass and column
// pass fused: the half-width temporaries of the two-pass form (k_ana_rows_tr + k_ana_cols_ring_tr, 2.67x the compulsory
// traffic at C5) never exist.
//
// Reference code replaced: w_kern_forward_pass1 + w_kern_forward_pass2 (src/separable.cu:91-176) of one iteration of
// w_forward_separable (:179-209).
//
// Forward.  A workgroup (4 waves) owns 256 output columns (one per lane) and walks DOWN a chunk of rows, two input rows
// (= one output row) per iteration:
//   * the two input rows (512 columns + hlen-2 of halo, periodic) are staged in LDS; every lane reads its hlen-sample window
//     as aligned 16-byte pairs (0.125 ds_read_b128 per FMA over both passes) and runs the ROW pass: (lo, hi) of its column
//     for both rows;
//   * (lo, hi) enter a register ring of hlen rows (compile-time slots: the body is unrolled over the ring period) and the
//     COLUMN pass emits one row of A, H, V, D from registers.
//   * Taps: 2*hlen doubles do not fit a wave's SGPRs (160 for db20) and v_readlane broadcasts cost an instruction per two FMAs
//     (cols_ring.inc).  Here an iteration is cut into NSEC sections of hlen/NSEC window positions; a section loads ITS taps
//     with scalar loads (constant address space: s_load) through a laundered pointer -- the compiler can neither hoist them
//     out of the section nor keep them live across sections -- and both passes consume them from SGPRs: no VALU
//     instruction besides the FMAs themselves.  The s_load latency of a section is covered by the other wave of the SIMD.
// Per-sample arithmetic: taps in ascending window position, one FMA per tap, rows before columns -- the reference's and the
// oracle's order: bit-identical to the two-pass kernels.
#include "dwt_f64_fused.hpp"

#include <algorithm>
#include <type_traits>

#include "stream_dev.hpp"

namespace pdwt {

constexpr int kTW = 256;  // output columns per workgroup = threads per workgroup

typedef const double __attribute__((address_space(4))) * ctaps_t;

// taps in consumption order: T[j] = { L[hlen-1-j], H[hlen-1-j] } for window position j (SURVEY A-1: out[i] = sum_j x[2i-c+j] F[hlen-1-j])
__global__ void k_f64_store_taps(Taps2<double> f, int hlen, double* __restrict__ dst)
{
    const int j = threadIdx.x;
    if (j < hlen) {
        dst[2 * j] = f.a[hlen - 1 - j];
        dst[2 * j + 1] = f.b[hlen - 1 - j];
    }
}

// UNR = iterations in the unrolled body.  The ring slots must be compile-time constants, so either the body covers a whole
// ring period (UNR = HLEN/2: 20 iterations = 78 KB of code for db20, more than the 64 KB instruction cache -> every wave
// streams its code from L2, measured 5 us per iteration instead of ~1) or the ring is SHIFTED down by 2*UNR rows after a short
// body (UNR = 2: 8 KB of code, 38 v_mov_b64 per iteration = +12% VALU, ring of HLEN-2+2*UNR rows).
template <int HLEN, int NSEC, int UNR>
__global__ __launch_bounds__(kTW, 2) void k_fwd2d_f64fused(const double* __restrict__ in, double* __restrict__ cA, double* __restrict__ cH,
                                                            double* __restrict__ cV, double* __restrict__ cD, int Nr, int Nc, int RO,
                                                            const double* __restrict__ taps)
{
    constexpr int C = HLEN / 2 - 1;
    constexpr int H2 = HLEN / 2;
    constexpr int TPS = HLEN / NSEC;  // window positions per section
    static_assert(HLEN % NSEC == 0 && TPS % 2 == 0, "sections hold whole 16-byte pairs of the window");
    constexpr int LW = 2 * kTW + HLEN;  // LDS row width in doubles (>= 2*kTW + HLEN - 2, even)
    extern __shared__ __attribute__((aligned(16))) double lds[];  // [2 buffers][2 rows][LW]
    const int tid = threadIdx.x;
    const int Nc2 = Nc >> 1, Nr2 = Nr >> 1;
    const int i0 = blockIdx.x * kTW;
    const int oc = i0 + tid;
    const bool col_ok = oc < Nc2;
    const int y0 = blockIdx.y * RO;
    const int nout = min(RO, Nr2 - y0);
    if (nout <= 0) return;
    const int niter = nout + H2 - 1;  // the first H2-1 iterations only warm the ring up
    // LDS column j <-> input column cbase + j; the window of output column oc starts at input column 2*oc - C = LDS column 2*tid
    const int cbase = 2 * i0 - C;
    // staging: thread t loads LDS columns 2t, 2t+1 and (t < H2-1) the halo columns 2*kTW + 2t, +1 -- periodic in the image
    const int gc0 = wrapi(cbase + 2 * tid, Nc), gc1 = wrapi(cbase + 2 * tid + 1, Nc);
    const bool has_halo = tid < H2 - 1;
    const int gh0 = wrapi(cbase + 2 * kTW + 2 * tid, Nc), gh1 = wrapi(cbase + 2 * kTW + 2 * tid + 1, Nc);
    auto grow = [&](int r) { return (size_t)wrapi(2 * y0 - C + r, Nr) * Nc; };  // chunk-local input row -> offset

    constexpr int RS = HLEN - 2 + 2 * UNR;  // ring[0] = oldest row at the start of a body; iteration U appends slots HLEN-2+2U, +1
    double ring[RS][2];                      // (lo, hi) of the lane's column
#pragma unroll
    for (int k = 0; k < RS; k++) ring[k][0] = ring[k][1] = 0.0;

    // rows of iteration 0 straight into buffer 0
    double s[2][2], sh[2][2];
    auto load_rows = [&](int u) {
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const size_t o = grow(min(2 * u + r, 2 * niter - 1));
            s[r][0] = in[o + gc0];
            s[r][1] = in[o + gc1];
            if (has_halo) {
                sh[r][0] = in[o + gh0];
                sh[r][1] = in[o + gh1];
            }
        }
    };
    auto store_rows = [&](int buf) {
#pragma unroll
        for (int r = 0; r < 2; r++) {
            double* row = lds + (size_t)(buf * 2 + r) * LW;
            *reinterpret_cast<double2*>(row + 2 * tid) = make_double2(s[r][0], s[r][1]);
            if (has_halo) *reinterpret_cast<double2*>(row + 2 * kTW + 2 * tid) = make_double2(sh[r][0], sh[r][1]);
        }
    };
    load_rows(0);
    store_rows(0);
    __syncthreads();

    ctaps_t tbase = (ctaps_t)taps;
    static_assert((UNR * NSEC) % 2 == 0, "the two tap buffers must be back in phase at the end of a body");
    double tl[2][TPS], th[2][TPS];  // taps (SGPRs) of the current and of the next section
#pragma unroll
    for (int jj = 0; jj < TPS; jj++) {
        tl[0][jj] = tbase[2 * jj];
        th[0][jj] = tbase[2 * jj + 1];
    }

    auto iteration = [&](auto UU, int ub) {
        constexpr int U = decltype(UU)::value;  // iteration within the ring period: new rows go to slots 2U, 2U+1
        const int u = ub + U;
        const int buf = u & 1;
        const bool emit = u >= H2 - 1;
        if (u + 1 < niter) load_rows(u + 1);  // in flight while this iteration computes
        const double* row0 = lds + (size_t)(buf * 2) * LW + 2 * tid;
        const double* row1 = row0 + LW;
        double lo0 = 0.0, hi0 = 0.0, lo1 = 0.0, hi1 = 0.0;  // row pass accumulators of the two new rows
        double a = 0.0, h = 0.0, v = 0.0, d = 0.0;            // column pass accumulators of output row u-(H2-1)
        static_for<NSEC>([&](auto SS) {
            constexpr int sec = decltype(SS)::value;
            constexpr int gsec = U * NSEC + sec;          // section counter within the unrolled body
            constexpr int cur = gsec & 1, nxt = cur ^ 1;  // tap buffers: this section's / the next one's
            constexpr int nsec = (sec + 1) % NSEC;        // (the last section of an iteration prefetches section 0 of the next)
            // Ordering point + prefetch of the NEXT section's taps: scalar loads through a laundered pointer.  The accumulators
            // pass through the same statement, so these loads cannot move above the previous section's FMAs (at most two
            // sections' taps are ever live) and they have this whole section's FMAs to land.
            ctaps_t tp = tbase;
            asm volatile("" : "+s"(tp), "+v"(lo0), "+v"(hi0), "+v"(lo1), "+v"(hi1), "+v"(a), "+v"(h), "+v"(v), "+v"(d));
#pragma unroll
            for (int jj = 0; jj < TPS; jj++) {
                tl[nxt][jj] = tp[2 * (nsec * TPS + jj)];
                th[nxt][jj] = tp[2 * (nsec * TPS + jj) + 1];
            }
            // row pass, window positions [sec*TPS, (sec+1)*TPS)
#pragma unroll
            for (int jj = 0; jj < TPS; jj += 2) {
                const double2 x0 = *reinterpret_cast<const double2*>(row0 + sec * TPS + jj);
                const double2 x1 = *reinterpret_cast<const double2*>(row1 + sec * TPS + jj);
                lo0 = __builtin_fma(x0.x, tl[cur][jj], lo0);
                hi0 = __builtin_fma(x0.x, th[cur][jj], hi0);
                lo1 = __builtin_fma(x1.x, tl[cur][jj], lo1);
                hi1 = __builtin_fma(x1.x, th[cur][jj], hi1);
                lo0 = __builtin_fma(x0.y, tl[cur][jj + 1], lo0);
                hi0 = __builtin_fma(x0.y, th[cur][jj + 1], hi0);
                lo1 = __builtin_fma(x1.y, tl[cur][jj + 1], lo1);
                hi1 = __builtin_fma(x1.y, th[cur][jj + 1], hi1);
            }
            if constexpr (sec == NSEC - 1) {
                // the new rows are complete: they are window positions HLEN-2, HLEN-1 of this iteration's output row
                ring[HLEN - 2 + 2 * U][0] = lo0;
                ring[HLEN - 2 + 2 * U][1] = hi0;
                ring[HLEN - 1 + 2 * U][0] = lo1;
                ring[HLEN - 1 + 2 * U][1] = hi1;
            }
            // column pass, same window positions: position j <-> ring slot 2U + j
            if (emit) {
                static_for<TPS>([&](auto JJ) {
                    constexpr int jj = decltype(JJ)::value;
                    constexpr int slot = 2 * U + sec * TPS + jj;
                    a = __builtin_fma(ring[slot][0], tl[cur][jj], a);
                    h = __builtin_fma(ring[slot][0], th[cur][jj], h);
                    v = __builtin_fma(ring[slot][1], tl[cur][jj], v);
                    d = __builtin_fma(ring[slot][1], th[cur][jj], d);
                });
            }
        });
        // The next rows go to LDS BEFORE this iteration's results are stored, and the stores are inline asm: hipcc counts loads and
        // stores in one vmcnt and, with both kinds pending, waits for vmcnt(0) -- i.e. for the stores to be acknowledged by memory
        // (5 us per iteration).  This way the wait before the ds_writes only sees the loads, and the stores drain during the
        // next iteration.
        if (u + 1 < niter) store_rows(buf ^ 1);
        if (emit && col_ok) {
            const size_t o = (size_t)(y0 + u - (H2 - 1)) * Nc2 + oc;
            double *pa = cA + o, *ph = cH + o, *pv = cV + o, *pd = cD + o;
            asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(pa), "v"(a) : "memory");
            asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(ph), "v"(h) : "memory");
            asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(pv), "v"(v) : "memory");
            asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(pd), "v"(d) : "memory");
        }
        // LDS only: __syncthreads() would also drain vmcnt, i.e. wait for this iteration's four stores to be acknowledged by
        // memory (measured: 5 us per iteration instead of ~1)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };

    for (int ub = 0; ub < niter; ub += UNR) {
        bool fin = false;
        static_for<UNR>([&](auto UU) {
            if (!fin) {
                iteration(UU, ub);
                fin = (ub + decltype(UU)::value + 1 >= niter);
            }
        });
        // the 2*UNR oldest rows are dead: shift the ring down
#pragma unroll
        for (int k = 0; k < HLEN - 2; k++) {
            ring[k][0] = ring[k + 2 * UNR][0];
            ring[k][1] = ring[k + 2 * UNR][1];
        }
    }
}

// =================================================================================================
// host side
// =================================================================================================
#define PDWT_CHECK_LAUNCH() PDWT_HIP_TRY(hipGetLastError())

int fwd2d_f64_fused(const double* in, double* cA, double* cH, double* cV, double* cD, double* taps_dev, int nr, int nc, int hlen,
                    const Taps2<double>& f)
{
    if (knob(KN_F64_FUSED) != 1 || !taps_dev) return 1;
    if (hlen != 40) return 1;                                     // instantiated lengths
    if ((nr & 1) || (nc & 1) || nc / 2 < kTW || nr / 2 < 4 * hlen) return 1;  // periodic wrap only; at least one full tile; chunks >> halo
    // the level must fill the machine with chunks that are long against their hlen-2 rows of ring warm-up: smaller levels are
    // a serial walk (measured 140-180 us per level below 4096^2 outputs against 60-120 for the two-pass kernels)
    if ((long long)nr * nc < (long long)knob(KN_F64_FUSED_MIN) * knob(KN_F64_FUSED_MIN)) return 1;
    const int nr2 = nr / 2, nc2 = nc / 2;
    const int tiles = idiv_up(nc2, kTW);
    // two workgroups per CU (2 waves per SIMD); a chunk recomputes hlen-2 rows of ring warm-up, so not shorter than 2*hlen output rows
    int chunks = std::max(1, 512 / tiles);
    int RO = idiv_up(nr2, chunks);
    if (RO < 2 * hlen) RO = 2 * hlen;
    chunks = idiv_up(nr2, RO);
    hipLaunchKernelGGL(k_f64_store_taps, dim3(1), dim3(64), 0, stream(), f, hlen, taps_dev);
    PDWT_CHECK_LAUNCH();
    const size_t lds = (size_t)2 * 2 * (2 * kTW + hlen) * sizeof(double);
    KTimer kt(K_FWD2D_F64);
    hipLaunchKernelGGL((k_fwd2d_f64fused<40, 5, 2>), dim3(tiles, chunks), dim3(kTW), lds, stream(), in, cA, cH, cV, cD, nr, nc, RO, (const double*)taps_dev);
    PDWT_CHECK_LAUNCH();
    return PDWT_OK;
}

int inv2d_f64_fused(const double*, const double*, const double*, const double*, double*, double*, int, int, int, int, int, const Taps2<double>&)
{
    return 1;  // (inverse: see below once built)
}

}  // namespace pdwt
