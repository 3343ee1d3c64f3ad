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

int f64_store_taps_fwd(const Taps2<double>& f, int hlen, double* taps_dev)
{
    hipLaunchKernelGGL(k_f64_store_taps, dim3(1), dim3(64), 0, stream(), f, hlen, taps_dev);
    PDWT_CHECK_LAUNCH();
    return PDWT_OK;
}

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

// =================================================================================================
// inverse level: column synthesis in registers (rings of H2 coefficient rows of A, H, V, D), row synthesis through LDS
// =================================================================================================
// Reference order (src/separable.cu:246-328): columns first -- t1 = IL_y(A) + IH_y(H), t2 = IL_y(V) + IH_y(D) -- then rows
// out = IL_x(t1) + IH_x(t2).  SURVEY A-2 with H2 = hlen/2 taps per output, C = H2/2, SHIFT = 1 - (H2 & 1):
//   window position p (coefficient rows p-C .. p-C+H2-1)  ->  output rows 2p-SHIFT (tap parity 1) and 2p+1-SHIFT (parity 0),
//   the same along x: coefficient column c -> output columns 2c-SHIFT, 2c+1-SHIFT from t columns c-C .. c-C+H2-1.
// A thread owns ONE coefficient column: rings of its A, H, V, D samples (4 x (H2-1+UNR) doubles), one new coefficient row per
// step; the step's two rows of (t1, t2) go to LDS as 16-byte pairs, one barrier, then every thread whose whole window lies
// inside the workgroup's 256 columns reads it (H2 aligned 16-byte reads per row) and emits its two output columns of both rows.
// Taps per window position j: { IL[h-2-2j], IL[h-1-2j], IH[h-2-2j], IH[h-1-2j] } (parity 1 / parity 0), by scalar loads per
// section as in the forward kernel.  Each output is (sum over the IL branch) + (sum over the IH branch), both ascending in j:
// the order of the two-pass kernels and of the oracle.
__global__ void k_f64_store_taps_inv(Taps2<double> f, int hlen, double* __restrict__ dst)
{
    const int j = threadIdx.x;
    if (j < hlen / 2) {
        dst[4 * j + 0] = f.a[hlen - 2 - 2 * j];
        dst[4 * j + 1] = f.a[hlen - 1 - 2 * j];
        dst[4 * j + 2] = f.b[hlen - 2 - 2 * j];
        dst[4 * j + 3] = f.b[hlen - 1 - 2 * j];
    }
}

int f64_store_taps_inv(const Taps2<double>& f, int hlen, double* taps_dev)
{
    hipLaunchKernelGGL(k_f64_store_taps_inv, dim3(1), dim3(64), 0, stream(), f, hlen, taps_dev);
    PDWT_HIP_TRY(hipGetLastError());
    return PDWT_OK;
}

template <int HLEN, int NSEC, int UNR>
__global__ __launch_bounds__(kTW, 2) void k_inv2d_f64fused(const double* __restrict__ cA, const double* __restrict__ cH,
                                                            const double* __restrict__ cV, const double* __restrict__ cD,
                                                            double* __restrict__ out, int Nri, int Nci, int NP, const double* __restrict__ taps)
{
    constexpr int H2 = HLEN / 2, C = H2 / 2, SHIFT = (H2 & 1) ? 0 : 1;
    constexpr int TPS = H2 / NSEC;          // window positions per section
    static_assert(H2 % NSEC == 0, "whole sections");
    constexpr int TWV = kTW - (H2 - 1);     // threads of a workgroup whose window is complete = coefficient columns per tile
    constexpr int RS = H2 - 1 + UNR;        // ring slots: slot 0 = oldest row at the start of a body, step U appends slot H2-1+U
    extern __shared__ __attribute__((aligned(16))) double lds[];  // [2 buffers][2 rows][kTW] of (t1, t2)
    const int tid = threadIdx.x;
    const int Nro = 2 * Nri, Nco = 2 * Nci;
    const int c0 = blockIdx.x * TWV;                 // first coefficient column the tile produces outputs for
    const int cc = c0 - C + tid;                     // this thread's coefficient column (unwrapped)
    const int ccw = wrapi(cc, Nci);
    // thread tid reads the window tid .. tid+H2-1 of the workgroup's t columns = coefficient columns cc .. cc+H2-1 = the window
    // of coefficient column co = cc + C: it produces the outputs of column co
    const int co = cc + C;
    const bool produces = (tid < TWV) && (co < Nci);
    const unsigned uq1 = (unsigned)wrapi(2 * co - SHIFT, Nco), uq0 = (unsigned)wrapi(2 * co + 1 - SHIFT, Nco);  // output columns
    const int p0 = blockIdx.y * NP;                  // first window position of the chunk
    const int np = min(NP, Nri - p0);
    if (np <= 0) return;
    const int nsteps = np + H2 - 1;                  // the first H2-1 steps only fill the rings
    // uniform row offset (scalar unit) + per-lane 32-bit column offset: no 64-bit vector address arithmetic per access
    auto grow = [&](int s) { return (size_t)wrapi(p0 - C + min(s, nsteps - 1), Nri) * Nci; };
    const unsigned ucc = (unsigned)ccw;

    double ra[RS], rh[RS], rv[RS], rd[RS];
#pragma unroll
    for (int k = 0; k < RS; k++) ra[k] = rh[k] = rv[k] = rd[k] = 0.0;
    // rows 0 .. H2-2 straight into the rings
#pragma unroll
    for (int k = 0; k < H2 - 1; k++) {
        const size_t o = grow(k);
        ra[k] = (cA + o)[ucc];
        rh[k] = (cH + o)[ucc];
        rv[k] = (cV + o)[ucc];
        rd[k] = (cD + o)[ucc];
    }
    // Memory pipeline: the UNR coefficient rows of a body are loaded straight into their ring slots at the START of the body
    // (4 x UNR eight-byte loads per lane in flight, no staging registers): with one row of prefetch the kernel ran at the
    // bytes-in-flight limit (2 workgroups x 8 KB per CU: 412 us at C5 level 1).
    auto load_body = [&](int sb) {
        static_for<UNR>([&](auto UU) {
            constexpr int U = decltype(UU)::value;
            const size_t o = grow(H2 - 1 + sb + U);
            ra[H2 - 1 + U] = (cA + o)[ucc];
            rh[H2 - 1 + U] = (cH + o)[ucc];
            rv[H2 - 1 + U] = (cV + o)[ucc];
            rd[H2 - 1 + U] = (cD + o)[ucc];
        });
    };

    ctaps_t tbase = (ctaps_t)taps;
    static_assert((UNR * 2 * NSEC) % 2 == 0, "tap buffers in phase at the end of a body");
    double tp1l[2][TPS], tp0l[2][TPS], tp1h[2][TPS], tp0h[2][TPS];  // taps of the current / next section: IL parity 1, 0; IH parity 1, 0
#pragma unroll
    for (int jj = 0; jj < TPS; jj++) {
        tp1l[0][jj] = tbase[4 * jj];
        tp0l[0][jj] = tbase[4 * jj + 1];
        tp1h[0][jj] = tbase[4 * jj + 2];
        tp0h[0][jj] = tbase[4 * jj + 3];
    }

    auto step = [&](auto UU, int sb) {
        constexpr int U = decltype(UU)::value;
        const int s = sb + U;            // this step pushes coefficient row H2-1+s and completes window position p0+s
        const int buf = s & 1;
        // (the coefficient row H2-1+s of this step is already in ring slot H2-1+U: load_body)
        // ---- column synthesis: (t1, t2) of this thread's column for the two output rows of window position p0+s ----
        double a1, h1, v1, d1, a0, h0, v0, d0;  // parity 1 (first row) / parity 0 (second row)
        static_for<NSEC>([&](auto SS) {
            constexpr int sec = decltype(SS)::value;
            constexpr int gsec = U * 2 * NSEC + sec;
            constexpr int cur = gsec & 1, nxt = cur ^ 1;
            constexpr int nsec = (sec + 1) % NSEC;  // (the row synthesis below starts again at section 0)
            ctaps_t tp = tbase;
            if constexpr (sec == 0) asm volatile("" : "+s"(tp), "+v"(ra[U]));
            else asm volatile("" : "+s"(tp), "+v"(a1), "+v"(h1), "+v"(v1), "+v"(d1), "+v"(a0), "+v"(h0), "+v"(v0), "+v"(d0));
#pragma unroll
            for (int jj = 0; jj < TPS; jj++) {
                tp1l[nxt][jj] = tp[4 * (nsec * TPS + jj)];
                tp0l[nxt][jj] = tp[4 * (nsec * TPS + jj) + 1];
                tp1h[nxt][jj] = tp[4 * (nsec * TPS + jj) + 2];
                tp0h[nxt][jj] = tp[4 * (nsec * TPS + jj) + 3];
            }
            static_for<TPS>([&](auto JJ) {
                constexpr int jj = decltype(JJ)::value;
                constexpr int slot = U + sec * TPS + jj;
                if constexpr (sec == 0 && jj == 0) {  // fma(x, t, 0) == x * t bit for bit: no zero-initialisation moves
                    a1 = ra[slot] * tp1l[cur][jj];
                    h1 = rh[slot] * tp1h[cur][jj];
                    v1 = rv[slot] * tp1l[cur][jj];
                    d1 = rd[slot] * tp1h[cur][jj];
                    a0 = ra[slot] * tp0l[cur][jj];
                    h0 = rh[slot] * tp0h[cur][jj];
                    v0 = rv[slot] * tp0l[cur][jj];
                    d0 = rd[slot] * tp0h[cur][jj];
                } else {
                    a1 = __builtin_fma(ra[slot], tp1l[cur][jj], a1);
                    h1 = __builtin_fma(rh[slot], tp1h[cur][jj], h1);
                    v1 = __builtin_fma(rv[slot], tp1l[cur][jj], v1);
                    d1 = __builtin_fma(rd[slot], tp1h[cur][jj], d1);
                    a0 = __builtin_fma(ra[slot], tp0l[cur][jj], a0);
                    h0 = __builtin_fma(rh[slot], tp0h[cur][jj], h0);
                    v0 = __builtin_fma(rv[slot], tp0l[cur][jj], v0);
                    d0 = __builtin_fma(rd[slot], tp0h[cur][jj], d0);
                }
            });
        });
        double2* const rowsb = reinterpret_cast<double2*>(lds) + (size_t)buf * 2 * kTW;
        rowsb[tid] = make_double2(a1 + h1, v1 + d1);        // row 2p-SHIFT
        rowsb[kTW + tid] = make_double2(a0 + h0, v0 + d0);  // row 2p+1-SHIFT
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // (LDS only: stores and prefetch loads stay in flight)
        // ---- row synthesis: the thread's two output columns of both rows, window = t columns tid .. tid+H2-1 ----
        double x1l[2], x1h[2], x0l[2], x0h[2];  // [row]; column parity 1 / 0; IL / IH branch
        static_for<NSEC>([&](auto SS) {
            constexpr int sec = decltype(SS)::value;
            constexpr int gsec = U * 2 * NSEC + NSEC + sec;
            constexpr int cur = gsec & 1, nxt = cur ^ 1;
            constexpr int nsec = (sec + 1) % NSEC;
            ctaps_t tp = tbase;
            if constexpr (sec == 0) asm volatile("" : "+s"(tp), "+v"(a1));
            else asm volatile("" : "+s"(tp), "+v"(x1l[0]), "+v"(x1h[0]), "+v"(x0l[0]), "+v"(x0h[0]), "+v"(x1l[1]), "+v"(x1h[1]), "+v"(x0l[1]), "+v"(x0h[1]));
#pragma unroll
            for (int jj = 0; jj < TPS; jj++) {
                tp1l[nxt][jj] = tp[4 * (nsec * TPS + jj)];
                tp0l[nxt][jj] = tp[4 * (nsec * TPS + jj) + 1];
                tp1h[nxt][jj] = tp[4 * (nsec * TPS + jj) + 2];
                tp0h[nxt][jj] = tp[4 * (nsec * TPS + jj) + 3];
            }
#pragma unroll
            for (int jj = 0; jj < TPS; jj++) {
#pragma unroll
                for (int r = 0; r < 2; r++) {
                    const double2 t = rowsb[r * kTW + tid + sec * TPS + jj];  // (t1, t2) at window position sec*TPS+jj
                    if (sec == 0 && jj == 0) {
                        x1l[r] = t.x * tp1l[cur][jj];
                        x1h[r] = t.y * tp1h[cur][jj];
                        x0l[r] = t.x * tp0l[cur][jj];
                        x0h[r] = t.y * tp0h[cur][jj];
                    } else {
                        x1l[r] = __builtin_fma(t.x, tp1l[cur][jj], x1l[r]);
                        x1h[r] = __builtin_fma(t.y, tp1h[cur][jj], x1h[r]);
                        x0l[r] = __builtin_fma(t.x, tp0l[cur][jj], x0l[r]);
                        x0h[r] = __builtin_fma(t.y, tp0h[cur][jj], x0h[r]);
                    }
                }
            }
        });
        if (s >= 0 && produces) {
            // window position p = p0 + s -> output rows 2p-SHIFT, 2p+1-SHIFT; coefficient column co -> output columns 2co-SHIFT, +1
            const int p = p0 + s;
#pragma unroll
            for (int r = 0; r < 2; r++) {
                double* orow = out + (size_t)wrapi(2 * p - SHIFT + r, Nro) * Nco;
                double* q1 = orow + uq1;
                double* q0 = orow + uq0;
                const double o1 = x1l[r] + x1h[r], o0 = x0l[r] + x0h[r];
                asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(q1), "v"(o1) : "memory");
                asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(q0), "v"(o0) : "memory");
            }
        }
    };

    // steps are numbered so that step s completes window position p0+s: the ring-filling rows were loaded above
    for (int sb = 0; sb < np; sb += UNR) {
        load_body(sb);
        bool fin = false;
        static_for<UNR>([&](auto UU) {
            if (!fin) {
                step(UU, sb);
                fin = (sb + decltype(UU)::value + 1 >= np);
            }
        });
#pragma unroll
        for (int k = 0; k < H2 - 1; k++) {
            ra[k] = ra[k + UNR];
            rh[k] = rh[k + UNR];
            rv[k] = rv[k + UNR];
            rd[k] = rd[k + UNR];
        }
    }
}

int inv2d_f64_fused(const double* cA, const double* cH, const double* cV, const double* cD, double* out, double* taps_dev, int nri, int nci,
                    int nro, int nco, int hlen, const Taps2<double>& f)
{
    if (knob(KN_F64_FUSED) != 1 || !taps_dev) return 1;
    if (hlen != 40) return 1;
    if (nro != 2 * nri || nco != 2 * nci || nci < kTW || nri < 4 * hlen) return 1;
    if ((long long)nro * nco < (long long)knob(KN_F64_FUSED_MIN) * knob(KN_F64_FUSED_MIN)) return 1;
    constexpr int H2 = 20;
    const int twv = kTW - (H2 - 1);
    const int tiles = idiv_up(nci, twv);
    int chunks = std::max(1, 512 / tiles);
    int NP = idiv_up(nri, chunks);
    if (NP < 2 * hlen) NP = 2 * hlen;
    chunks = idiv_up(nri, NP);
    hipLaunchKernelGGL(k_f64_store_taps_inv, dim3(1), dim3(64), 0, stream(), f, hlen, taps_dev);
    PDWT_CHECK_LAUNCH();
    const size_t lds = ((size_t)2 * 2 * kTW + H2) * 2 * sizeof(double);  // (+ H2 entries: the window reads of the non-producing threads)
    KTimer kt(K_INV2D_F64);
    hipLaunchKernelGGL((k_inv2d_f64fused<40, 5, 4>), dim3(tiles, chunks), dim3(kTW), lds, stream(), cA, cH, cV, cD, out, nri, nci, NP,
                       (const double*)taps_dev);
    PDWT_CHECK_LAUNCH();
    return PDWT_OK;
}

}  // namespace pdwt
