// dwt_lat.hip -- one 2-D DWT level per launch for ORTHOGONAL double-precision banks: row pass in direct form, column pass as a
// paraunitary LATTICE (tools/gen_lattice.py, lattice_table.inc).  Written for BASELINE config 5 (8192^2 float64 db20, 6 levels).
//
// Reference code replaced: w_kern_forward_pass1 + w_kern_forward_pass2 (src/separable.cu:91-176) of one iteration of
// w_forward_separable (:179-209), and w_kern_inverse_pass1 + w_kern_inverse_pass2 (:246-328) of one iteration of w_inverse_separable
// (:332-364), at DTYPE = double (src/filters.h:16-30, Makefile:36-39).
//
// Why.  The direct form costs 4K multiply-adds per (lo, hi) output pair of a 2K-tap bank, in each pass: 160 n FMAs for a level of n
// samples, and the level kernels of dwt_lds.hip (one FMA per tap, the reference's order) run C5's level 1 at 0.38-0.43 of the FP64
// vector peak with the LDS two-thirds busy -- in-core bound at 3.7 TB/s of traffic.  A paraunitary bank factors into K plane
// rotations separated by one-pair delays (lattice_table.inc): 2K + 2 multiply-adds per output pair, and the "window" becomes K - 1
// values of per-column state.  In the COLUMN direction that state lives in the registers of the thread that streams down its column:
// no ring in LDS, no window reads, half the arithmetic.  Along a row the recursion would run across lanes (or need a transposed tile
// per step), so the ROW pass stays in direct form, register-blocked over four outputs (23 16-byte LDS reads per 320 FMAs instead of
// 21 per 160).  Per output sample: 61 FP64 operations instead of 80, 0.09 LDS reads per operation instead of 0.27.
//
// Numerics.  Not the reference's summation order: the lattice reproduces the direct form to ~2e-15 of the signal's scale (each stage
// is a scaled rotation: backward stable; the table is derived in 120-digit arithmetic from the bank's own taps), far inside the
// 1e-10 the double-precision parity tests ask for (tests/test_gpu_parity.py::test_lattice_levels_vs_oracle).  The two passes also run
// in the opposite order on the inverse (rows first, then columns) -- separable, so the same operator.  Only banks with a table entry
// (exact tap match: db2..db20, sym9, coif1..5), even sizes and levels that fill the chip come here; everything else keeps dwt_lds.hip.
//
// Forward level (k_fwd2d_lat), a workgroup of 256 threads = 128 output columns, walking down a chunk, 8 input rows per step:
//   * staging: 8 rows x (256 + HLEN - 2) input samples by aligned 16-byte global loads into LDS (double-buffered, a step ahead);
//   * ROW pass (direct): thread = (input row, four adjacent output columns): 46-sample window = 23 aligned 16-byte LDS reads, 320 FMAs
//     (taps by scalar loads per section of 8 window positions, as in dwt_lds.hip); a wave holds four rows x sixteen column groups,
//     dealt so that every 16-lane LDS service group sees four rows (an odd number of 16-byte slots apart) x four neighbouring column
//     groups: all 64 banks, once; results (lo, hi) go to a second LDS buffer [plane][row][column];
//   * COLUMN pass (lattice) of the PREVIOUS step's 8 rows (one barrier per step): thread = (plane, column) holds the K - 1 delay
//     values of its column; four pair-steps per step run stage by stage (eight independent FMAs per stage); emits (A, H) or (V, D).
// Inverse level (k_inv2d_lat), the mirror image: coefficient rows of the four bands staged as (A, V) / (H, D) pairs, ROW synthesis in
// direct form (thread = (coefficient row, band pair, four adjacent coefficient columns) -> eight output samples), then the synthesis
// lattice down the output columns (thread = output column).
#include "dwt_lat.hpp"

#include <algorithm>
#include <atomic>
#include <cstring>
#include <mutex>
#include <vector>

#include "stream_dev.hpp"

namespace pdwt {

#ifndef PDWT_LAT_DIAG  // (diagnostic builds, tools/build_variant.sh: bit 0 = every step loads the chunk's first rows again (L2 hits), bit 1 = stores folded onto 8 rows)
#define PDWT_LAT_DIAG 0
#endif
#ifndef PDWT_LAT_PF  // steps the global loads run ahead of the step that consumes them
#define PDWT_LAT_PF 2
#endif
namespace {
#ifndef PDWT_LAT_NW  // waves per workgroup: 8 = ONE workgroup per CU (two waves per SIMD that advance in step), 4 = two workgroups per CU
#define PDWT_LAT_NW 4
#endif
constexpr int kNW = PDWT_LAT_NW;
constexpr int kNT = 64 * kNW;   // threads per workgroup
constexpr int kNCW = 32 * kNW;  // forward: output columns per workgroup; inverse: coefficient columns per workgroup
constexpr int kTPR = kNT / 8;   // forward staging: threads per row of the step
constexpr int kWGPerCU = kNW == 4 ? 2 : 1;
typedef double v2d __attribute__((ext_vector_type(2)));
typedef const double __attribute__((address_space(4))) * ctab_t;

// The coefficient table travels as the FIRST kernel argument (kernarg segment = constant memory: scalar loads, no staging launch).
// Sections of kSEC doubles (one s_load_dwordx16 each; two section buffers = 32 scalar registers -- with 16-double sections the two
// buffers alone were 64 of the 102 and the compiler parked 123 scalars in vector lanes, 195 v_readlane per step): the row-pass sections
// (4 window positions x 2 filters, resp. 2 window slots x 4 taps), then the lattice sections in the order the kernel consumes them.
#ifndef PDWT_LAT_SEC
#define PDWT_LAT_SEC 8
#endif
constexpr int kSEC = PDWT_LAT_SEC;
template <int NS>
struct LatTable {
    double t[kSEC * NS];
};
__device__ __forceinline__ ctab_t kernarg_tab() { return (ctab_t)__builtin_amdgcn_kernarg_segment_ptr(); }

__device__ __forceinline__ void xcd_tile(int strips, int& strip, int& chunk)
{  // workgroup id -> (strip, chunk): every XCD gets a contiguous run of the strip-major tile order (neighbouring strips share halo columns in L2)
    const int T = gridDim.x, w = blockIdx.x;
    const int x = w & 7, per = T >> 3, rem = T & 7;
    const int L = x * per + min(x, rem) + (w >> 3);
    strip = L % strips;
    chunk = L / strips;
}
// ---- hand-counted memory pipeline (stream_dev.hpp): hipcc ends every loop body with vmcnt(0) and sizes every wait for its worst incoming
// edge, so loads it schedules itself never run more than a fraction of a step ahead (measured: prefetch distances 1..4 all at the same time).
// Loads and stores are inline asm; the one consumer of a step's rows waits with an exact vmcnt(N).  Rules that keep N exact: every step
// issues the same VMEM instructions in the same order; a store of rows the chunk does not own runs with EXEC = 0 (it still takes its
// place in the in-order count: selfcheck.hip -- the dispatcher asks counted_waits_ok()); no VMEM instruction sits under a branch.
typedef unsigned long long lanemask_t;
// (strip, first row y0, rows nout) of this workgroup; `chunks` workgroups share a strip.  skew = 0: equal chunks of RO rows.  skew > 0 (two
// workgroups per CU): the issue arbiter serves the OLDER waves first -- the workgroup dispatched first on a CU runs at full speed, the
// other in what is left, and then finishes alone at one wave per SIMD (workgroup 0 of an evenly split level-1 launch ends after 163 of
// 300 us).  As in dwt_lds.hip, the rows of a strip are therefore dealt by WEIGHT: the first half of the dispatch order takes the odd
// chunks of every strip with weight skew, the second half the even chunks with weight 100 - skew (boundaries on multiples of 4 rows).
__device__ __forceinline__ void lat_tile(int strips, int chunks, int RO, int nrows, int skew, int& strip, int& y0, int& nout)
{
    if (skew > 0) {
        const int T = gridDim.x, E = strips * (chunks >> 1);
        const int late = (int)blockIdx.x >= E ? 1 : 0;
        const int wh = blockIdx.x - late * E, n = late ? T - E : E;
        const int x = wh & 7, per = n >> 3, rem = n & 7;
        const int P = x * per + min(x, rem) + (wh >> 3);
        strip = P % strips;
        const int chunk = 2 * (P / strips) + 1 - late;
        const int wa = skew, wb = 100 - skew;
        const int before = wb * ((chunk + 1) >> 1) + wa * (chunk >> 1), total = wb * ((chunks + 1) >> 1) + wa * (chunks >> 1);
        const int q = (nrows + 3) >> 2;
        const int g0 = (int)(((long long)q * before + (total >> 1)) / total);
        const int g1 = (int)(((long long)q * (before + (late ? wb : wa)) + (total >> 1)) / total);
        y0 = 4 * g0;
        nout = min(4 * g1, nrows) - y0;
    } else {
        int chunk;
        xcd_tile(strips, strip, chunk);
        y0 = chunk * RO;
        nout = min(RO, nrows - y0);
    }
}
#ifndef PDWT_LAT_NT  // (cache-policy experiments: bit 0 forward loads, bit 1 inverse loads, bit 2 forward stores, bit 3 inverse stores non-temporal)
#define PDWT_LAT_NT 0
#endif
template <bool NT>
__device__ __forceinline__ void asm_load(v2d& d, const char* p)
{
    if constexpr (NT) asm volatile("global_load_dwordx4 %0, %1, off nt" : "+v"(d) : "v"(p) : "memory");
    else asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(d) : "v"(p) : "memory");
}
template <bool NT>
__device__ __forceinline__ void st2_sv_m(double* b0, double* b1, unsigned boff, double v0, double v1, lanemask_t mask)
{
    lanemask_t saved;
    if constexpr (NT)
        asm volatile("s_and_saveexec_b64 %0, %6\n\tglobal_store_dwordx2 %1, %2, %4 nt\n\tglobal_store_dwordx2 %1, %3, %5 nt\n\ts_mov_b64 exec, %0"
                     : "=&s"(saved) : "v"(boff), "v"(v0), "v"(v1), "s"(b0), "s"(b1), "s"(mask) : "memory", "scc");
    else
        asm volatile("s_and_saveexec_b64 %0, %6\n\tglobal_store_dwordx2 %1, %2, %4\n\tglobal_store_dwordx2 %1, %3, %5\n\ts_mov_b64 exec, %0"
                     : "=&s"(saved) : "v"(boff), "v"(v0), "v"(v1), "s"(b0), "s"(b1), "s"(mask) : "memory", "scc");
}
// all lanes when 0 <= i < n (one unsigned compare on the scalar unit), none otherwise
__device__ __forceinline__ lanemask_t mask_in_range(int i, int n)
{
    lanemask_t m;
    asm("s_cmp_lt_u32 %1, %2\n\ts_cselect_b64 %0, -1, 0" : "=s"(m) : "s"(i), "s"(n) : "scc");
    return m;
}
// counted wait chosen at run time inside ONE asm statement (an if / else around two tied waits would make the load registers PHI values)
template <int N>
__device__ __forceinline__ void wait_sel5(bool counted, v2d& a, v2d& b, v2d& c, v2d& d, v2d& e)
{
    asm volatile("s_cmp_eq_u32 %5, 0\n\ts_cbranch_scc1 .Lws0_%=\n\ts_waitcnt vmcnt(%6)\n\ts_branch .Lws1_%=\n.Lws0_%=:\n\ts_waitcnt vmcnt(0)\n.Lws1_%=:"
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e) : "s"(__builtin_amdgcn_readfirstlane((int)counted)), "n"(N) : "memory", "scc");
}
template <int N>
__device__ __forceinline__ void wait_sel6(bool counted, v2d& a, v2d& b, v2d& c, v2d& d, v2d& e, v2d& f)
{
    asm volatile("s_cmp_eq_u32 %6, 0\n\ts_cbranch_scc1 .Lws0_%=\n\ts_waitcnt vmcnt(%7)\n\ts_branch .Lws1_%=\n.Lws0_%=:\n\ts_waitcnt vmcnt(0)\n.Lws1_%=:"
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) : "s"(__builtin_amdgcn_readfirstlane((int)counted)), "n"(N) : "memory", "scc");
}
// lane -> (row of the wave's four, column group of the wave's sixteen) such that every ds_read_b128 service group of 16 lanes --
// quads {0,3,5,6}, {1,2,4,7}, {8,11,13,14}, {9,10,12,15} (MI355X_MICROARCH.md, LDS) -- holds 4 rows x 4 neighbouring groups
__device__ __forceinline__ void lane_deal(int lane, int& r4, int& g16)
{
    const int quad = lane >> 2;
    r4 = (quad & 7) >> 1;
    const int gq = ((quad >> 3) << 1) | (__builtin_popcount(quad & 7) & 1);
    g16 = 4 * gq + (lane & 3);
}

template <int HLEN>
struct LatGeo {
    static constexpr int K = HLEN / 2, C = K - 1;
    static constexpr int NSEC = 2 * HLEN / kSEC;   // row-pass sections
    // lattice table, forward: M0[4], t_1 .. t_{K-1};  inverse: t_{K-1} .. t_1, then MI[4] inside ONE section (padded up when it would straddle)
    static constexpr int IMP = ((K - 1) / kSEC == (K + 2) / kSEC) ? K - 1 : (K - 1 + kSEC - 1) / kSEC * kSEC;
    static constexpr int NLS = (IMP + 4 + kSEC - 1) / kSEC;  // lattice sections (>= (K + 3 + kSEC - 1) / kSEC, the forward's count)
    static constexpr int NS = NSEC + NLS;
    // forward
    static constexpr int FSLOTS = kNCW + K + 1;                // 16-byte slots per staged row: 2 pad + (2 kNCW + HLEN - 2) samples + pad, odd
    static constexpr int FROW = FSLOTS * 16;                   // bytes
    static constexpr int FPAIRS = kNCW + K;                    // aligned global pairs per staged row
    static constexpr int FIN = 8 * FROW;                       // one input buffer (8 rows)
    static constexpr int FMID = 2 * 8 * kNCW * 8;              // one (lo, hi) buffer: [plane][row][column]
    static constexpr int FLDS = 2 * FIN + 2 * FMID;
    // inverse
    static constexpr int H2 = K, CI = K / 2, SHIFT = (K & 1) ? 0 : 1;
    static constexpr int ICOLS = kNCW + K;                     // coefficient columns staged per row (even)
    static constexpr int ISLOTS = ICOLS + 1;                   // (a, b) slots per staged row of a band pair, odd
    static constexpr int IROW = ISLOTS * 16;
    static constexpr int IIN = 2 * 4 * IROW;                   // one input buffer: [band pair][coefficient row 0..3]
    static constexpr int IMID = 2 * 4 * (2 * kNCW) * 8;        // one (u1, u2) buffer: [pair][row 0..3][output column]
    static constexpr int ILDS = 2 * IIN + 2 * IMID;
    static_assert(HLEN % 8 == 0 && (FSLOTS & 1) && (ISLOTS & 1) && !(ICOLS & 1), "geometry");
    static_assert((FSLOTS % 4) & 1 && (ISLOTS % 4) & 1, "row strides must be odd in units of four slots' worth of banks");
};

// scalar FMA helpers: one sample, an adjacent pair of table entries
__device__ __forceinline__ void fma2(double& a0, double& a1, double x, double t0, double t1)
{
    a0 = __builtin_fma(x, t0, a0);
    a1 = __builtin_fma(x, t1, a1);
}
__device__ __forceinline__ void mul2(double& a0, double& a1, double x, double t0, double t1)
{
    a0 = x * t0;
    a1 = x * t1;
}
// a - t * x with t in scalar registers: the negation as a source modifier (left to itself the compiler negates every t_i with an s_xor into
// a second scalar pair -- twice the scalar registers for the lattice coefficients, which is what tipped the forward kernel into lane spills)
__device__ __forceinline__ double fnma_s(double t, double x, double a)
{
    double r;
    asm("v_fma_f64 %0, -%1, %2, %3" : "=v"(r) : "s"(t), "v"(x), "v"(a));
    return r;
}
}  // namespace

// =================================================================================================
// forward level
// =================================================================================================
template <int HLEN>
__global__ __launch_bounds__(kNT, kWGPerCU) void k_fwd2d_lat(LatTable<LatGeo<HLEN>::NS> /*read through kernarg_tab()*/, const double* __restrict__ in,
                                                       double* __restrict__ cA, double* __restrict__ cH, double* __restrict__ cV,
                                                       double* __restrict__ cD, int Nr, int Nc, int RO, int strips, int skew, unsigned long long* probe, int probe_all)
{
    clock_probe_stamp(probe, 0, probe_all);
    using G = LatGeo<HLEN>;
    constexpr int K = G::K, C = G::C, NSEC = G::NSEC, NS = G::NS;
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Nc2 = Nc >> 1, Nr2 = Nr >> 1;
    int strip, y0, nout;
    lat_tile(strips, gridDim.x / strips, RO, Nr2, skew, strip, y0, nout);
    if (nout <= 0) return;
    const int i0 = strip * kNCW;
    const int nsteps = (C + nout + 3) >> 2;  // row-pass steps (4 pair-steps each); the lattice lags one step behind

    char* const in_lds = lds_raw;
    char* const mid_lds = lds_raw + 2 * G::FIN;

    // ---- staging role: thread -> (row of the step, aligned pairs k0 + 32 m); LDS column j <-> input column 2 i0 - C + j sits at byte 8 (j + 2)
    const int srow = tid / kTPR, k0 = tid % kTPR;
    const int cbase = 2 * i0 - C;  // odd
    unsigned gc[5];
#pragma unroll
    for (int m = 0; m < 5; m++) gc[m] = 8u * (unsigned)wrapi(cbase - 1 + 2 * (k0 + kTPR * m), Nc);
    const bool has5 = k0 + 4 * kTPR < G::FPAIRS;
    static_assert(G::FPAIRS <= 5 * kTPR && G::NSEC >= 10, "five aligned pairs per thread cover a staged row; VMEM issue slots at sections 0..8");
    int rnext = wrapi(2 * y0 - C, Nr);  // image row of the first row of the step to stage
    constexpr int PF = PDWT_LAT_PF, UB = (PF & 1) ? 2 * PF : PF;  // prefetch distance; steps per unrolled body (slot and section parity are constants)
    v2d st[PF][5];
#pragma unroll
    for (int sl = 0; sl < PF; sl++)
#pragma unroll
        for (int m = 0; m < 5; m++) st[sl][m] = v2d{0.0, 0.0};
    if (!has5) gc[4] = gc[3];  // (every lane issues the fifth load -- the count must not depend on the lane; only lanes with a fifth pair store it)
    // five loads per step, always (rows past the chunk are rows of the image all the same), issued ONE AT A TIME between the sections of
    // the step (load_one): a wave that meets a full memory pipeline stalls at the VMEM instruction with its FMAs behind it, and a burst of
    // 5 loads + 8 stores per wave at the step boundary did exactly that to every wave of the chip at once (level 1: 224 us without
    // traffic, 215 us of traffic, 300 us together)
    const char* lp = nullptr;
    auto load_begin = [&]() {
        int row = rnext + srow;
        row = row >= Nr ? row - Nr : row;
        lp = reinterpret_cast<const char*>(in + (size_t)row * Nc);
        if constexpr (!(PDWT_LAT_DIAG & 1)) rnext += 8;
        rnext = rnext >= Nr ? rnext - Nr : rnext;
    };
    auto load_one = [&](auto SL, auto MM) { asm_load<(PDWT_LAT_NT & 1) != 0>(st[decltype(SL)::value][decltype(MM)::value], lp + gc[decltype(MM)::value]); };
    auto load_rows = [&](auto SL) {
        load_begin();
        static_for<5>([&](auto MM) { load_one(SL, MM); });
    };
    const int stage_off = srow * G::FROW + 16 * k0 + 8;
    auto store_rows = [&](auto SL, int buf) {
        constexpr int sl = decltype(SL)::value;
        char* b = in_lds + buf * G::FIN + stage_off;
#pragma unroll
        for (int m = 0; m < 4; m++) {
            *reinterpret_cast<double*>(b + 16 * kTPR * m) = st[sl][m].x;
            *reinterpret_cast<double*>(b + 16 * kTPR * m + 8) = st[sl][m].y;
        }
        if (has5) {
            *reinterpret_cast<double*>(b + 64 * kTPR) = st[sl][4].x;
            *reinterpret_cast<double*>(b + 64 * kTPR + 8) = st[sl][4].y;
        }
    };

    // ---- row-pass role: (row of the step, group of four output columns)
    int r4, g16;
    lane_deal(lane, r4, g16);
    const int rrow = 4 * (w & 1) + r4, g = 16 * (w >> 1) + g16;
    const int row_rd = rrow * G::FROW + 64 * g + 16;              // window of output column 4g starts at LDS column 8g
    const int mid_wr = rrow * (kNCW * 8) + 32 * g;                // plane 0; plane 1 at + 8 * kNCW * 8
    // ---- column-pass role: (plane, column)
    const int plane = w / (kNW / 2), cc = tid & (kNCW - 1);
    const int mid_rd = plane * (8 * kNCW * 8) + 8 * cc;
    double* const outL = plane ? cV : cA;
    double* const outH = plane ? cD : cH;
    const unsigned ocol = 8u * (unsigned)(i0 + cc);
    double pu[4] = {0.0, 0.0, 0.0, 0.0}, pv[4] = {0.0, 0.0, 0.0, 0.0};  // the previous step's outputs, stored during this one
    // running store row: band row y0 + so of the chunk (so < 0 during the warm-up: the pointers then sit in front of the chunk and the stores run
    // with EXEC = 0); one row further after every store pair
    int so = -8 - C;
    const long long spitch = (long long)Nc2 * 8;
    double* spL = reinterpret_cast<double*>(reinterpret_cast<char*>(outL) + ((long long)y0 + so) * spitch);
    double* spH = reinterpret_cast<double*>(reinterpret_cast<char*>(outH) + ((long long)y0 + so) * spitch);
    double D[K];  // D[i], i = 1..K-1: the delayed second branch in front of stage i
#pragma unroll
    for (int i = 0; i < K; i++) D[i] = 0.0;

    ctab_t tbase = kernarg_tab();
    double tb[2][kSEC];
#pragma unroll
    for (int j = 0; j < kSEC; j++) tb[0][j] = tbase[j];

    // the first lattice pass (step 0) runs on rows that do not exist: zeros keep its state clean
    {
        v2d z = {0.0, 0.0};
        char* m1 = mid_lds + G::FMID + tid * 64;
#pragma unroll
        for (int q = 0; q < 4; q++) *reinterpret_cast<v2d*>(m1 + 16 * q) = z;
    }
    static_for<PF>([&](auto SL) { load_rows(SL); });  // rows of steps 0 .. PF-1
    wait_sel5<5 * (PF - 1)>(true, st[0][0], st[0][1], st[0][2], st[0][3], st[0][4]);
    store_rows(std::integral_constant<int, 0>{}, 0);
    __syncthreads();

    auto step = [&](auto PAR, auto SLOT, int s) {
        constexpr int par = decltype(PAR)::value;  // parity of the section counter at the start of this step
        constexpr int slot = decltype(SLOT)::value;  // = s % PF: the slot whose rows (step s) went to LDS at the end of step s-1
        const int buf = s & 1;
        load_begin();  // rows of step s+PF: their five loads go out at sections 0, 2, 4, 6, 8; the previous step's four store pairs at 1, 3, 5, 7
        const char* xr = in_lds + buf * G::FIN + row_rd;
        const char* mr = mid_lds + (buf ^ 1) * G::FMID + mid_rd;  // the previous step's (lo, hi) rows
        double lo[4], hi[4];
        double u[4], v[4];
        double ev[4], od[4];
        constexpr int NM = kSEC / 4;  // 16-byte window slots a row-pass section advances by
        v2d P[HLEN / 2 + 3];
        auto ldP = [&](auto MM) {
            constexpr int m = decltype(MM)::value;
            P[m] = *reinterpret_cast<const v2d*>(xr + 16 * m);
        };
        static_for<NM + 3>([&](auto MM) { ldP(MM); });
#pragma unroll
        for (int q = 0; q < 4; q++) {
            ev[q] = *reinterpret_cast<const double*>(mr + (2 * q) * (kNCW * 8));
            od[q] = *reinterpret_cast<const double*>(mr + (2 * q + 1) * (kNCW * 8));
        }
        static_for<NS>([&](auto SS) {
            constexpr int sec = decltype(SS)::value;
            constexpr int cur = (par + sec) & 1, nxt = cur ^ 1;
            constexpr int nsec = (sec + 1) % NS;
            // ordering point: what this section consumes has been asked for one section ago; the one wait it needs sits here
            ctab_t tp = tbase;
            if constexpr (sec == 0)
                asm volatile("" : "+s"(tp), "+v"(P[0]), "+v"(P[NM + 2]), "+s"(tb[cur][0]), "+s"(tb[cur][kSEC - 1]));
            else if constexpr (sec < NSEC)
                asm volatile(""
                             : "+s"(tp), "+v"(lo[0]), "+v"(hi[0]), "+v"(lo[1]), "+v"(hi[1]), "+v"(lo[2]), "+v"(hi[2]), "+v"(lo[3]), "+v"(hi[3]),
                               "+v"(P[NM * sec + 3]), "+v"(P[NM * sec + NM + 2]), "+s"(tb[cur][0]), "+s"(tb[cur][kSEC - 1]));
            else if constexpr (sec == NSEC)
                asm volatile("" : "+s"(tp), "+v"(lo[0]), "+v"(hi[0]), "+v"(lo[1]), "+v"(hi[1]), "+v"(lo[2]), "+v"(hi[2]), "+v"(lo[3]), "+v"(hi[3]),
                             "+v"(ev[0]), "+v"(od[0]), "+v"(ev[3]), "+v"(od[3]), "+s"(tb[cur][0]), "+s"(tb[cur][kSEC - 1]));
            else  // (the lattice sections stay in their order: the pair-steps' running values pass through the statement)
                asm volatile("" : "+s"(tp), "+v"(u[0]), "+v"(v[0]), "+v"(u[1]), "+v"(v[1]), "+v"(u[2]), "+v"(v[2]), "+v"(u[3]), "+v"(v[3]), "+s"(tb[cur][0]),
                             "+s"(tb[cur][kSEC - 1]));
#pragma unroll
            for (int j = 0; j < kSEC; j++) tb[nxt][j] = (PDWT_LAT_DIAG & 32) ? tb[cur][j] : tp[kSEC * nsec + j];  // (bit 5: no scalar loads -- wrong results, timing only)
            if constexpr (sec + 1 < NSEC) static_for<NM>([&](auto MM) {
                constexpr int pm = NM * (sec + 1) + 3 + decltype(MM)::value;
                if constexpr (PDWT_LAT_DIAG & 16) P[pm] = P[pm - NM - 3];  // (diagnostic: no window reads after the first -- wrong results, timing only)
                else ldP(std::integral_constant<int, pm>{});
            });
            if constexpr (sec < 10 && !(sec & 1)) load_one(SLOT, std::integral_constant<int, sec / 2>{});
            if constexpr (sec < 8 && (sec & 1)) {
                // outputs of the step BEFORE the previous one's rows (kept in pu / pv): chunk-local pair qq -> band row y0 + qq - C; outside the chunk: EXEC = 0
                constexpr int q = sec / 2;
                st2_sv_m<(PDWT_LAT_NT & 4) != 0>(spL, spH, ocol, pu[q], pv[q], mask_in_range(so, nout));
                so++;
                if constexpr (!(PDWT_LAT_DIAG & 2)) {
                    spL = reinterpret_cast<double*>(reinterpret_cast<char*>(spL) + spitch);
                    spH = reinterpret_cast<double*>(reinterpret_cast<char*>(spH) + spitch);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (sec < NSEC) {
                // row pass: window position j = (kSEC/2) sec + 2 m + h meets sample P[NM sec + m + a].{x, y} for output a
                static_for<NM>([&](auto MM) {
                    constexpr int m = decltype(MM)::value;
                    static_for<2>([&](auto HH) {
                        constexpr int h = decltype(HH)::value;
                        static_for<4>([&](auto AA) {
                            constexpr int a = decltype(AA)::value;
                            const v2d p = P[NM * sec + m + a];
                            const double x = h ? p.y : p.x;
                            if constexpr (sec == 0 && m == 0 && h == 0) mul2(lo[a], hi[a], x, tb[cur][0], tb[cur][1]);
                            else fma2(lo[a], hi[a], x, tb[cur][4 * m + 2 * h], tb[cur][4 * m + 2 * h + 1]);
                        });
                    });
                });
            } else {
                // column lattice on the previous step's rows: four pair-steps, stage by stage; table positions: M0 at 0..3, t_i at 3 + i
                constexpr int ls = sec - NSEC;
                constexpr int i_first = (kSEC * ls - 3 > 1) ? kSEC * ls - 3 : 1;
                constexpr int i_last = (kSEC * ls + kSEC - 4 < K - 1) ? kSEC * ls + kSEC - 4 : K - 1;
                if constexpr (ls == 0) {
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        u[q] = __builtin_fma(tb[cur][1], od[q], tb[cur][0] * ev[q]);
                        v[q] = __builtin_fma(tb[cur][3], od[q], tb[cur][2] * ev[q]);
                    }
                }
                if constexpr (i_last >= i_first) {
                    // pair-step by pair-step through the section's stages: the value that enters the delay in front of stage i at pair-step q
                    // is read back at pair-step q+1 (W), the last one stays in D[i] for the next step -- written by the FMA that makes it,
                    // after D[i]'s old value was consumed at pair-step 0: no register copies (the stage-major order needed 19 per step)
                    constexpr int NI = i_last - i_first + 1;
                    double W[NI];
                    static_for<4>([&](auto QQ) {
                        constexpr int q = decltype(QQ)::value;
                        static_for<NI>([&](auto II) {
                            constexpr int k = decltype(II)::value, i = i_first + k;
                            const double t = tb[cur][3 + i - kSEC * ls];
                            const double vin = v[q];
                            const double vd = q == 0 ? D[i] : W[k];
                            if constexpr (q == 3) D[i] = vin;
                            else W[k] = vin;
                            const double nu = fnma_s(t, vd, u[q]);
                            v[q] = __builtin_fma(t, u[q], vd);
                            u[q] = nu;
                        });
                    });
                }
            }
        });
        // this step's (lo, hi) rows
        {
            char* mw = mid_lds + buf * G::FMID + mid_wr;
            *reinterpret_cast<v2d*>(mw) = v2d{lo[0], lo[1]};
            *reinterpret_cast<v2d*>(mw + 16) = v2d{lo[2], lo[3]};
            *reinterpret_cast<v2d*>(mw + 8 * kNCW * 8) = v2d{hi[0], hi[1]};
            *reinterpret_cast<v2d*>(mw + 8 * kNCW * 8 + 16) = v2d{hi[2], hi[3]};
        }
        {
            // the rows of step s+1 (asked for PF steps ago; the last of their loads was the last VMEM instruction of its step).  Younger VMEM
            // instructions at this point: PF-1 whole steps of 13.  The first PF-1 steps consume the prologue's loads: vmcnt(0).
            constexpr int ns = (slot + 1) % PF;
            wait_sel5<13 * (PF - 1)>(s >= PF - 1, st[ns][0], st[ns][1], st[ns][2], st[ns][3], st[ns][4]);
            store_rows(std::integral_constant<int, ns>{}, buf ^ 1);
        }
#pragma unroll
        for (int q = 0; q < 4; q++) pu[q] = u[q], pv[q] = v[q];  // stored during the next step
        if constexpr (PDWT_LAT_DIAG & 4) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (diagnostic: no barrier -- wrong results, timing only)
        else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };

    for (int sb = 0; sb <= nsteps + 1; sb += UB) {  // (row pass: steps 0 .. nsteps-1; lattice: one step behind; stores: two steps behind)
        bool fin = false;
        static_for<UB>([&](auto UU) {
            constexpr int uu = decltype(UU)::value;
            if (!fin) {
                step(std::integral_constant<int, (uu * NS) & 1>{}, std::integral_constant<int, uu % PF>{}, sb + uu);
                fin = sb + uu + 1 > nsteps + 1;
            }
        });
    }
    clock_probe_stamp(probe, 1, probe_all);
}

// =================================================================================================
// inverse level
// =================================================================================================
template <int HLEN>
__global__ __launch_bounds__(kNT, kWGPerCU) void k_inv2d_lat(LatTable<LatGeo<HLEN>::NS> /*read through kernarg_tab()*/, const double* __restrict__ cA,
                                                       const double* __restrict__ cH, const double* __restrict__ cV, const double* __restrict__ cD,
                                                       double* __restrict__ out, int Nri, int Nci, int NP, int strips, int skew, unsigned long long* probe, int probe_all)
{
    clock_probe_stamp(probe, 0, probe_all);
    using G = LatGeo<HLEN>;
    constexpr int K = G::K, C = G::C, NSEC = G::NSEC, NS = G::NS, CI = G::CI, SHIFT = G::SHIFT;
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Nro = 2 * Nri, Nco = 2 * Nci;
    int strip, m0, nm;
    lat_tile(strips, gridDim.x / strips, NP, Nri, skew, strip, m0, nm);
    // the chunk produces output pairs m0 .. m0 + nm - 1 (rows 2m + sig, 2m + sig + 1, sig = (K-1) & 1): pair m leaves the lattice when
    // coefficient row m + ceil(C/2) has entered it, K - 1 rows after the first one it depends on
    if (nm <= 0) return;
    const int c0 = strip * kNCW;
    constexpr int sig = C & 1;
    const int nsteps = (C + nm + 3) >> 2;  // row-synthesis steps (4 coefficient rows each); the lattice lags one step behind

    char* const in_lds = lds_raw;
    char* const mid_lds = lds_raw + 2 * G::IIN;

    // ---- staging role: task = (coefficient row of the step, band pair, two adjacent coefficient columns): tasks tid, tid + 256, tid + 512
    constexpr int TPR = G::ICOLS / 2;  // tasks per (row, pair)
    constexpr int NTASK = 4 * 2 * TPR;
    static_assert(NTASK <= 3 * kNT, "three staging tasks per thread cover a step");
    int t_row[3], t_pair[3];
    unsigned t_gc[3];
    int t_lds[3];
    bool t_ok[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int t = tid + kNT * k;
        t_ok[k] = t < NTASK;
        const int tt = t_ok[k] ? t : 0;
        const int rp = tt / TPR, cp2 = tt % TPR;
        t_row[k] = rp >> 1;
        t_pair[k] = rp & 1;
        t_gc[k] = 8u * (unsigned)wrapi(c0 - CI + 2 * cp2, Nci);  // even column: 16-byte aligned
        t_lds[k] = (t_pair[k] * 4 + t_row[k]) * G::IROW + 32 * cp2;  // [band pair][row]: a wave's four rows an odd number of slots apart
    }
    int pnext = wrapi(m0 + (C + 1) / 2 - C, Nri);  // coefficient row of the first row of the step to stage
    constexpr int PF = PDWT_LAT_PF, UB = (PF & 1) ? 2 * PF : PF;
    v2d sa[PF][3], sb[PF][3];
#pragma unroll
    for (int sl = 0; sl < PF; sl++)
#pragma unroll
        for (int k = 0; k < 3; k++) sa[sl][k] = sb[sl][k] = v2d{0.0, 0.0};
    // six loads per step, always (a thread without a third task loads task 0 again and does not store it), one per section 0 .. 5
    const char* lpa[3] = {nullptr, nullptr, nullptr};
    const char* lpb[3] = {nullptr, nullptr, nullptr};
    auto load_begin = [&]() {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            int row = pnext + t_row[k];
            row = row >= Nri ? row - Nri : row;
            const size_t ro = (size_t)row * Nci * 8 + t_gc[k];
            lpa[k] = reinterpret_cast<const char*>(t_pair[k] ? cH : cA) + ro;
            lpb[k] = reinterpret_cast<const char*>(t_pair[k] ? cD : cV) + ro;
        }
        if constexpr (!(PDWT_LAT_DIAG & 1)) pnext += 4;
        pnext = pnext >= Nri ? pnext - Nri : pnext;
    };
    auto load_one = [&](auto SL, auto MM) {
        constexpr int sl = decltype(SL)::value, m = decltype(MM)::value;
        if constexpr (m & 1) asm_load<(PDWT_LAT_NT & 2) != 0>(sb[sl][m / 2], lpb[m / 2]);
        else asm_load<(PDWT_LAT_NT & 2) != 0>(sa[sl][m / 2], lpa[m / 2]);
    };
    auto load_rows = [&](auto SL) {
        load_begin();
        static_for<6>([&](auto MM) { load_one(SL, MM); });
    };
    auto store_rows = [&](auto SL, int buf) {
        constexpr int sl = decltype(SL)::value;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            if (k < 2 || t_ok[k]) {
                char* b = in_lds + buf * G::IIN + t_lds[k];
                *reinterpret_cast<v2d*>(b) = v2d{sa[sl][k].x, sb[sl][k].x};
                *reinterpret_cast<v2d*>(b + 16) = v2d{sa[sl][k].y, sb[sl][k].y};
            }
        }
    };

    // ---- row-synthesis role: wave -> band pair (waves 0,1: (A, V) -> u1; waves 2,3: (H, D) -> u2), lane -> (row of four, group of four columns)
    int r4, g16;
    lane_deal(lane, r4, g16);
    const int bpair = w / (kNW / 2), g = 16 * (w % (kNW / 2)) + g16;
    const int row_rd = (bpair * 4 + r4) * G::IROW + 64 * g;      // window of coefficient column c0 + 4g starts at staged column 4g (= c0 + 4g - CI)
    const int mid_wr = bpair * (4 * 2 * kNCW * 8) + r4 * (2 * kNCW * 8);  // + 8 * (output column within the strip)
    // output columns of window position c: 2c - SHIFT (parity 1) and 2c + 1 - SHIFT (parity 0); strip-local, wrapped inside the strip's 256
    // (the strip's first output column 2 c0 - SHIFT belongs to the previous strip's last position only when SHIFT = 1: handled by the store)
    // ---- column-lattice role: thread = output column 2 c0 - SHIFT + tid
    const int oc = wrapi(2 * c0 - SHIFT + tid, Nco);
    const unsigned ocol = 8u * (unsigned)oc;
    const int mid_rd = 8 * tid;
    double pe[4] = {0.0, 0.0, 0.0, 0.0}, po[4] = {0.0, 0.0, 0.0, 0.0};  // the previous step's outputs, stored during this one
    int so = -8 - C;  // running output pair of the chunk (negative during the warm-up: EXEC = 0 stores in front of the chunk's rows)
    const long long spitch = (long long)Nco * 8;
    double* sp0 = reinterpret_cast<double*>(reinterpret_cast<char*>(out) + (2 * ((long long)m0 + so) + sig) * spitch);
    double A[K];  // A[i], i = 1..K-1: the delayed first branch behind stage i
#pragma unroll
    for (int i = 0; i < K; i++) A[i] = 0.0;

    ctab_t tbase = kernarg_tab();
    double tb[2][kSEC];
#pragma unroll
    for (int j = 0; j < kSEC; j++) tb[0][j] = tbase[j];
    {
        v2d z = {0.0, 0.0};
        char* m1 = mid_lds + G::IMID + tid * 64;
#pragma unroll
        for (int q = 0; q < 4; q++) *reinterpret_cast<v2d*>(m1 + 16 * q) = z;
    }
    static_for<PF>([&](auto SL) { load_rows(SL); });
    wait_sel6<6 * (PF - 1)>(true, sa[0][0], sb[0][0], sa[0][1], sb[0][1], sa[0][2], sb[0][2]);
    store_rows(std::integral_constant<int, 0>{}, 0);
    __syncthreads();

    auto step = [&](auto PAR, auto SLOT, int s) {
        constexpr int par = decltype(PAR)::value;
        constexpr int slot = decltype(SLOT)::value;
        const int buf = s & 1;
        load_begin();  // rows of step s+PF: six loads at sections 0 .. 5; the previous step's four store pairs at sections 6 .. 9
        const char* xr = in_lds + buf * G::IIN + row_rd;
        const char* mr = mid_lds + (buf ^ 1) * G::IMID + mid_rd;
        double x1l[4], x0l[4], x1h[4], x0h[4];  // [window position]: IL / IH branch, parity 1 / 0
        double a[4], b[4], e[4], o[4];
        constexpr int NJ = kSEC / 4;  // window slots (4 taps each) per row-synthesis section
        v2d P[HLEN / 2 + 3];
        auto ldP = [&](auto MM) {
            constexpr int m = decltype(MM)::value;
            P[m] = *reinterpret_cast<const v2d*>(xr + 16 * m);
        };
        static_for<NJ + 3>([&](auto MM) { ldP(MM); });
#pragma unroll
        for (int q = 0; q < 4; q++) {
            a[q] = *reinterpret_cast<const double*>(mr + q * (2 * kNCW * 8));
            b[q] = *reinterpret_cast<const double*>(mr + 4 * 2 * kNCW * 8 + q * (2 * kNCW * 8));
        }
        static_for<NS>([&](auto SS) {
            constexpr int sec = decltype(SS)::value;
            constexpr int cur = (par + sec) & 1, nxt = cur ^ 1;
            constexpr int nsec = (sec + 1) % NS;
            ctab_t tp = tbase;
            if constexpr (sec == 0)
                asm volatile("" : "+s"(tp), "+v"(P[0]), "+v"(P[NJ + 2]), "+s"(tb[cur][0]), "+s"(tb[cur][kSEC - 1]));
            else if constexpr (sec < NSEC)
                asm volatile(""
                             : "+s"(tp), "+v"(x1l[0]), "+v"(x0l[0]), "+v"(x1h[0]), "+v"(x0h[0]), "+v"(x1l[3]), "+v"(x0l[3]), "+v"(x1h[3]), "+v"(x0h[3]),
                               "+v"(P[NJ * sec + 3]), "+v"(P[NJ * sec + NJ + 2]), "+s"(tb[cur][0]), "+s"(tb[cur][kSEC - 1]));
            else if constexpr (sec == NSEC)
                asm volatile("" : "+s"(tp), "+v"(x1l[0]), "+v"(x0l[0]), "+v"(x1h[0]), "+v"(x0h[0]), "+v"(x1l[3]), "+v"(x0l[3]), "+v"(x1h[3]), "+v"(x0h[3]),
                             "+v"(a[0]), "+v"(b[0]), "+v"(a[3]), "+v"(b[3]), "+s"(tb[cur][0]), "+s"(tb[cur][kSEC - 1]));
            else
                asm volatile("" : "+s"(tp), "+v"(a[0]), "+v"(b[0]), "+v"(a[1]), "+v"(b[1]), "+v"(a[2]), "+v"(b[2]), "+v"(a[3]), "+v"(b[3]), "+s"(tb[cur][0]),
                             "+s"(tb[cur][kSEC - 1]));
#pragma unroll
            for (int j = 0; j < kSEC; j++) tb[nxt][j] = (PDWT_LAT_DIAG & 32) ? tb[cur][j] : tp[kSEC * nsec + j];  // (bit 5: no scalar loads -- wrong results, timing only)
            if constexpr (sec + 1 < NSEC) static_for<NJ>([&](auto MM) {
                constexpr int pm = NJ * (sec + 1) + 3 + decltype(MM)::value;
                if constexpr (PDWT_LAT_DIAG & 16) P[pm] = P[pm - NJ - 3];
                else ldP(std::integral_constant<int, pm>{});
            });
            if constexpr (sec < 6) load_one(SLOT, std::integral_constant<int, sec>{});
            if constexpr (sec >= 6 && sec < 10) {
                // outputs kept from the previous step (pe / po): chunk-local pair-step qq -> output pair m0 + qq - C -> rows 2m + sig, 2m + sig + 1
                constexpr int q = sec - 6;
                // rows 2m + sig and 2m + sig + 1 of pair m = m0 + so; the second one wraps to row 0 for the image's last pair when sig = 1
                double* r1p = reinterpret_cast<double*>(reinterpret_cast<char*>(sp0) + spitch);
                if (sig && m0 + so == Nri - 1) r1p = out;  // (uniform)
                st2_sv_m<(PDWT_LAT_NT & 8) != 0>(sp0, r1p, ocol, pe[q], po[q], mask_in_range(so, nm));
                so++;
                if constexpr (!(PDWT_LAT_DIAG & 2)) sp0 = reinterpret_cast<double*>(reinterpret_cast<char*>(sp0) + 2 * spitch);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (sec < NSEC) {
                // row synthesis: window slot j = NJ sec + jj of position pos is sample P[pos + j] = (a-band, b-band) at coefficient column c - CI + j
                static_for<NJ>([&](auto JJ) {
                    constexpr int jj = decltype(JJ)::value;
                    static_for<4>([&](auto PP) {
                        constexpr int pos = decltype(PP)::value;
                        const v2d p = P[pos + NJ * sec + jj];
                        if constexpr (sec == 0 && jj == 0) {
                            mul2(x1l[pos], x0l[pos], p.x, tb[cur][0], tb[cur][1]);
                            mul2(x1h[pos], x0h[pos], p.y, tb[cur][2], tb[cur][3]);
                        } else {
                            fma2(x1l[pos], x0l[pos], p.x, tb[cur][4 * jj], tb[cur][4 * jj + 1]);
                            fma2(x1h[pos], x0h[pos], p.y, tb[cur][4 * jj + 2], tb[cur][4 * jj + 3]);
                        }
                    });
                });
            } else {
                // synthesis lattice on the previous step's rows: stages K-1 .. 1, then the end matrix; four pair-steps side by side
                // table order: t_{K-1}, t_{K-2}, ..., t_1 at positions 0 .. K-2, the end matrix MI[4] at G::IMP (inside one section)
                constexpr int ls = sec - NSEC;
                constexpr int n_here = (K - 1 - kSEC * ls < 0) ? 0 : ((K - 1 - kSEC * ls < kSEC) ? K - 1 - kSEC * ls : kSEC);
                if constexpr (n_here > 0) {
                    double W[n_here];
                    static_for<4>([&](auto QQ) {
                        constexpr int q = decltype(QQ)::value;
                        static_for<n_here>([&](auto II) {
                            constexpr int k = decltype(II)::value;
                            constexpr int i = K - 1 - (kSEC * ls + k);  // stage
                            const double t = tb[cur][k];
                            const double na = __builtin_fma(t, b[q], a[q]);
                            b[q] = fnma_s(t, a[q], b[q]);
                            // the first branch is delayed by one pair-step
                            a[q] = q == 0 ? A[i] : W[k];
                            if constexpr (q == 3) A[i] = na;
                            else W[k] = na;
                        });
                    });
                }
                if constexpr (ls == G::IMP / kSEC) {
                    constexpr int mp = G::IMP - kSEC * ls;  // position of MI[0] in this section
                    static_assert(mp >= 0 && mp + 3 < kSEC, "the end matrix must sit inside one lattice section");
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        e[q] = __builtin_fma(tb[cur][mp + 1], b[q], tb[cur][mp] * a[q]);
                        o[q] = __builtin_fma(tb[cur][mp + 3], b[q], tb[cur][mp + 2] * a[q]);
                    }
                }
            }
        });
        // this step's (u1 | u2) rows: window position pos -> strip-local output columns 2 (4g + pos) + {0, 1} (the strip's column 0 is 2 c0 - SHIFT)
        {
            char* mw = mid_lds + buf * G::IMID + mid_wr + 8 * (8 * g);
#pragma unroll
            for (int pos = 0; pos < 4; pos++)
                *reinterpret_cast<v2d*>(mw + 16 * pos) = v2d{x1l[pos] + x1h[pos], x0l[pos] + x0h[pos]};
        }
        {
            // (younger VMEM instructions: the 8 stores of the step that issued the loads, PF-1 whole steps of 14)
            constexpr int ns = (slot + 1) % PF;
            wait_sel6<8 + 14 * (PF - 1)>(s >= PF - 1, sa[ns][0], sb[ns][0], sa[ns][1], sb[ns][1], sa[ns][2], sb[ns][2]);
            store_rows(std::integral_constant<int, ns>{}, buf ^ 1);
        }
#pragma unroll
        for (int q = 0; q < 4; q++) pe[q] = e[q], po[q] = o[q];  // stored during the next step
        if constexpr (PDWT_LAT_DIAG & 4) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (diagnostic: no barrier -- wrong results, timing only)
        else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };

    for (int sb = 0; sb <= nsteps + 1; sb += UB) {
        bool fin = false;
        static_for<UB>([&](auto UU) {
            constexpr int uu = decltype(UU)::value;
            if (!fin) {
                step(std::integral_constant<int, (uu * NS) & 1>{}, std::integral_constant<int, uu % PF>{}, sb + uu);
                fin = sb + uu + 1 > nsteps + 1;
            }
        });
    }
    clock_probe_stamp(probe, 1, probe_all);
}

// =================================================================================================
// host side
// =================================================================================================
namespace {
struct LatBank {
    const char* name;
    int K;
    double m0[4], mi[4], t[PDWT_MAX_FILTER_WIDTH / 2];
};
#define PDWT_LATTICE(name, K, ...) {name, K, __VA_ARGS__},
const LatBank g_lat[] = {
#include "lattice_table.inc"
};
#undef PDWT_LATTICE
constexpr int kNumLat = sizeof(g_lat) / sizeof(g_lat[0]);

// the lattice of a bank: found by EXACT comparison of the taps the caller passes with the named bank's (a custom bank, or a named one
// somebody scaled, has no entry and keeps the direct-form kernels)
const LatBank* find_lattice(const Taps2<double>& f, int hlen, bool inverse)
{
    for (int b = 0; b < kNumLat; b++) {
        if (2 * g_lat[b].K != hlen) continue;
        pdwt_filters_f64 fb;
        if (pdwt_compute_filters_separable_f64(g_lat[b].name, 0, &fb) != hlen) continue;
        const double* A = inverse ? fb.IL : fb.L;
        const double* B = inverse ? fb.IH : fb.H;
        if (!memcmp(A, f.a, sizeof(double) * hlen) && !memcmp(B, f.b, sizeof(double) * hlen)) return &g_lat[b];
    }
    return nullptr;
}
}  // namespace

#define PDWT_CHECK_LAUNCH() PDWT_HIP_TRY(hipGetLastError())
#define PDWT_LAT_HLENS(X) X(40)

// uneven split of a strip's rows between the two workgroups of a CU (lat_tile): only when the grid really puts two on every CU
static int lat_skew(int strips, int chunks)
{
    static int cus[64] = {0};
    int dev = 0;
    if (kWGPerCU != 2 || hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    if (!cus[dev]) {
        hipDeviceProp_t p;
        cus[dev] = hipGetDeviceProperties(&p, dev) == hipSuccess ? p.multiProcessorCount : -1;
    }
    const int T = strips * chunks, sk = knob(KN_F64_LDS_SKEW);
    if (sk <= 50 || sk >= 100 || cus[dev] <= 0 || chunks < 2 || chunks > 64 || (chunks & 1)) return 0;
    if (T <= cus[dev] || T > 2 * cus[dev]) return 0;
    return sk;
}

static bool lat_geometry_ok(int nr, int nc)
{
    const int mn = knob(KN_F64_LAT_MIN);
    return !(nr & 7) && !(nc % (2 * kNCW)) && nr >= mn && nc >= mn && nr >= 256;
}

template <int HLEN>
static int launch_fwd_lat(const double* in, double* cA, double* cH, double* cV, double* cD, int nr, int nc, const Taps2<double>& f, const LatBank* lb)
{
    using G = LatGeo<HLEN>;
    constexpr int K = G::K;
    const int nr2 = nr / 2, nc2 = nc / 2;
    const int strips = nc2 / kNCW;
    const int target = knob(KN_F64_LDS_WGS) * kWGPerCU / 2;
    int chunks = std::max(1, target / strips);
    int RO = idiv_up(idiv_up(nr2, chunks), 4) * 4;
    RO = std::max(RO, 32);
    chunks = idiv_up(nr2, RO);
    if (lds_opt_in<k_fwd2d_lat<HLEN>>() != PDWT_OK) return PDWT_EHIP;  // (> 64 KB of dynamic LDS: opt-in once per device)
    LatTable<G::NS> tt;
    memset(&tt, 0, sizeof(tt));
    for (int j = 0; j < HLEN; j++) {  // window position j meets { L[hlen-1-j], H[hlen-1-j] } (SURVEY A-1)
        tt.t[2 * j] = f.a[HLEN - 1 - j];
        tt.t[2 * j + 1] = f.b[HLEN - 1 - j];
    }
    double* lt = tt.t + kSEC * G::NSEC;
    for (int j = 0; j < 4; j++) lt[j] = lb->m0[j];
    for (int i = 1; i < K; i++) lt[3 + i] = lb->t[i - 1];
    int pall = 0;
    unsigned long long* const pbuf = clock_probe_all(&pall);  // (diagnostic: every workgroup stamps its start and end)
    if (strips * chunks > kClockProbeAllBlocks) pall = 0;
    stat_lat(0);
    KTimer kt(K_FWD2D_F64);
    hipLaunchKernelGGL((k_fwd2d_lat<HLEN>), dim3(strips * chunks), dim3(kNT), G::FLDS, stream(), tt, in, cA, cH, cV, cD, nr, nc, RO, strips, lat_skew(strips, chunks),
                       pall == 1 ? pbuf : clock_probe_slot(clock_probe_size_class(nr)), pall == 1 ? 1 : 0);
    PDWT_CHECK_LAUNCH();
    return PDWT_OK;
}

template <int HLEN>
static int launch_inv_lat(const double* cA, const double* cH, const double* cV, const double* cD, double* out, int nri, int nci, const Taps2<double>& f,
                          const LatBank* lb)
{
    using G = LatGeo<HLEN>;
    constexpr int K = G::K;
    const int strips = nci / kNCW;
    const int target = knob(KN_F64_LDS_WGS) * kWGPerCU / 2;
    int chunks = std::max(1, target / strips);
    int NP = idiv_up(idiv_up(nri, chunks), 4) * 4;
    NP = std::max(NP, 32);
    chunks = idiv_up(nri, NP);
    if (lds_opt_in<k_inv2d_lat<HLEN>>() != PDWT_OK) return PDWT_EHIP;
    LatTable<G::NS> tt;
    memset(&tt, 0, sizeof(tt));
    for (int j = 0; j < HLEN / 2; j++) {  // window slot j meets { IL[h-2-2j], IL[h-1-2j], IH[h-2-2j], IH[h-1-2j] } (parity 1 / parity 0)
        tt.t[4 * j + 0] = f.a[HLEN - 2 - 2 * j];
        tt.t[4 * j + 1] = f.a[HLEN - 1 - 2 * j];
        tt.t[4 * j + 2] = f.b[HLEN - 2 - 2 * j];
        tt.t[4 * j + 3] = f.b[HLEN - 1 - 2 * j];
    }
    double* lt = tt.t + kSEC * G::NSEC;
    for (int r = 0; r < K - 1; r++) lt[r] = lb->t[K - 2 - r];  // t_{K-1}, t_{K-2}, ..., t_1
    for (int j = 0; j < 4; j++) lt[G::IMP + j] = lb->mi[j];
    int pall = 0;
    unsigned long long* const pbuf = clock_probe_all(&pall);
    if (strips * chunks > kClockProbeAllBlocks) pall = 0;
    stat_lat(1);
    KTimer kt(K_INV2D_F64);
    hipLaunchKernelGGL((k_inv2d_lat<HLEN>), dim3(strips * chunks), dim3(kNT), G::ILDS, stream(), tt, cA, cH, cV, cD, out, nri, nci, NP, strips, lat_skew(strips, chunks),
                       pall == 2 ? pbuf : clock_probe_slot(8 + clock_probe_size_class(2 * nri)), pall == 2 ? 1 : 0);
    PDWT_CHECK_LAUNCH();
    return PDWT_OK;
}

int fwd2d_f64_lat(const double* in, double* cA, double* cH, double* cV, double* cD, int nr, int nc, int hlen, const Taps2<double>& f)
{
    if (knob(KN_F64_LAT) < 1 || !lat_geometry_ok(nr, nc) || !counted_waits_ok()) return 1;
    switch (hlen) {
#define X(H) \
    case H: { \
        const LatBank* lb = find_lattice(f, hlen, false); \
        return lb ? launch_fwd_lat<H>(in, cA, cH, cV, cD, nr, nc, f, lb) : 1; \
    }
        PDWT_LAT_HLENS(X)
#undef X
        default: return 1;
    }
}

int inv2d_f64_lat(const double* cA, const double* cH, const double* cV, const double* cD, double* out, int nri, int nci, int nro, int nco, int hlen,
                  const Taps2<double>& f)
{
    if (knob(KN_F64_LAT) < 1 || nro != 2 * nri || nco != 2 * nci || !lat_geometry_ok(nro, nco) || !counted_waits_ok()) return 1;
    switch (hlen) {
#define X(H) \
    case H: { \
        const LatBank* lb = find_lattice(f, hlen, true); \
        return lb ? launch_inv_lat<H>(cA, cH, cV, cD, out, nri, nci, f, lb) : 1; \
    }
        PDWT_LAT_HLENS(X)
#undef X
        default: return 1;
    }
}

}  // namespace pdwt
