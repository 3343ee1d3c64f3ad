// dwt_lds.hip -- one 2-D DWT level per launch with both passes fed from LDS, templated over the element type and the bank length.
// Runs: every double-precision bank of even length; the float32 banks of more than 16 taps; the float32 banks of up to 16 taps on
// the geometries the cascade / streaming kernels do not take (odd sizes, sizes that are not a multiple of 4).  Written for db20 in
// double (40 taps: configuration C5) -- the figures in the comments are that case.  Lengths that are not a multiple of 8 run the next
// one with a symmetrically zero-padded bank; odd sizes follow the reference's repeat-last-sample-then-periodic rule.
//
// Reference code replaced: w_kern_forward_pass1 + w_kern_forward_pass2 (src/separable.cu:91-176) of one iteration of
// w_forward_separable (:179-209).
//
// Why the ring is not in REGISTERS (the first fused form, removed in round 3, kept it there): a 40-row ring of (lo, hi) doubles is 160 VGPRs,
// which pinned that kernel at 2 waves per SIMD with no register left to prefetch its LDS window reads -- the VALU is busy 55-62 %
// of the time.  Here the ring lives in LDS and every thread is register-blocked over TWO outputs of the pass it runs:
//   * a workgroup (256 threads, two per CU) owns 64 output columns and walks down a chunk, 8 input rows (4 output rows) per step;
//   * ROW pass: thread = (input row, pair of adjacent output columns); its two 40-sample windows overlap in 38 samples, so 21
//     aligned two-sample LDS reads (16 bytes in double) feed 160 FMAs; results (lo, hi) go to a ring of 56 rows in LDS (two planes,
//     column-major so that one read delivers two ring rows);
//   * COLUMN pass: thread = (plane, column, output rows q and q+2 of the step's four): 22 two-row reads feed 160 FMAs, emits
//     (A,H) or (V,D).  The two row pairs of a step differ by two ring rows; that offset sits in the base register, so the
//     wrap-around of the ring is the same immediate for both and rows 0, 1 are mirrored behind the last ring row;
//   * 0.13 LDS reads per FMA as before, but ~100 VGPRs of read-ahead instead of none; the column pass of a group of rows runs in the
//     same step as the row pass of the NEXT rows (one barrier per step), and both passes of a section share the section's taps,
//     which arrive by scalar loads straight from the kernarg segment (the tap table is the first kernel argument);
//   * Bank conflicts: the lanes an LDS cycle serves start their windows two output columns apart -- a 2-way conflict.  A wave
//     therefore works on TWO input rows whose LDS images are an ODD number of two-sample slots apart (row = 83 slots at db20), and
//     the lanes are dealt so that every lane group holds as many windows of one row as of the other: all banks, once.
//   * The ring position of a step repeats every 7 steps; the body is unrolled over those 7 phases so that every LDS address is
//     base register + immediate.
// Per-sample arithmetic: taps in ascending window position, one FMA per tap, rows before columns -- the reference's and the
// oracle's order: bit-identical to the two-pass kernels (float32: two such FMAs per v_pk_fma_f32).
#include "dwt_lds.hpp"

#include <algorithm>
#include <atomic>
#include <type_traits>

#include "stream_dev.hpp"

namespace pdwt {

namespace {
constexpr int kNT = 256;      // threads per workgroup
constexpr int kNCW = 64;      // output columns per workgroup
constexpr int kNIR = 8;       // input rows per step (= 4 output rows)
constexpr int kRing = 56;     // ring rows in LDS: window of 46 rows (two output rows + their neighbours) + the 8 rows being written
constexpr int kPhases = 7;    // kRing / kNIR
constexpr int kLag = 6;       // the column pass of step s emits output rows 4(s-kLag) .. +3
template <typename T> using ctaps_t = const T __attribute__((address_space(4))) *;
template <typename T> using pair_t = T __attribute__((ext_vector_type(2)));  // a register pair the inline asm can take as ONE operand
// The taps, in the order the kernels consume them, travel as the FIRST kernel argument: the kernarg segment is constant memory, so
// the sections' scalar loads read it directly (no staging launch, no device scratch).
template <typename T>
struct TapTable {
    T t[2 * PDWT_MAX_FILTER_WIDTH];
};
template <typename T> __device__ __forceinline__ ctaps_t<T> kernarg_taps() { return (ctaps_t<T>)__builtin_amdgcn_kernarg_segment_ptr(); }
// uniform base (SGPR pair) + per-lane 32-bit BYTE offset: the addressing mode that needs no 64-bit vector arithmetic per access
template <typename T>
__device__ __forceinline__ T ld_sv(const T* base, unsigned boff)
{
    typedef const char __attribute__((address_space(1))) * gbytes_t;
    typedef const T __attribute__((address_space(1))) * gelem_t;
    // opaque to the optimiser: keeps hipcc from folding the per-lane offset into a loop-invariant 64-bit VGPR pointer
    asm("" : "+s"(base));
    return *(gelem_t)((gbytes_t)base + boff);
}
// Workgroup id -> (strip, chunk), XCD-aware.  The dispatcher deals consecutive workgroup ids round-robin over the 8 XCDs; every
// XCD is given a CONTIGUOUS run of the strip-major tile order instead, so that horizontally adjacent strips -- whose input windows
// overlap by hlen-2 columns -- run on the same XCD at the same time and the overlap is served by that XCD's L2.
__device__ __forceinline__ void xcd_tile(int strips, int& strip, int& chunk)
{
    const int T = gridDim.x, w = blockIdx.x;
    const int x = w & 7, per = T >> 3, rem = T & 7;
    const int L = x * per + min(x, rem) + (w >> 3);
    strip = L % strips;
    chunk = L / strips;
}
// (strip, first row y0, rows nout) of this workgroup; `chunks` workgroups share a strip.  skew = 0: equal chunks of RO rows.
// skew > 0: two workgroups share a CU and the issue arbiter serves the OLDER waves first, so the workgroup that was dispatched
// first runs at full speed, the other at ~0.55 of it, and then finishes alone at half the FP64 issue rate (per-workgroup
// timestamps, tools/lds_trace.py: 242 vs 347 us forward, 276 vs 400 us inverse at 8192^2 -- and exactly the other way round when the
// later half is given a higher s_setprio; alternating priorities only moves the first group's end, the launch still ends with
// the last one).  The rows of a strip are therefore dealt by WEIGHT: skew to a workgroup of the first part of the dispatch order,
// 100 - skew to one of the second part (boundaries rounded to multiples of 4 rows).
__device__ __forceinline__ void lds_tile(int strips, int chunks, int RO, int nrows, int skew, int& strip, int& y0, int& nout)
{
    if (skew > 0) {
        // first part of the dispatch order -> the ODD chunks of every strip (never more than half the grid: those workgroups are
        // the first on their CU), the rest -> the even chunks (each part dealt to the XCDs like xcd_tile does); a strip's rows go
        // to its chunks by weight, skew : 100 - skew
        const int T = gridDim.x, E = strips * (chunks >> 1);
        const int late = (int)blockIdx.x >= E ? 1 : 0;
        const int wh = blockIdx.x - late * E, n = late ? T - E : E;
        const int x = wh & 7, per = n >> 3, rem = n & 7;
        const int P = x * per + min(x, rem) + (wh >> 3);
        strip = P % strips;
        const int chunk = 2 * (P / strips) + 1 - late;
        const int wa = skew, wb = 100 - skew;
        const int before = wb * ((chunk + 1) >> 1) + wa * (chunk >> 1), total = wb * ((chunks + 1) >> 1) + wa * (chunks >> 1);
        const int q = (nrows + 3) >> 2;  // groups of 4 rows
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
__device__ __forceinline__ void st_sv(double* base, unsigned boff, double v)
{
    asm volatile("global_store_dwordx2 %0, %1, %2" ::"v"(boff), "v"(v), "s"(base) : "memory");
}
__device__ __forceinline__ void st_sv(float* base, unsigned boff, float v)
{
    asm volatile("global_store_dword %0, %1, %2" ::"v"(boff), "v"(v), "s"(base) : "memory");
}
// (a0, a1) (+)= x * (t0, t1): two accumulators fed by one sample and an adjacent pair of taps.  float: ONE v_pk_fma_f32 (the sample is
// broadcast by op_sel, the taps are an aligned SGPR pair); each half is the same single FMA as in the scalar form.
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
__device__ __forceinline__ void fma2(float& a0, float& a1, float x, float t0, float t1)
{
    const v2f r = pk_fma(v2f{x, x}, v2f{t0, t1}, v2f{a0, a1});
    a0 = r.x;
    a1 = r.y;
}
__device__ __forceinline__ void mul2(float& a0, float& a1, float x, float t0, float t1)
{
    const v2f r = v2f{x, x} * v2f{t0, t1};
    a0 = r.x;
    a1 = r.y;
}
__device__ __forceinline__ void st_flat(double* p, double v) { asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void st_flat(float* p, float v) { asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"(v) : "memory"); }
}  // namespace

template <typename T, int HLEN>
struct F64Lds {
    static constexpr int ES = sizeof(T);
    static constexpr int C = HLEN / 2 - 1;
    static constexpr int LWI = 2 * kNCW + HLEN - 2;  // input columns staged per row (elements)
    static constexpr int PAIRS = LWI / 2;            // two-element slots per staged row; must be odd (bank spreading, see above)
    static constexpr int kColStride = (kRing + 2) * ES;          // ring planes are column-major: [column][ring row], + mirror of rows 0, 1
    static constexpr int kPlaneBytes = kColStride * kNCW;
    static constexpr int kInBufBytes = kNIR * LWI * ES;
    static constexpr int kLdsBytes = 2 * kPlaneBytes + 2 * kInBufBytes;
    static_assert(PAIRS % 2 == 1, "staged rows must be an odd number of 16-byte slots apart");
    static_assert(HLEN % 8 == 0, "sections of 8 taps");
    // the window of the last output row of a group ends kLag steps back
    static_assert(2 * 3 + 2 + HLEN - 1 <= kNIR * (kLag - 1) + kNIR - 1 && kRing <= 58 && 2 * kPlaneBytes + 2 * kInBufBytes <= 81920, "column windows must be complete when their step starts");
    static_assert(kNIR * (kLag + 1) - kRing <= 2, "the rows a step writes must not be part of the windows it reads");
};

template <typename T, int HLEN>
__global__ __launch_bounds__(kNT, 2) void k_fwd2d_f64lds(TapTable<T> /*read through kernarg_taps()*/, const T* __restrict__ in,
                                                          T* __restrict__ cA, T* __restrict__ cH, T* __restrict__ cV,
                                                          T* __restrict__ cD, int Nr, int Nc, int RO, int strips, unsigned long long* probe, int probe_all, int skew,
                                                          const void* tbl)
{
    clock_probe_stamp(probe, 0, probe_all);
    if (tbl) {  // batched launch (pdwt_batch2d_*_f64): this workgroup's image -- five pointers per image, read through the constant address space
        typedef const unsigned long long __attribute__((address_space(4))) * tbl_t;
        const tbl_t q = (tbl_t)(const unsigned long long*)tbl + 5 * (size_t)blockIdx.y;
        in = (const T*)q[0];
        cA = (T*)q[1];
        cH = (T*)q[2];
        cV = (T*)q[3];
        cD = (T*)q[4];
    }
    using G = F64Lds<T, HLEN>;
    using V2 = pair_t<T>;
    constexpr int ES = sizeof(T);
    constexpr int C = G::C, LWI = G::LWI, PAIRS = G::PAIRS;
    constexpr int NSEC = HLEN / 8;
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    // odd sizes: the reference's rule (src/separable.cu:116-121) -- repeat the last sample once, then periodic -- is applied while
    // the rows are staged: Nv x (Nc + (Nc & 1)) is the virtual image, ceil-half the band size
    const int Nc2 = (Nc + 1) >> 1, Nr2 = (Nr + 1) >> 1;
    const int Nv = Nr + (Nr & 1);
    int strip, y0, nout;
    lds_tile(strips, gridDim.x / strips, RO, Nr2, skew, strip, y0, nout);
    const int i0 = strip * kNCW;
    if (nout <= 0) return;
    const int ngroups = (nout + 3) >> 2;
    const int nsteps = ngroups + kLag;

    // ---- staging role: thread -> (row of the step, slots k0, k0+32, k0+64 of that row)
    const int srow = tid >> 5, k0 = tid & 31;
    const int cbase = 2 * i0 - C;  // LDS column j <-> input column cbase + j
    unsigned gc[3][2];  // byte offsets within an image row
#pragma unroll
    for (int m = 0; m < 3; m++) {
        gc[m][0] = (unsigned)ES * (unsigned)wrap_ext(cbase + 2 * (k0 + 32 * m), Nc);
        gc[m][1] = (unsigned)ES * (unsigned)wrap_ext(cbase + 2 * (k0 + 32 * m) + 1, Nc);
    }
    const unsigned rowb = (unsigned)ES * (unsigned)Nc * (unsigned)srow;  // the thread's row within the 8 rows of a step (when they do not wrap)
    const bool third = k0 + 64 < PAIRS;
    // chunk-local row rho <-> image row 2*y0 - C - 2 + rho (two leading rows align the output groups with the steps)
    int rnext = wrapi(2 * y0 - C - 2, Nv);  // (virtual) image row of rho = 8*(step to stage)
    T st[3][2];
    auto load_rows = [&]() {
        if (rnext + kNIR <= Nr) {  // the step's 8 rows are consecutive: uniform base + 32-bit per-lane offset
            const T* p = in + (size_t)rnext * Nc;
            st[0][0] = ld_sv(p, rowb + gc[0][0]);
            st[0][1] = ld_sv(p, rowb + gc[0][1]);
            st[1][0] = ld_sv(p, rowb + gc[1][0]);
            st[1][1] = ld_sv(p, rowb + gc[1][1]);
            if (third) {
                st[2][0] = ld_sv(p, rowb + gc[2][0]);
                st[2][1] = ld_sv(p, rowb + gc[2][1]);
            }
        } else {  // (uniform; at most two steps of a chunk) they wrap around the bottom edge: per-lane 64-bit row pointers
            const int r = rnext + srow;
            const int rv = r >= Nv ? r - Nv : r;  // virtual row; row Nr (odd heights) is row Nr-1 again
            const char* p = reinterpret_cast<const char*>(in + (size_t)min(rv, Nr - 1) * Nc);
            st[0][0] = *reinterpret_cast<const T*>(p + gc[0][0]);
            st[0][1] = *reinterpret_cast<const T*>(p + gc[0][1]);
            st[1][0] = *reinterpret_cast<const T*>(p + gc[1][0]);
            st[1][1] = *reinterpret_cast<const T*>(p + gc[1][1]);
            if (third) {
                st[2][0] = *reinterpret_cast<const T*>(p + gc[2][0]);
                st[2][1] = *reinterpret_cast<const T*>(p + gc[2][1]);
            }
        }
        rnext += kNIR;
        rnext = rnext >= Nv ? rnext - Nv : rnext;
    };
    char* const in_lds = lds_raw + 2 * G::kPlaneBytes;
    int stage_off = srow * LWI * ES + k0 * 2 * ES;  // within an input buffer
    auto store_rows = [&](int buf) {
        char* b = in_lds + buf * G::kInBufBytes + stage_off;
        *reinterpret_cast<V2*>(b) = V2{st[0][0], st[0][1]};
        *reinterpret_cast<V2*>(b + 32 * 2 * ES) = V2{st[1][0], st[1][1]};
        if (third) *reinterpret_cast<V2*>(b + 64 * 2 * ES) = V2{st[2][0], st[2][1]};
    };

    // ---- row-pass role: lane -> (row bit, column pair); every 16-lane LDS group holds 8 windows of each of the wave's two rows
    const int rbit = (lane >> 3) & 1;
    const int cp = (((lane >> 5) & 1) << 4) | (((lane >> 2) & 1) << 3) | (((lane >> 4) & 1) << 2) | (lane & 3);
    const int rrow = 2 * w + rbit;
    const int row_rd = rrow * LWI * ES + cp * 4 * ES;             // window of output column 2cp starts at LDS column 4cp
    const int row_wr = 2 * cp * G::kColStride + rrow * ES;     // ring row (8*phase + rrow), columns 2cp, 2cp+1 (lo plane; hi plane + kPlaneBytes)
    // ---- column-pass role: wave -> (plane, row pair), lane -> column
    const int plane = w & 1, rp = w >> 1;
    const char* const col_rd = lds_raw + plane * G::kPlaneBytes + lane * G::kColStride + rp * 2 * ES;
    T* const outL = plane ? cV : cA;
    T* const outH = plane ? cD : cH;
    const bool col_ok = i0 + lane < Nc2;
    const unsigned ocol = (unsigned)ES * (unsigned)(i0 + lane);

    ctaps_t<T> tbase = kernarg_taps<T>();
    T tl[2][8], th[2][8];
#pragma unroll
    for (int jj = 0; jj < 8; jj++) {
        tl[0][jj] = tbase[2 * jj];
        th[0][jj] = tbase[2 * jj + 1];
    }

    load_rows();
    store_rows(0);
    __syncthreads();

    auto step = [&](auto PHC, int s) {
        constexpr int PH = decltype(PHC)::value;
        constexpr int cbase_row = kNIR * ((PH + kPhases - (kLag % kPhases)) % kPhases) + 2;  // ring row of k = 0 (rp = 0)
        const int buf = s & 1;
        load_rows();  // rows of step s+1, in flight while this step computes
        const char* xr = in_lds + buf * G::kInBufBytes + row_rd;
        T lo0, hi0, lo1, hi1;  // row pass: output columns 2cp, 2cp+1 of row rrow
        T a0, h0, a1, h1;      // column pass: output rows 4g+rp, 4g+rp+2 of this plane
        // P[m] = samples 2m, 2m+1 of the row window (42 samples); Q[u] = ring rows k = 2u, 2u+1 of the column window (44 rows).
        // Section sec consumes P[4sec .. 4sec+4] and Q[4sec .. 4sec+5]; they are loaded one section AHEAD.
        V2 P[HLEN / 2 + 1], Q[HLEN / 2 + 2];
        auto ldP = [&](auto MM) {
            constexpr int m = decltype(MM)::value;
            P[m] = *reinterpret_cast<const V2*>(xr + m * 2 * ES);
        };
        auto ldQ = [&](auto UU) {
            constexpr int u = decltype(UU)::value;
            Q[u] = *reinterpret_cast<const V2*>(col_rd + ((cbase_row + 2 * u) % kRing) * ES);
        };
        static_for<5>([&](auto MM) { ldP(MM); });
        static_for<6>([&](auto UU) { ldQ(UU); });
        static_for<NSEC>([&](auto SS) {
            constexpr int sec = decltype(SS)::value;
            constexpr int gsec = PH * NSEC + sec;
            constexpr int cur = gsec & 1, nxt = cur ^ 1;
            constexpr int nsec = (sec + 1) % NSEC;
            // Ordering point.  Everything this section consumes (its LDS data and its taps) passes through the statement, so the
            // one wait it needs -- lgkmcnt(0): scalar loads return out of order -- sits HERE, before the next section's loads are
            // issued, and nothing further has to be waited for until the next ordering point.
            ctaps_t<T> tp = tbase;
            if constexpr (sec == 0)
                asm volatile(""
                             : "+s"(tp), "+v"(P[0]), "+v"(Q[0]), "+v"(Q[1]), "+v"(P[1]), "+v"(P[2]), "+v"(P[3]), "+v"(P[4]), "+v"(Q[2]), "+v"(Q[3]),
                               "+v"(Q[4]), "+v"(Q[5]), "+s"(tl[cur][0]), "+s"(tl[cur][7]));
            else
                asm volatile(""
                             : "+s"(tp), "+v"(lo0), "+v"(hi0), "+v"(lo1), "+v"(hi1), "+v"(a0), "+v"(h0), "+v"(a1), "+v"(h1),
                               "+v"(P[4 * sec + 1]), "+v"(P[4 * sec + 2]), "+v"(P[4 * sec + 3]), "+v"(P[4 * sec + 4]), "+v"(Q[4 * sec + 2]),
                               "+v"(Q[4 * sec + 3]), "+v"(Q[4 * sec + 4]), "+v"(Q[4 * sec + 5]), "+s"(tl[cur][0]), "+s"(tl[cur][7]));
#pragma unroll
            for (int jj = 0; jj < 8; jj++) {
                tl[nxt][jj] = tp[2 * (nsec * 8 + jj)];
                th[nxt][jj] = tp[2 * (nsec * 8 + jj) + 1];
            }
            if constexpr (sec + 1 < NSEC) {
                static_for<4>([&](auto MM) { ldP(std::integral_constant<int, 4 * (sec + 1) + 1 + decltype(MM)::value>{}); });
                static_for<4>([&](auto UU) { ldQ(std::integral_constant<int, 4 * (sec + 1) + 2 + decltype(UU)::value>{}); });
            }
            __builtin_amdgcn_sched_barrier(0);  // the loads above are issued BEFORE this section's arithmetic, not wherever the scheduler likes
            // row pass: tap j meets sample j (first output) and sample j+2 (second output)
            static_for<4>([&](auto MM) {
                constexpr int m = decltype(MM)::value;
                constexpr int jj = 2 * m;
                const V2 p = P[4 * sec + m], q = P[4 * sec + m + 1];
                if constexpr (sec == 0 && m == 0) {  // fma(x, t, 0) == x * t: no zero-initialisation moves
                    mul2(lo0, hi0, p.x, tl[cur][jj], th[cur][jj]);
                    mul2(lo1, hi1, q.x, tl[cur][jj], th[cur][jj]);
                } else {
                    fma2(lo0, hi0, p.x, tl[cur][jj], th[cur][jj]);
                    fma2(lo1, hi1, q.x, tl[cur][jj], th[cur][jj]);
                }
                fma2(lo0, hi0, p.y, tl[cur][jj + 1], th[cur][jj + 1]);
                fma2(lo1, hi1, q.y, tl[cur][jj + 1], th[cur][jj + 1]);
            });
            // column pass: output rows 4g+rp and 4g+rp+2 (g = s-kLag); ring rows rho = 8g + 2 + 2rp + k, tap j meets k = j (first
            // row) and k = j+4 (second row).  (During the first kLag steps of a chunk this works on rows that do not exist yet; nothing
            // is stored.)
            static_for<4>([&](auto MM) {
                constexpr int m = decltype(MM)::value;
                constexpr int jj = 2 * m;
                const V2 p = Q[4 * sec + m], q = Q[4 * sec + m + 2];
                if constexpr (sec == 0 && m == 0) {
                    mul2(a0, h0, p.x, tl[cur][jj], th[cur][jj]);
                    mul2(a1, h1, q.x, tl[cur][jj], th[cur][jj]);
                } else {
                    fma2(a0, h0, p.x, tl[cur][jj], th[cur][jj]);
                    fma2(a1, h1, q.x, tl[cur][jj], th[cur][jj]);
                }
                fma2(a0, h0, p.y, tl[cur][jj + 1], th[cur][jj + 1]);
                fma2(a1, h1, q.y, tl[cur][jj + 1], th[cur][jj + 1]);
            });
        });
        // new ring rows 8*PH + rrow (ring planes are stored column-major: the column pass reads two rows per 16 bytes)
        {
            char* wr = lds_raw + PH * kNIR * ES + row_wr;
            *reinterpret_cast<T*>(wr) = lo0;
            *reinterpret_cast<T*>(wr + G::kColStride) = lo1;
            *reinterpret_cast<T*>(wr + G::kPlaneBytes) = hi0;
            *reinterpret_cast<T*>(wr + G::kPlaneBytes + G::kColStride) = hi1;
            if constexpr (PH == 0) {
                if (w == 0) {  // ring rows 0, 1 again behind row kRing-1
                    *reinterpret_cast<T*>(wr + kRing * ES) = lo0;
                    *reinterpret_cast<T*>(wr + kRing * ES + G::kColStride) = lo1;
                    *reinterpret_cast<T*>(wr + kRing * ES + G::kPlaneBytes) = hi0;
                    *reinterpret_cast<T*>(wr + kRing * ES + G::kPlaneBytes + G::kColStride) = hi1;
                }
            }
        }
        store_rows(buf ^ 1);
        if (s >= kLag && col_ok) {
            const int qq = 4 * (s - kLag) + rp;
            const size_t o = (size_t)(y0 + qq) * Nc2;  // uniform
            if (qq < nout) {
                st_sv(outL + o, ocol, a0);
                st_sv(outH + o, ocol, h0);
            }
            if (qq + 2 < nout) {
                st_sv(outL + o + 2 * (size_t)Nc2, ocol, a1);
                st_sv(outH + o + 2 * (size_t)Nc2, ocol, h1);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };

    for (int sb = 0; sb < nsteps; sb += kPhases) {
        bool fin = false;
        static_for<kPhases>([&](auto PHC) {
            if (!fin) {
                step(PHC, sb + decltype(PHC)::value);
                fin = (sb + decltype(PHC)::value + 1 >= nsteps);
            }
        });
        if constexpr ((kPhases * NSEC) % 2 == 1) {  // an odd number of sections per body: bring the tap buffers back in phase
#pragma unroll
            for (int jj = 0; jj < 8; jj++) {
                tl[0][jj] = tl[1][jj];
                th[0][jj] = th[1][jj];
            }
        }
    }
    clock_probe_stamp(probe, 1, probe_all);
}

// =================================================================================================
// host side
// =================================================================================================
#define PDWT_CHECK_LAUNCH() PDWT_HIP_TRY(hipGetLastError())

// filter lengths with an instantiation (sections of 8 taps; odd slot count of the staged rows): db4/sym4, db8/sym8,
// db12/sym12/coif4, db16/sym16, db20/sym20.  Measured against the kernels they replace (two levels, 8192^2, forward / inverse):
// db4 554/664 -> 283/378 us, db8 591/564 -> 312/421, db12 580/647 -> 442/435, db16 1089/1463 -> 477/495.
#define PDWT_F64LDS_HLENS(X) X(8) X(16) X(24) X(32) X(40)

// uneven split of chunk pairs between the two workgroups of a CU (lds_tile): only when the grid really puts two on every CU
static int lds_skew(int strips, int chunks)
{
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    if (!cus[dev]) {
        hipDeviceProp_t p;
        cus[dev] = hipGetDeviceProperties(&p, dev) == hipSuccess ? p.multiProcessorCount : -1;
    }
    const int T = strips * chunks, sk = knob(KN_F64_LDS_SKEW);
    if (sk <= 50 || sk >= 100 || cus[dev] <= 0 || chunks < 2 || chunks > 64) return 0;
    if (T <= cus[dev] || T > 2 * cus[dev]) return 0;
    return sk;
}

// HLEN = the instantiated length, hlen <= HLEN the filter's own (even) length: the bank is zero-padded SYMMETRICALLY, q = (HLEN-hlen)/2
// taps at either end.  out[i] = sum_j x[2i - C + j] F[hlen-1-j] with C = hlen/2 - 1 becomes the same sum over the padded window
// (C' = C + q, the original taps at positions q .. q+hlen-1); the extra terms are fma(x, 0, acc) = acc, so the result is the one
// of the unpadded filter bit for bit (finite data).
template <typename T, int HLEN>
static int launch_fwd_f64lds(const T* in, T* cA, T* cH, T* cV, T* cD, int nr, int nc, int hlen, const Taps2<T>& f, const void* d_tbl = nullptr,
                             int nimg = 1)
{
    using G = F64Lds<T, HLEN>;
    const int nr2 = div2(nr), nc2 = div2(nc);
    const int strips = idiv_up(nc2, kNCW);
    // two workgroups per CU when the level is large; one (steps run ~1.7x faster alone) when a chunk is mostly warm-up anyway
    // (a batch shares the target between its images: gridDim.y = image)
    const int target = ((long long)nr * nc * nimg >= 2048LL * 2048 ? knob(KN_F64_LDS_WGS) : knob(KN_F64_LDS_WGS) / 2) / nimg;
    int chunks = std::max(1, target / strips);
    int RO = idiv_up(idiv_up(nr2, chunks), 4) * 4;
    RO = std::max(RO, 4 * knob(KN_F64_LDS_MINGROUPS));
    chunks = idiv_up(nr2, RO);
    {  // > 64 KB of dynamic LDS is opt-in, per device (one process may drive several: wt_batch.h)
        static std::atomic<unsigned long long> done{0};
        int dev = 0;
        PDWT_HIP_TRY(hipGetDevice(&dev));
        const unsigned long long bit = 1ull << (dev & 63);
        if (!(done.load(std::memory_order_relaxed) & bit)) {
            PDWT_HIP_TRY(hipFuncSetAttribute((const void*)k_fwd2d_f64lds<T, HLEN>, hipFuncAttributeMaxDynamicSharedMemorySize, G::kLdsBytes));
            done.fetch_or(bit, std::memory_order_relaxed);
        }
    }
    TapTable<T> tt;  // window position j meets { L[hlen-1-j], H[hlen-1-j] } (SURVEY A-1: out[i] = sum_j x[2i-c+j] F[hlen-1-j])
    const int q = (HLEN - hlen) / 2;
    for (int j = 0; j < HLEN; j++) {
        const int k = j - q;
        const bool in_bank = k >= 0 && k < hlen;
        tt.t[2 * j] = in_bank ? f.a[hlen - 1 - k] : T(0);
        tt.t[2 * j + 1] = in_bank ? f.b[hlen - 1 - k] : T(0);
    }
    int pall = 0;
    unsigned long long* const pbuf = clock_probe_all(&pall);  // (diagnostic: every workgroup stamps its start and end)
    if (strips * chunks > kClockProbeAllBlocks) pall = 0;
    KTimer kt(K_FWD2D_F64);
    constexpr size_t lds = G::kLdsBytes;
    if (nimg > 1) pall = 0;
    hipLaunchKernelGGL((k_fwd2d_f64lds<T, HLEN>), dim3(strips * chunks, nimg), dim3(kNT), lds, stream(), tt, in, cA, cH, cV, cD, nr, nc, RO, strips,
                       pall == 1 ? pbuf : (nimg > 1 ? nullptr : clock_probe_slot(clock_probe_size_class(nr))), pall == 1 ? 1 : 0,
                       nimg > 1 ? 0 : lds_skew(strips, chunks), d_tbl);
    PDWT_CHECK_LAUNCH();
    return PDWT_OK;
}

// every EVEN filter length up to 40 runs the next instantiated length (zero-padded, see launch_fwd_f64lds)
static int f64lds_padded_len(int hlen) { return (hlen >= 2 && hlen <= 40 && !(hlen & 1)) ? (hlen + 7) / 8 * 8 : 0; }

// d_tbl != NULL: a batch of nimg images (five device pointers each) in one launch; small levels included (the size floor of the
// single-image path -- below it the per-level launch is latency-bound either way -- does not apply: batching is what fills the chip)
template <typename T>
static int fwd2d_lds_any(const T* in, T* cA, T* cH, T* cV, T* cD, int nr, int nc, int hlen, const Taps2<T>& f, const void* d_tbl = nullptr, int nimg = 1)
{
    const int hp = f64lds_padded_len(hlen);
    if (knob(KN_F64_LDS) < 1 || !hp || (hp != hlen && knob(KN_F64_LDS) == 3)) return 1;  // (3: exact lengths only)
    if (nr < 2 * kNIR || nr < hp || nc < hp) return 1;
    if (!d_tbl && (long long)nr * nc < (long long)knob(KN_F64_LDS_MIN) * knob(KN_F64_LDS_MIN)) return 1;
    switch (hp) {
#define X(H) \
    case H: return launch_fwd_f64lds<T, H>(in, cA, cH, cV, cD, nr, nc, hlen, f, d_tbl, nimg);
        PDWT_F64LDS_HLENS(X)
#undef X
        default: return 1;
    }
}

int fwd2d_f64_lds(const double* in, double* cA, double* cH, double* cV, double* cD, double* taps_dev, int nr, int nc, int hlen,
                  const Taps2<double>& f)
{
    (void)taps_dev;
    return fwd2d_lds_any<double>(in, cA, cH, cV, cD, nr, nc, hlen, f);
}

// float32: the cascade / streaming kernels own the banks of up to 16 taps on the geometries they take (they are traffic-bound
// there) and are tried first by the level driver; this is what runs otherwise -- longer banks (they ran the two-pass kernels, 2.67x
// the traffic, and were SLOWER than their double-precision counterparts once those had moved here) and short banks on odd or
// not-multiple-of-4 sizes (the LDS-tiled kernel before: 4095x4097 db4 L3 186 -> 116 us per pair, 1001x1003 69 -> 48)
int fwd2d_f64_lds_batch(const void* d_tbl, int nimg, int nr, int nc, int hlen, const Taps2<double>& f)
{
    if (!d_tbl || nimg < 1 || nimg > 65535) return 1;
    return fwd2d_lds_any<double>(nullptr, nullptr, nullptr, nullptr, nullptr, nr, nc, hlen, f, d_tbl, nimg);
}

int fwd2d_f32_lds(const float* in, float* cA, float* cH, float* cV, float* cD, int nr, int nc, int hlen, const Taps2<float>& f)
{
    if (hlen <= 16 && knob(KN_F64_LDS) == 4) return 1;  // (4: long banks only, for comparison)
    return fwd2d_lds_any<float>(in, cA, cH, cV, cD, nr, nc, hlen, f);
}

// =================================================================================================
// inverse level
// =================================================================================================
// Reference code replaced: w_kern_inverse_pass1 + w_kern_inverse_pass2 (src/separable.cu:246-328) of one iteration of
// w_inverse_separable (:332-364).  Order kept: columns first -- t1 = IL_y(A) + IH_y(H), t2 = IL_y(V) + IH_y(D) -- then rows
// out = IL_x(t1) + IH_x(t2).  SURVEY A-2 with H2 = hlen/2 taps per output, C = H2/2, SHIFT = 1 - (H2 & 1):
//   window position p (coefficient rows p-C .. p-C+H2-1) -> output rows 2p-SHIFT (tap parity 1) and 2p+1-SHIFT (parity 0);
//   the same along x: coefficient column c -> output columns 2c-SHIFT, 2c+1-SHIFT from t columns c-C .. c-C+H2-1.
//
// The first fused form kept rings of all four bands in every lane (184 VGPRs).  Here a thread owns ONE coefficient column of ONE band
// pair -- waves 0,1: (A, H) -> t1, waves 2,3: (V, D) -> t2 -- so its rings are half as large, and the row synthesis is register-blocked
// over two adjacent coefficient columns (21 16-byte LDS reads of (t1, t2) pairs feed 160 FMAs; the fused kernel reads 40):
//   * a workgroup owns 108 coefficient columns (+ H2-1 of halo = 127 threads per band pair) and walks down a chunk, two window
//     positions (four output rows) per step;
//   * COLUMN synthesis from the register rings (new coefficient rows are loaded straight into their ring slots, a whole body of
//     steps ahead), results (t1 | t2) to an LDS buffer of 4 rows;
//   * ROW synthesis of the PREVIOUS step's 4 rows in the same step (one barrier per step, the two passes share the section's taps):
//     thread = (row, pair of coefficient columns) -> 4 output samples of that row.  Rows are an odd number of 16-byte slots apart and
//     the lanes are dealt as in the forward kernel: no bank conflicts.
// Every output is (sum over the IL branch) + (sum over the IH branch), both ascending in the window position: the order of the
// two-pass kernels and of the oracle.
#ifndef PDWT_F64_ISB  // (experiments, tools/variant_lib.sh: steps per body / waves per SIMD the inverse kernel is compiled for)
#define PDWT_F64_ISB 4
#endif
#ifndef PDWT_F64_INV_WPS
#define PDWT_F64_INV_WPS 2
#endif
namespace {
constexpr int kISB = PDWT_F64_ISB;     // steps per unrolled body (ring slots are compile-time constants; the ring is shifted once per body)
}  // namespace

// NT threads per workgroup: NT/2 per band pair = coefficient columns whose t values the workgroup computes; H2-1 of them are halo.
// NT = 256: 108 columns produced at db20 (17.6 % of the column synthesis is halo), two workgroups per CU.  (NT = 512 -- 236 columns,
// 8 % halo, one workgroup per CU -- measured the same time and is not instantiated.)
template <typename T, int HLEN, int NT>
struct F64Inv {
    static constexpr int ES = sizeof(T);
    static constexpr int H2 = HLEN / 2, C = H2 / 2, SHIFT = (H2 & 1) ? 0 : 1;
    static constexpr int NCOL = NT / 2;
    static constexpr int INCW = (NCOL - (H2 - 1)) & ~1;  // coefficient columns a workgroup produces outputs for
    static constexpr int NPAIR = INCW / 2;
    static constexpr int WPP = NT / 128;                 // waves per window position in the row synthesis
    static constexpr int PPW = (NPAIR + WPP - 1) / WPP;  // column pairs per wave (<= 32)
    static constexpr int TSLOTS = NCOL + 1;              // (t1, t2) slots per row of the t buffer (odd: bank spreading)
    static constexpr int kTRowBytes = TSLOTS * 2 * ES;
    static constexpr int kTBufBytes = 4 * kTRowBytes;
    static constexpr int kLdsBytes = 2 * kTBufBytes;
    static constexpr int RS = H2 - 1 + 2 * kISB;         // ring slots
    static_assert(PPW <= 32 && H2 % 4 == 0 && TSLOTS % 2 == 1, "geometry");
};

template <typename T, int HLEN, int NT>
__global__ __launch_bounds__(NT, NT == 256 ? PDWT_F64_INV_WPS : 1) void k_inv2d_f64lds(TapTable<T> /*read through kernarg_taps()*/, const T* __restrict__ cA,
                                                          const T* __restrict__ cH, const T* __restrict__ cV,
                                                          const T* __restrict__ cD, T* __restrict__ out, int Nri, int Nci, int Nro, int Nco, int NP, int strips,
                                                          unsigned long long* probe, int probe_all, int skew, const void* tbl, int knob_idle)
{
    clock_probe_stamp(probe, 0, probe_all);
    if (tbl) {  // batched launch: (cA, cH, cV, cD, out) of this workgroup's image
        typedef const unsigned long long __attribute__((address_space(4))) * tbl_t;
        const tbl_t q = (tbl_t)(const unsigned long long*)tbl + 5 * (size_t)blockIdx.y;
        cA = (const T*)q[0];
        cH = (const T*)q[1];
        cV = (const T*)q[2];
        cD = (const T*)q[3];
        out = (T*)q[4];
    }
    using G = F64Inv<T, HLEN, NT>;
    using V2 = pair_t<T>;
    constexpr int ES = sizeof(T);
    constexpr int H2 = G::H2, C = G::C, SHIFT = G::SHIFT, RS = G::RS, kINCW = G::INCW;
    constexpr int NSEC = H2 / 4;
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    // (Nro, Nco): the output size, 2*Nri or 2*Nri - 1 (odd outputs: the synthesis runs on the even size, the last row / column is not stored)
    const int Nrv = 2 * Nri, Ncv = 2 * Nci;
    int strip, p0, np;
    lds_tile(strips, gridDim.x / strips, NP, Nri, skew, strip, p0, np);
    const int c0 = strip * kINCW;
    if (np <= 0) return;
    const int nsteps = (np + 1) >> 1;  // column-synthesis steps; one more step drains the row synthesis

    // ---- column-synthesis role: waves 0,1 -> (A, H), waves 2,3 -> (V, D); thread -> coefficient column c0 - C + lc
    const int pair = w / (NT / 128);
    const int lc = tid & (NT / 2 - 1);
    const T* const bL = pair ? cV : cA;
    const T* const bH = pair ? cD : cH;
    const unsigned ucc = (unsigned)ES * (unsigned)wrapi(c0 - C + lc, Nci);  // byte offset within a band row
    const int t_wr = lc * 2 * ES + pair * ES;
    // chunk-local coefficient row k <-> band row p0 - C + k; rows past the last one the chunk needs are clamped (never consumed)
    const int klast = 2 * nsteps - 1 + H2 - 1;
    // (single conditional wrap: -C <= p0 - C + k < Nri + H2 + 2*kISB, and the dispatcher only sends levels with Nri >= 2*H2 here)
    auto grow = [&](int k) { return (size_t)wrap1(p0 - C + min(k, klast), Nri) * Nci; };
    T r1[RS], r2[RS];
#pragma unroll
    for (int k = 0; k < RS; k++) r1[k] = r2[k] = T(0);
#pragma unroll
    for (int k = 0; k < H2 - 1; k++) {
        const size_t o = grow(k);
        r1[k] = ld_sv(bL + o, ucc);
        r2[k] = ld_sv(bH + o, ucc);
    }
    auto load_body = [&](int sb) {  // the 2*kISB coefficient rows the body's steps append, straight into their ring slots
        static_for<2 * kISB>([&](auto KK) {
            constexpr int k = decltype(KK)::value;
            const size_t o = grow(H2 - 1 + 2 * sb + k);
            r1[H2 - 1 + k] = ld_sv(bL + o, ucc);
            r2[H2 - 1 + k] = ld_sv(bH + o, ucc);
        });
    };

    // ---- row-synthesis role: wave -> (position of the step, column half), lane -> (parity row, pair of coefficient columns)
    const int rg = w / G::WPP, ch = w % G::WPP;
    const int rbit = (lane >> 3) & 1;
    const int cp = (((lane >> 5) & 1) << 4) | (((lane >> 2) & 1) << 3) | (((lane >> 4) & 1) << 2) | (lane & 3);
    const bool row_thread = cp < G::PPW && G::PPW * ch + cp < G::NPAIR;
    const int q = row_thread ? G::PPW * ch + cp : 0;         // column pair: coefficient columns c0 + 2q, c0 + 2q + 1
    // Lanes without a column pair (cp >= PPW: 5 of 32 at db20) still take part in the window reads.  ds_read_b128 is serviced in four
    // fixed 16-lane groups -- quads {0,3,5,6}, {1,2,4,7}, {8,11,13,14}, {9,10,12,15} of a wave (MI355X_MICROARCH.md, LDS) -- and a lane that
    // reads a DIFFERENT address on a busy bank costs its group a second cycle: parked on pair 0, the idle lanes did exactly that in two
    // of the four groups of every read (r05 counters: SQ_LDS_BANK_CONFLICT = 1.7 x SQ_ACTIVE_INST_LDS in this kernel).  They now read the
    // address of an active lane of their own group and row (identical addresses broadcast): cp 28..31 -> cp - 12 (quad 13 -> 8, 15 -> 10),
    // cp 27 -> 26 (same quad).
    const int cpr = row_thread ? cp : (cp >= 28 ? cp - 12 : G::PPW - 1);
    const int qr = (G::PPW * ch + cpr < G::NPAIR) ? G::PPW * ch + cpr : 0;
    const int t_rd = (2 * rg + rbit) * G::kTRowBytes + (knob_idle ? qr : q) * 4 * ES;  // window of column c0+2q starts at t slot 2q
    const int co = c0 + 2 * q;
    unsigned uq[4];
#pragma unroll
    for (int k = 0; k < 4; k++) uq[k] = (unsigned)ES * (unsigned)wrapi(2 * co - SHIFT + k, Ncv);  // byte offsets within an output row
    unsigned uqr[4];
#pragma unroll
    for (int k = 0; k < 4; k++) uqr[k] = uq[k] + (rbit ? (unsigned)ES * (unsigned)Nco : 0u);
    const unsigned colend = (unsigned)ES * (unsigned)Nco;  // (columns >= Nco exist only for odd Nco: the dropped one)
    const bool okA = row_thread && co < Nci, okB = row_thread && co + 1 < Nci;
    const bool ok0 = okA && uq[0] < colend, ok1 = okA && uq[1] < colend, ok2 = okB && uq[2] < colend, ok3 = okB && uq[3] < colend;

    ctaps_t<T> tbase = kernarg_taps<T>();
    T tp1l[2][4], tp0l[2][4], tp1h[2][4], tp0h[2][4];  // taps of the current / next section: IL parity 1, 0; IH parity 1, 0
#pragma unroll
    for (int jj = 0; jj < 4; jj++) {
        tp1l[0][jj] = tbase[4 * jj];
        tp0l[0][jj] = tbase[4 * jj + 1];
        tp1h[0][jj] = tbase[4 * jj + 2];
        tp0h[0][jj] = tbase[4 * jj + 3];
    }

    auto step = [&](auto UU, int s) {
        constexpr int U = decltype(UU)::value;  // step within the body: window positions 2U, 2U+1 of the body = ring slots 2U+pos+j
        const char* trow = lds_raw + ((s + 1) & 1) * G::kTBufBytes + t_rd;  // the previous step's rows
        T cs1[2], cg1[2], cs0[2], cg0[2];  // column synthesis [position]: IL/IH branch, parity 1/0
        T x1l[2], x1h[2], x0l[2], x0h[2];  // row synthesis [column of the pair]
        V2 P[H2 + 1];                         // (t1, t2) at window slots 0 .. H2 of the pair
        auto ldP = [&](auto MM) {
            constexpr int m = decltype(MM)::value;
            P[m] = *reinterpret_cast<const V2*>(trow + m * 2 * ES);
        };
        static_for<5>([&](auto MM) { ldP(MM); });
        static_for<NSEC>([&](auto SS) {
            constexpr int sec = decltype(SS)::value;
            constexpr int gsec = U * NSEC + sec;
            constexpr int cur = gsec & 1, nxt = cur ^ 1;
            constexpr int nsec = (sec + 1) % NSEC;
            ctaps_t<T> tp = tbase;
            if constexpr (sec == 0)
                asm volatile("" : "+s"(tp), "+v"(P[0]), "+v"(P[1]), "+v"(P[2]), "+v"(P[3]), "+v"(P[4]), "+s"(tp1l[cur][0]), "+s"(tp0h[cur][3]));
            else
                asm volatile(""
                             : "+s"(tp), "+v"(cs1[0]), "+v"(cg1[0]), "+v"(cs0[0]), "+v"(cg0[0]), "+v"(cs1[1]), "+v"(cg1[1]), "+v"(cs0[1]), "+v"(cg0[1]),
                               "+v"(x1l[0]), "+v"(x1h[0]), "+v"(x0l[0]), "+v"(x0h[0]), "+v"(x1l[1]), "+v"(x1h[1]), "+v"(x0l[1]), "+v"(x0h[1]),
                               "+v"(P[4 * sec + 1]), "+v"(P[4 * sec + 2]), "+v"(P[4 * sec + 3]), "+v"(P[4 * sec + 4]), "+s"(tp1l[cur][0]),
                               "+s"(tp0h[cur][3]));
#pragma unroll
            for (int jj = 0; jj < 4; jj++) {
                tp1l[nxt][jj] = tp[4 * (nsec * 4 + jj)];
                tp0l[nxt][jj] = tp[4 * (nsec * 4 + jj) + 1];
                tp1h[nxt][jj] = tp[4 * (nsec * 4 + jj) + 2];
                tp0h[nxt][jj] = tp[4 * (nsec * 4 + jj) + 3];
            }
            if constexpr (sec + 1 < NSEC) {
                static_for<4>([&](auto MM) { ldP(std::integral_constant<int, 4 * (sec + 1) + 1 + decltype(MM)::value>{}); });
            }
            __builtin_amdgcn_sched_barrier(0);
            static_for<4>([&](auto JJ) {
                constexpr int jj = decltype(JJ)::value;
                constexpr int j = 4 * sec + jj;
                // column synthesis of the two window positions of this step
                static_for<2>([&](auto PP) {
                    constexpr int pos = decltype(PP)::value;
                    constexpr int slot = 2 * U + pos + j;
                    if constexpr (j == 0) {  // fma(x, t, 0) == x * t: no zero-initialisation moves
                        mul2(cs1[pos], cs0[pos], r1[slot], tp1l[cur][jj], tp0l[cur][jj]);
                        mul2(cg1[pos], cg0[pos], r2[slot], tp1h[cur][jj], tp0h[cur][jj]);
                    } else {
                        fma2(cs1[pos], cs0[pos], r1[slot], tp1l[cur][jj], tp0l[cur][jj]);
                        fma2(cg1[pos], cg0[pos], r2[slot], tp1h[cur][jj], tp0h[cur][jj]);
                    }
                });
                // row synthesis of the previous step's rows: the pair's columns read window slots j and j+1
                static_for<2>([&](auto KK) {
                    constexpr int k = decltype(KK)::value;
                    const V2 t = P[j + k];
                    if constexpr (j == 0) {
                        mul2(x1l[k], x0l[k], t.x, tp1l[cur][jj], tp0l[cur][jj]);
                        mul2(x1h[k], x0h[k], t.y, tp1h[cur][jj], tp0h[cur][jj]);
                    } else {
                        fma2(x1l[k], x0l[k], t.x, tp1l[cur][jj], tp0l[cur][jj]);
                        fma2(x1h[k], x0h[k], t.y, tp1h[cur][jj], tp0h[cur][jj]);
                    }
                });
            });
        });
        // this step's four rows of t: [position 0: parity 1, parity 0][position 1: parity 1, parity 0]
        {
            char* tw = lds_raw + (s & 1) * G::kTBufBytes + t_wr;
            *reinterpret_cast<T*>(tw) = cs1[0] + cg1[0];
            *reinterpret_cast<T*>(tw + G::kTRowBytes) = cs0[0] + cg0[0];
            *reinterpret_cast<T*>(tw + 2 * G::kTRowBytes) = cs1[1] + cg1[1];
            *reinterpret_cast<T*>(tw + 3 * G::kTRowBytes) = cs0[1] + cg0[1];
        }
        // outputs of the previous step: window position p -> output rows 2p-SHIFT (+rbit); coefficient columns co, co+1 -> 4 columns
        if (s >= 1) {
            const int pl = 2 * (s - 1) + rg;  // chunk-local window position
            if (pl < np) {
                const int row0 = wrap1(2 * (p0 + pl) - SHIFT, Nrv);  // parity-1 row; the parity-0 row is the next one (periodic)
                const T o1a = x1l[0] + x1h[0], o0a = x0l[0] + x0h[0], o1b = x1l[1] + x1h[1], o0b = x0l[1] + x0h[1];
                if (row0 + 1 < Nrv) {  // uniform base + per-lane offset (the lane's row is part of the offset)
                    T* orow = out + (size_t)row0 * Nco;
                    if (row0 + rbit < Nro) {
                        if (ok0) st_sv(orow, uqr[0], o1a);
                        if (ok1) st_sv(orow, uqr[1], o0a);
                        if (ok2) st_sv(orow, uqr[2], o1b);
                        if (ok3) st_sv(orow, uqr[3], o0b);
                    }
                } else if (rbit || row0 < Nro) {  // rows Nrv-1 (dropped when the output height is odd) and 0
                    T* orow = out + (rbit ? (size_t)0 : (size_t)row0 * Nco);
                    if (ok0) st_flat(orow + uq[0] / ES, o1a);
                    if (ok1) st_flat(orow + uq[1] / ES, o0a);
                    if (ok2) st_flat(orow + uq[2] / ES, o1b);
                    if (ok3) st_flat(orow + uq[3] / ES, o0b);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };

    for (int sb = 0; sb <= nsteps; sb += kISB) {
        load_body(sb);
        bool fin = false;
        static_for<kISB>([&](auto UU) {
            if (!fin) {
                step(UU, sb + decltype(UU)::value);
                fin = (sb + decltype(UU)::value + 1 > nsteps);
            }
        });
#pragma unroll
        for (int k = 0; k < H2 - 1; k++) {
            r1[k] = r1[k + 2 * kISB];
            r2[k] = r2[k + 2 * kISB];
        }
        if constexpr ((kISB * NSEC) % 2 == 1) {
#pragma unroll
            for (int jj = 0; jj < 4; jj++) {
                tp1l[0][jj] = tp1l[1][jj];
                tp0l[0][jj] = tp0l[1][jj];
                tp1h[0][jj] = tp1h[1][jj];
                tp0h[0][jj] = tp0h[1][jj];
            }
        }
    }
    clock_probe_stamp(probe, 1, probe_all);
}

// (zero-padded like the forward bank: out[n] = sum_k c[k] IL[n - 2k + hlen/2 - 1], so q = (HLEN-hlen)/2 zeros in FRONT of the bank
// keep every product where it was)
template <typename T, int HLEN>
static int launch_inv_f64lds(const T* cA, const T* cH, const T* cV, const T* cD, T* out, int nri, int nci, int nro, int nco, int hlen,
                             const Taps2<T>& f, const void* d_tbl = nullptr, int nimg = 1)
{
    const bool big = (long long)nro * nco * nimg >= 2048LL * 2048;
    const int strips = idiv_up(nci, F64Inv<T, HLEN, 256>::INCW);
    const int wgs = knob(KN_EXP1) > 0 ? knob(KN_EXP1) : knob(KN_F64_LDS_WGS);  // (exp1: workgroup target of the INVERSE alone)
    const int target = (big ? wgs : wgs / 2) / nimg;
    int chunks = std::max(1, target / strips);
    int NP = idiv_up(idiv_up(nri, chunks), 2) * 2;
    NP = std::max(NP, 2 * knob(KN_F64_LDS_MINGROUPS));
    chunks = idiv_up(nri, NP);
    TapTable<T> tt;  // window position j meets { IL[h-2-2j], IL[h-1-2j], IH[h-2-2j], IH[h-1-2j] } (parity 1 / parity 0)
    const int q = (HLEN - hlen) / 2;
    auto pad = [&](const T* bank, int t) { return (t - q >= 0 && t - q < hlen) ? bank[t - q] : T(0); };
    for (int j = 0; j < HLEN / 2; j++) {
        tt.t[4 * j + 0] = pad(f.a, HLEN - 2 - 2 * j);
        tt.t[4 * j + 1] = pad(f.a, HLEN - 1 - 2 * j);
        tt.t[4 * j + 2] = pad(f.b, HLEN - 2 - 2 * j);
        tt.t[4 * j + 3] = pad(f.b, HLEN - 1 - 2 * j);
    }
    int pall = 0;
    unsigned long long* const pbuf = clock_probe_all(&pall);
    if (strips * chunks > kClockProbeAllBlocks) pall = 0;
    KTimer kt(K_INV2D_F64);
    constexpr size_t lds256 = F64Inv<T, HLEN, 256>::kLdsBytes;
    if (nimg > 1) pall = 0;
    hipLaunchKernelGGL((k_inv2d_f64lds<T, HLEN, 256>), dim3(strips * chunks, nimg), dim3(256), lds256, stream(), tt, cA, cH, cV, cD, out, nri, nci, nro, nco, NP, strips,
                       pall == 2 ? pbuf : (nimg > 1 ? nullptr : clock_probe_slot(8 + clock_probe_size_class(nro))), pall == 2 ? 1 : 0,
                       nimg > 1 ? 0 : lds_skew(strips, chunks), d_tbl, knob(KN_EXP0) == 0 ? 1 : 0);  // (exp0 = 1: idle lanes parked on pair 0, the round-4 form)
    PDWT_CHECK_LAUNCH();
    return PDWT_OK;
}

template <typename T>
static int inv2d_lds_any(const T* cA, const T* cH, const T* cV, const T* cD, T* out, int nri, int nci, int nro, int nco, int hlen,
                         const Taps2<T>& f, const void* d_tbl = nullptr, int nimg = 1)
{
    const int hp = f64lds_padded_len(hlen);
    if (knob(KN_F64_LDS) < 1 || !hp || (hp != hlen && knob(KN_F64_LDS) == 3)) return 1;
    if (nri != div2(nro) || nci != div2(nco) || nri < hp || nci < 2) return 1;  // (row indices wrap at most once: wrap1)
    if (!d_tbl && (long long)nro * nco < (long long)knob(KN_F64_LDS_MIN) * knob(KN_F64_LDS_MIN)) return 1;
    switch (hp) {
#define X(H) \
    case H: return launch_inv_f64lds<T, H>(cA, cH, cV, cD, out, nri, nci, nro, nco, hlen, f, d_tbl, nimg);
        PDWT_F64LDS_HLENS(X)
#undef X
        default: return 1;
    }
}

int inv2d_f64_lds(const double* cA, const double* cH, const double* cV, const double* cD, double* out, double* taps_dev, int nri, int nci,
                  int nro, int nco, int hlen, const Taps2<double>& f)
{
    (void)taps_dev;
    return inv2d_lds_any<double>(cA, cH, cV, cD, out, nri, nci, nro, nco, hlen, f);
}

int inv2d_f64_lds_batch(const void* d_tbl, int nimg, int nri, int nci, int nro, int nco, int hlen, const Taps2<double>& f)
{
    if (!d_tbl || nimg < 1 || nimg > 65535) return 1;
    return inv2d_lds_any<double>(nullptr, nullptr, nullptr, nullptr, nullptr, nri, nci, nro, nco, hlen, f, d_tbl, nimg);
}

int inv2d_f32_lds(const float* cA, const float* cH, const float* cV, const float* cD, float* out, int nri, int nci, int nro, int nco,
                  int hlen, const Taps2<float>& f)
{
    if (hlen <= 16 && knob(KN_F64_LDS) == 4) return 1;
    return inv2d_lds_any<float>(cA, cH, cV, cD, out, nri, nci, nro, nco, hlen, f);
}

}  // namespace pdwt
