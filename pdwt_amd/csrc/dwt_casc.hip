// dwt_casc.hip -- TWO levels of the 2D DWT per launch, float32, streaming form ("cascade").
//
// After the level kernels of dwt_stream.hip reached the copy roofline, what is left of a multi-level
// transform is (a) the approximation of level l written to HBM and read back by level l+1 and (b) the
// launch of level l+1, which is latency-bound at its size (profiles/: 8.5-10 us for 32 MB).  Here a wave
// that streams down its strip of level l feeds the A row it has just produced straight into a SECOND
// register ring and emits level l+1 from registers:
//   lane = 4 input columns -> 2 columns of level l (A1,H1,V1,D1) -> 1 column of level l+1 (A2,H2,V2,D2)
//   every 2 input rows: one row of H1,V1,D1 (8-byte stores); the A1 pair goes through the level-(l+1) row
//   pass (halo from neighbouring lanes by DPP, as for the input) into ring2;
//   every 4 input rows: one row of A2,H2,V2,D2 (4-byte stores).
// Halo: NB1 lanes (input) + NB2 lanes (A1) per side -> 58 producing lanes of 64 for hlen 8; vertically a
// chunk of level-(l+1) rows recomputes hlen-2 rows of A1 (and re-reads 3(hlen-2) input rows).
// The loop issues every store unconditionally: lanes that own no output are masked through EXEC, rows
// outside the chunk's own range and the ring warm-up are stored with EXEC = 0 (forward; they still count in vmcnt, in order:
// tools/probes/vmcnt_order.hip) or go to a trash row (inverse), so the hand-counted s_waitcnt pipeline of
// dwt_stream.hip carries over with per-position constants (casc_fwd_after).
// Arithmetic per sample = the single-level kernels' (row pass then column pass, taps ascending, one FMA
// per tap), so the result is bit-identical to running the two levels separately.
// Reference code replaced: two iterations of the level loop of w_forward_separable / w_inverse_separable
// (src/separable.cu:179-209, 332-364) with their four kernels each.
#include "dwt_casc.hpp"

#include <algorithm>

#include "dwt_stream.hpp"
#include "stream_dev.hpp"
#include "casc_dev.hpp"

namespace pdwt {

// forward: VMEM instructions a wave issues per A1 row `p` of a super-body after that row's two loads
constexpr int casc_fwd_stores(int p) { return 3 + 4 * (p & 1); }
// ... and between the second load issued at position p and the next use of those registers
// (position p + DIST, DIST = A1 rows of prefetch distance)
template <int DIST>
constexpr int casc_fwd_after(int p)
{
    int n = casc_fwd_stores(p);
    for (int k = 1; k < DIST; k++) n += 2 + casc_fwd_stores(p + k);
    return n;
}

// LDS hand-off area of a W-wave workgroup: one region per CONSUMER wave k in [0, W-1), written by wave k+1:
//   (HLEN-2) ring rows of 64 lanes x 16 B  (level-1 row-pass results of the producer's first HLEN-2 input rows)
//   (HLEN-2) ring2 rows of 64 lanes x 8 B  (level-2 row-pass results of the producer's first HLEN-2 A1 rows)
template <int HLEN>
constexpr int casc_fwd_region_bytes() { return (HLEN - 2) * 64 * (16 + 8); }

// The forward column passes of an 8-tap bank as ONE asm statement each (cf. col_synth4x4 in dwt_casc_inv3.hip: between separate statements that
// depend on each other hipcc inserts a wait state it does not need).  (A, H) += lo * (L[k], H[k]) and (V, D) += hi * (L[k], H[k]) for the
// two columns of a lane, ring entries r?0 / r?1 = (lo, hi) of column 0 / 1 in window order, taps t0..t7 in window order.
#define PDWT_VB_F0 " op_sel_hi:[0,1,0]"
#define PDWT_VB_N0 " op_sel_hi:[0,1,1]"
#define PDWT_VB_F1 " op_sel:[1,0,0] op_sel_hi:[1,1,0]"
#define PDWT_VB_N1 " op_sel:[1,0,0]"
// WITH_VD = false: only the (A, H) sums (rows whose H, V, D the wave does not store: the V, D sums would be dead)
template <bool WITH_VD>
__device__ __forceinline__ void col_pass8x2(v2f& ah0, v2f& ah1, v2f& vd0, v2f& vd1, v2f a0, v2f a1, v2f a2, v2f a3, v2f a4, v2f a5, v2f a6, v2f a7, v2f b0, v2f b1,
                                            v2f b2, v2f b3, v2f b4, v2f b5, v2f b6, v2f b7, v2f t0, v2f t1, v2f t2, v2f t3, v2f t4, v2f t5, v2f t6, v2f t7)
{
#define PDWT_CP_ROW(A, B, T, AC)                                                                                              \
    "v_pk_fma_f32 %0, " A ", " T ", " AC "0\n\tv_pk_fma_f32 %1, " B ", " T ", " AC "1\n\t"
    if constexpr (WITH_VD) {
        asm("v_pk_fma_f32 %0, %4, %20, 0" PDWT_VB_F0 "\n\tv_pk_fma_f32 %1, %12, %20, 0" PDWT_VB_F0 "\n\tv_pk_fma_f32 %2, %4, %20, 0" PDWT_VB_F1 "\n\tv_pk_fma_f32 %3, %12, %20, 0" PDWT_VB_F1
            "\n\tv_pk_fma_f32 %0, %5, %21, %0" PDWT_VB_N0 "\n\tv_pk_fma_f32 %1, %13, %21, %1" PDWT_VB_N0 "\n\tv_pk_fma_f32 %2, %5, %21, %2" PDWT_VB_N1 "\n\tv_pk_fma_f32 %3, %13, %21, %3" PDWT_VB_N1
            "\n\tv_pk_fma_f32 %0, %6, %22, %0" PDWT_VB_N0 "\n\tv_pk_fma_f32 %1, %14, %22, %1" PDWT_VB_N0 "\n\tv_pk_fma_f32 %2, %6, %22, %2" PDWT_VB_N1 "\n\tv_pk_fma_f32 %3, %14, %22, %3" PDWT_VB_N1
            "\n\tv_pk_fma_f32 %0, %7, %23, %0" PDWT_VB_N0 "\n\tv_pk_fma_f32 %1, %15, %23, %1" PDWT_VB_N0 "\n\tv_pk_fma_f32 %2, %7, %23, %2" PDWT_VB_N1 "\n\tv_pk_fma_f32 %3, %15, %23, %3" PDWT_VB_N1
            "\n\tv_pk_fma_f32 %0, %8, %24, %0" PDWT_VB_N0 "\n\tv_pk_fma_f32 %1, %16, %24, %1" PDWT_VB_N0 "\n\tv_pk_fma_f32 %2, %8, %24, %2" PDWT_VB_N1 "\n\tv_pk_fma_f32 %3, %16, %24, %3" PDWT_VB_N1
            "\n\tv_pk_fma_f32 %0, %9, %25, %0" PDWT_VB_N0 "\n\tv_pk_fma_f32 %1, %17, %25, %1" PDWT_VB_N0 "\n\tv_pk_fma_f32 %2, %9, %25, %2" PDWT_VB_N1 "\n\tv_pk_fma_f32 %3, %17, %25, %3" PDWT_VB_N1
            "\n\tv_pk_fma_f32 %0, %10, %26, %0" PDWT_VB_N0 "\n\tv_pk_fma_f32 %1, %18, %26, %1" PDWT_VB_N0 "\n\tv_pk_fma_f32 %2, %10, %26, %2" PDWT_VB_N1 "\n\tv_pk_fma_f32 %3, %18, %26, %3" PDWT_VB_N1
            "\n\tv_pk_fma_f32 %0, %11, %27, %0" PDWT_VB_N0 "\n\tv_pk_fma_f32 %1, %19, %27, %1" PDWT_VB_N0 "\n\tv_pk_fma_f32 %2, %11, %27, %2" PDWT_VB_N1 "\n\tv_pk_fma_f32 %3, %19, %27, %3" PDWT_VB_N1
            : "=&v"(ah0), "=&v"(ah1), "=&v"(vd0), "=&v"(vd1)
            : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7), "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(b4), "v"(b5), "v"(b6), "v"(b7),
              "s"(t0), "s"(t1), "s"(t2), "s"(t3), "s"(t4), "s"(t5), "s"(t6), "s"(t7));
    } else {
        asm("v_pk_fma_f32 %0, %2, %18, 0" PDWT_VB_F0 "\n\tv_pk_fma_f32 %1, %10, %18, 0" PDWT_VB_F0
            "\n\tv_pk_fma_f32 %0, %3, %19, %0" PDWT_VB_N0 "\n\tv_pk_fma_f32 %1, %11, %19, %1" PDWT_VB_N0
            "\n\tv_pk_fma_f32 %0, %4, %20, %0" PDWT_VB_N0 "\n\tv_pk_fma_f32 %1, %12, %20, %1" PDWT_VB_N0
            "\n\tv_pk_fma_f32 %0, %5, %21, %0" PDWT_VB_N0 "\n\tv_pk_fma_f32 %1, %13, %21, %1" PDWT_VB_N0
            "\n\tv_pk_fma_f32 %0, %6, %22, %0" PDWT_VB_N0 "\n\tv_pk_fma_f32 %1, %14, %22, %1" PDWT_VB_N0
            "\n\tv_pk_fma_f32 %0, %7, %23, %0" PDWT_VB_N0 "\n\tv_pk_fma_f32 %1, %15, %23, %1" PDWT_VB_N0
            "\n\tv_pk_fma_f32 %0, %8, %24, %0" PDWT_VB_N0 "\n\tv_pk_fma_f32 %1, %16, %24, %1" PDWT_VB_N0
            "\n\tv_pk_fma_f32 %0, %9, %25, %0" PDWT_VB_N0 "\n\tv_pk_fma_f32 %1, %17, %25, %1" PDWT_VB_N0
            : "=&v"(ah0), "=&v"(ah1)
            : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7), "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(b4), "v"(b5), "v"(b6), "v"(b7),
              "s"(t0), "s"(t1), "s"(t2), "s"(t3), "s"(t4), "s"(t5), "s"(t6), "s"(t7));
    }
#undef PDWT_CP_ROW
}
// the level-2 column pass: one column per lane, (A2, H2) += lo * taps, (V2, D2) += hi * taps
__device__ __forceinline__ void col_pass8x1(v2f& ah, v2f& vd, v2f a0, v2f a1, v2f a2, v2f a3, v2f a4, v2f a5, v2f a6, v2f a7, v2f t0, v2f t1, v2f t2, v2f t3, v2f t4,
                                            v2f t5, v2f t6, v2f t7)
{
    asm("v_pk_fma_f32 %0, %2, %10, 0" PDWT_VB_F0 "\n\tv_pk_fma_f32 %1, %2, %10, 0" PDWT_VB_F1
        "\n\tv_pk_fma_f32 %0, %3, %11, %0" PDWT_VB_N0 "\n\tv_pk_fma_f32 %1, %3, %11, %1" PDWT_VB_N1
        "\n\tv_pk_fma_f32 %0, %4, %12, %0" PDWT_VB_N0 "\n\tv_pk_fma_f32 %1, %4, %12, %1" PDWT_VB_N1
        "\n\tv_pk_fma_f32 %0, %5, %13, %0" PDWT_VB_N0 "\n\tv_pk_fma_f32 %1, %5, %13, %1" PDWT_VB_N1
        "\n\tv_pk_fma_f32 %0, %6, %14, %0" PDWT_VB_N0 "\n\tv_pk_fma_f32 %1, %6, %14, %1" PDWT_VB_N1
        "\n\tv_pk_fma_f32 %0, %7, %15, %0" PDWT_VB_N0 "\n\tv_pk_fma_f32 %1, %7, %15, %1" PDWT_VB_N1
        "\n\tv_pk_fma_f32 %0, %8, %16, %0" PDWT_VB_N0 "\n\tv_pk_fma_f32 %1, %8, %16, %1" PDWT_VB_N1
        "\n\tv_pk_fma_f32 %0, %9, %17, %0" PDWT_VB_N0 "\n\tv_pk_fma_f32 %1, %9, %17, %1" PDWT_VB_N1
        : "=&v"(ah), "=&v"(vd)
        : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7), "s"(t0), "s"(t1), "s"(t2), "s"(t3), "s"(t4), "s"(t5), "s"(t6), "s"(t7));
}

// Bookkeeping of a straight-line wave program (see the kernel): R2 level-2 rows, last wave of its workgroup or not, DIST A1 rows
// of prefetch.  Everything here is a function of the A1 row index alone.
template <int HLEN, int R2, bool LV, int DIST>
struct CascSpec {
    static constexpr int NA = 2 * R2, NA1 = NA + HLEN - 2, NL1 = NA - HLEN / 2 + 1;
    static constexpr int LIM = LV ? NA1 : NL1;  // A1 rows below LIM take their two new input rows from memory
    static constexpr int stores_at(int q) { return ((q < NA) ? 3 : 0) + (((q & 1) && q >= HLEN - 1) ? 4 : 0); }
    static constexpr int loads_at(int q) { return (q < LIM && q + DIST < LIM) ? 2 : 0; }
    // VMEM instructions between the loads issued at A1 row n - DIST and their use at A1 row n
    static constexpr int wait_cnt(int n)
    {
        const int m = n - DIST;
        int c = stores_at(m);
        for (int k = 1; k < DIST; k++) c += loads_at(m + k) + stores_at(m + k);
        return c;
    }
};

// NV = row registers = input rows (KiB) in flight per wave = prefetch distance (HLEN/2, HLEN or 2*HLEN)
// W  = waves of a workgroup stacked vertically in ONE strip (1 = independent waves, four strips per workgroup).
//
// Chunk-local numbering (all W): local input row 0 = global 4*j0 - 3C for a wave whose level-2 rows are [j0, j0+rows2);
// A1 row n <-> local input rows 2n .. 2n+HLEN-1; level-2 row jl <-> A1 rows 2jl .. 2jl+HLEN-1.  A wave OWNS (stores) the
// level-1 rows n in [0, NA), NA = 2*rows2 -- global row 2*j0 + n - C -- and the level-2 rows jl in [0, rows2).  Its last
// HLEN/2-1 own A1 rows need HLEN-2 input rows beyond its 2*NA first-hand ones, its last HLEN/2-1 level-2 rows need HLEN-2
// A1 rows beyond NA: exactly the rows the wave BELOW starts with.
//   W == 1 (and the last wave of a workgroup): recompute -- load the 3(HLEN-2) extra input rows, run NA1 = NA+HLEN-2 A1 rows.
//   W  > 1: the wave below has those row-pass results in its rings anyway (its ring warm-up).  It drops them in LDS during
//           its first super-body; one s_barrier later every wave can pick up its bottom halo at the END of its chunk: no
//           extra loads, no recomputation, and the waves of a workgroup are balanced by giving the last one fewer rows.
// SPEC = true: the kernel consists of the straight-line wave programs only (the host launches it when every wave of the geometry has one)
template <int HLEN, int NV, int W, bool SPEC = false>
__global__ __launch_bounds__(W == 1 ? 256 : 64 * W) void k_fwd2d_casc(const float* in, CascBands b, int Nr, int Nc, int VL,
                                                                       float* __restrict__ trash, CascMap cm, TapsLH f)
{
    using G = CascGeom<HLEN>;
    constexpr int C = G::C, NB1 = G::NB1, NB2 = G::NB2, NBT = G::NBT, WIN1 = G::WIN1, WIN2 = G::WIN2;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    CASC_TRACE_DECL;
    CASC_TRACE(0);
    if constexpr (W > 1) {
        if (cm.tbl) {  // batched launch: this workgroup's image -- read through the constant address space: scalar loads, pointers in SGPRs
            static_assert(sizeof(CascBatchF) == 8 * sizeof(void*), "eight pointers per image");
            typedef const unsigned long long __attribute__((address_space(4))) * tbl_t;
            const tbl_t q = (tbl_t)(const unsigned long long*)cm.tbl + 8 * (size_t)blockIdx.y;
            in = (const float*)q[0];
            b = CascBands{(float*)q[1], (float*)q[2], (float*)q[3], (float*)q[4], (float*)q[5], (float*)q[6], (float*)q[7]};
        }
    }
    const int lane = threadIdx.x & 63;
    const int Nc2 = Nc >> 1, Nr2 = Nr >> 1, Nr4 = Nr >> 2, Nc4 = Nc >> 2;
    // the wave index is uniform, but only readfirstlane tells the compiler: everything derived from it (rows, row
    // bases, trip counts, predicates) then lives on the scalar unit
    const int kw = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int j0, rows2, strip;
    bool last;  // this wave recomputes its bottom halo (nobody hands it over)
    if constexpr (W == 1) {
        // wave -> (chunk row, strip): XCD x (= blockIdx % 8, private L2) owns the contiguous band of chunk rows
        // [x*cpx, (x+1)*cpx) and its waves walk that band strip by strip, so neighbours in space are neighbours in time
        const int xcd = blockIdx.x & 7;
        const int wi = (blockIdx.x >> 3) * 4 + kw;
        if (wi >= cm.cpx * cm.strips) return;
        const int cy = xcd * cm.cpx + wi / cm.strips;
        strip = wi % cm.strips;
        const int nchunks = 8 * cm.cpx;
        j0 = (int)(((long long)cy * Nr4) / nchunks);  // level-2 rows [j0, j1) of the chunk
        rows2 = (int)(((long long)(cy + 1) * Nr4) / nchunks) - j0;
        if (rows2 <= 0) return;
        last = true;
    } else {
        // workgroup -> (workgroup-chunk row, strip); XCD x owns the logical workgroups [x*cpx, (x+1)*cpx)
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const int wg = xcd * cm.cpx + slot;
        if (slot >= cm.cpx || wg >= cm.gy * cm.strips) return;  // (uniform over the workgroup: nobody is left at the barrier)
        const int gy = wg / cm.strips;
        strip = wg % cm.strips;
        const int J0 = casc_chunk_start(gy, strip, Nr4, cm.gy, cm.strips, cm.cpx, cm.flags);
        const int R = casc_chunk_start(gy + 1, strip, Nr4, cm.gy, cm.strips, cm.cpx, cm.flags) - J0;
        // split of the R level-2 rows: the last wave also streams the 3(HLEN-2) halo input rows below the workgroup's
        // chunk (worth E level-2 rows of work), so it gets E rows fewer (at least one is left); the host guarantees
        // R / W >= HLEN / 2 (two barriers in the first super-body order the hand-off, see the loop)
        const int E = min((3 * (HLEN - 2) + 3) / 4, R / W - 1);
        const int base = (R + E) / W, rem = (R + E) % W;
        const int start = kw * base + min(kw, rem);
        rows2 = (kw < W - 1) ? base + (kw < rem ? 1 : 0) : R - start;
        j0 = J0 + start;
        last = (kw == W - 1);
    }
    const int xs = strip * VL * 4;  // first input column this strip produces outputs for
    const int x = xs + 4 * (lane - NBT);
    const bool valid = (lane >= NBT) && (lane < NBT + VL) && (x < Nc);
    const int xo = wrapi(x, Nc);
    const int yb = 4 * j0 - 3 * C;           // global input row of chunk-local row 0
    const int NA = 2 * rows2;                // A1 rows the wave owns
    const int NA1 = NA + HLEN - 2;           // A1 rows it runs through (the last HLEN-2 only feed level 2)
    // W > 1, not the last wave: A1 rows >= NL1 take their two new input rows from the hand-off area, A1 rows >= NA are not
    // computed at all (their level-2 row-pass results come from the hand-off area)
    const int NL1 = NA - HLEN / 2 + 1;
    const int rlast = last ? 2 * NA1 + HLEN - 3 : 2 * NA - 1;  // last input row the wave loads

    v2f ring[HLEN][2];  // level 1: (lo,hi) row-pass results of the last HLEN input rows, 2 columns
    v2f ring2[HLEN];    // level 2: (lo,hi) row-pass results of the last HLEN A1 rows, 1 column
#pragma unroll
    for (int k = 0; k < HLEN; k++) ring2[k] = v2f{0.f, 0.f};

    const float* const lbase = in + xo;
    auto rowptr = [&](int r) { return lbase + (size_t)wrap1(yb + r, Nr) * Nc; };
    const unsigned xoff = (unsigned)xo * 4u;

    // hand-off area: region kw is READ by this wave (written by wave kw+1), region kw-1 is WRITTEN by it
    constexpr int REG = casc_fwd_region_bytes<HLEN>();
    unsigned char* const lds_rd = lds_raw + (size_t)(W > 1 ? kw : 0) * REG;
    unsigned char* const lds_wr = lds_raw + (size_t)(W > 1 && kw > 0 ? kw - 1 : 0) * REG;
    auto lds_ring = [&](unsigned char* reg, int r) { return reinterpret_cast<v4f*>(reg + ((size_t)r * 64 + lane) * 16); };
    auto lds_ring2 = [&](unsigned char* reg, int r) { return reinterpret_cast<v2f*>(reg + (size_t)(HLEN - 2) * 64 * 16 + ((size_t)r * 64 + lane) * 8); };
    auto row_pass1 = [&](const v4f& v, v2f (&lh)[2]) {
        float w[WIN1];
#pragma unroll
        for (int q = 0; q < 4; q++) w[NB1 * 4 + q] = v[q];
#pragma unroll
        for (int k = 0; k < NB1; k++) {
            const int dl = (NB1 - 1 - k) * 4, sl = (NB1 - k) * 4, dr = (NB1 + 1 + k) * 4, sr = (NB1 + k) * 4;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                w[dl + q] = dpp_shr1(w[sl + q]);
                w[dr + q] = dpp_shl1(w[sr + q]);
            }
        }
#pragma unroll
        for (int p = 0; p < 2; p++) {
            v2f acc = {0.f, 0.f};
#pragma unroll
            for (int j = 0; j < HLEN; j++) acc = pk_fma(splat(w[NB1 * 4 - C + 2 * p + j]), f.t[HLEN - 1 - j], acc);
            lh[p] = acc;
        }
    };
    auto row_pass2 = [&](float a0, float a1, v2f& lh) {
        float w[WIN2];
        w[NB2 * 2] = a0;
        w[NB2 * 2 + 1] = a1;
#pragma unroll
        for (int k = 0; k < NB2; k++) {
            const int dl = (NB2 - 1 - k) * 2, sl = (NB2 - k) * 2, dr = (NB2 + 1 + k) * 2, sr = (NB2 + k) * 2;
#pragma unroll
            for (int q = 0; q < 2; q++) {
                w[dl + q] = dpp_shr1(w[sl + q]);
                w[dr + q] = dpp_shl1(w[sr + q]);
            }
        }
        v2f acc = {0.f, 0.f};
#pragma unroll
        for (int j = 0; j < HLEN; j++) acc = pk_fma(splat(w[NB2 * 2 - C + j]), f.t[HLEN - 1 - j], acc);
        lh = acc;
    };

    // ring prologue rows 0..HLEN-3 and the first body's rows, issued together
    constexpr int DIST = NV / 2;    // prefetch distance in A1 rows
    static_assert(casc_fwd_after<DIST>(1) <= 63, "vmcnt is a 6-bit counter");
    v4f v[NV];
    {
        v4f pv[HLEN > 2 ? HLEN - 2 : 1];
#pragma unroll
        for (int r = 0; r < HLEN - 2; r++) pv[r] = *reinterpret_cast<const v4f*>(rowptr(r));
#pragma unroll
        for (int u = 0; u < NV; u++) v[u] = *reinterpret_cast<const v4f*>(rowptr(min(HLEN - 2 + u, rlast)));
        static_for<HLEN - 2>([&](auto Rr) {
            constexpr int r = decltype(Rr)::value;
            row_pass1(pv[r], ring[r]);
        });
    }
    if constexpr (W > 1) {
        // hand over the ring warm-up rows: the wave above needs exactly these at the end of its chunk
        if (kw > 0) {
#pragma unroll
            for (int r = 0; r < HLEN - 2; r++) *lds_ring(lds_wr, r) = v4f{ring[r][0].x, ring[r][0].y, ring[r][1].x, ring[r][1].y};
        }
    }

    const unsigned off1 = (unsigned)(valid ? x >> 1 : 0) * 4u, off2 = (unsigned)(valid ? x >> 2 : 0) * 4u;
    const lanemask_t vmask = __ballot(valid);
    // Scalar bookkeeping.  The scalar unit, not the vector ALUs or memory, bounds this kernel (40 extra scalar instructions per A1 row:
    // +3.5 us at C2; 20 extra vector instructions: +0.75), so nothing per row is recomputed from indices:
    //  * loads walk consecutive rows: a 64-bit row pointer advanced by the row stride (reset at the image's last row, frozen at the
    //    wave's last row), instead of wrap + 64-bit multiply per load;
    //  * stores take the band base (loop-invariant SGPR pair) and a per-lane offset = lane offset + 32-bit row offset (one v_add per
    //    row; the host sends images whose level-1 bands exceed 4 GiB to the level kernels), the row offsets advance by the band's row
    //    stride; rows the wave does not own are stored with EXEC = 0 (no trash row, no select per band);
    //  * the three / four stores of a row share one exec save / restore (asm_store3_sm / asm_store4_sm).
    const int lim1 = ((W == 1) || last) ? 0x7fffffff : NL1;  // A1 rows below lim1 take their input rows from memory ...
    const int lim2 = ((W == 1) || last) ? 0x7fffffff : NA;   // ... below lim2 they are computed here
    // prefetch cursor: rows come in pairs (even, odd chunk-local row; rlast is odd, so a pair is either two new rows or -- past the
    // wave's last row -- twice that row, a cached re-read); the image wrap (at most once per wave) takes the slow branch
    int lr = HLEN - 2 + NV;                                   // chunk-local row of the next pair (even)
    int lgr = wrap1(yb + min(lr, rlast), Nr);                 // its image row
    const size_t strideB = (size_t)Nc * 4;                    // (byte pointers: no shift per advance)
    const char* const inB = reinterpret_cast<const char*>(in);
    const char* lp = inB + (size_t)lgr * strideB;
    const char* lplast = lp;                                  // the last row fetched (what a frozen pair re-reads)
    auto next_rows = [&](const char*& p0, const char*& p1) {
        int lrq = lr;
        asm("" : "+s"(lrq));  // (a compare of its own: hipcc otherwise keeps the predicate as a lane mask, 3 scalar instructions per use)
        if (lrq < rlast) {
            p0 = lp;
            if (lgr + 2 < Nr) {
                p1 = lp + strideB;
                lp = p1 + strideB;
                lgr += 2;
            } else {
                p1 = (lgr + 1 == Nr) ? inB : lp + strideB;
                lgr = lgr + 2 - Nr;
                lp = inB + (size_t)lgr * strideB;
            }
            lplast = p1;
        } else {
            p0 = p1 = lplast;
        }
        lr += 2;
    };
    unsigned s1off = (unsigned)wrap1(2 * j0 - C, Nr2) * (unsigned)Nc2 * 4u;  // byte offset of the level-1 row of A1 row 0 inside a band
    const unsigned s1end = (unsigned)Nr2 * (unsigned)Nc2 * 4u;               // (bands of 4 GiB and more never get here)
    unsigned s2off = (unsigned)j0 * (unsigned)Nc4 * 4u;       // byte offset of the wave's next own level-2 row
    (void)trash;
    const int hand_sb = __builtin_amdgcn_readfirstlane((W > 1 && kw > 0) ? 0 : -1);  // the super-body in which the ring2 rows are handed up
    CASC_TRACE(1);  // ring prologue computed (its loads landed)
    static_for<NV>([&](auto K) { asm_drain1(v[decltype(K)::value]); });
    CASC_TRACE(2);  // first body's rows landed
    auto a1_row = [&](auto A, int sb) {
            constexpr int a = decltype(A)::value;  // A1 row within the super-body
            constexpr int u = a % (HLEN / 2);
            constexpr int s0 = (2 * u + HLEN - 2) % HLEN, s1 = (2 * u + HLEN - 1) % HLEN;
            const int n = sb * HLEN + a;  // chunk-local A1 row
            constexpr int r0 = (2 * a) % NV, r1 = r0 + 1;
            // (every use of a predicate is a scalar compare of its own -- laundered copies of n: hipcc otherwise carries the predicates
            // as lane masks from use to use, ~40 scalar instructions per A1 row)
            int nq1 = n, nq2 = n;
            asm("" : "+s"(nq1));
            asm("" : "+s"(nq2));
            if (nq2 < lim2) {  // the A1 row is computed here (a nested if / else tree: two sequential ifs made hipcc thread the cases through mask flags)
                if (nq1 < lim1) {
                    // its two new input rows come from memory: v[r0], v[r1] were loaded DIST A1 rows ago, at position (a - DIST) mod HLEN
                    asm_wait2<casc_fwd_after<DIST>((a + HLEN - DIST) % HLEN)>(v[r0], v[r1]);
                    row_pass1(v[r0], ring[s0]);
                    row_pass1(v[r1], ring[s1]);
                } else {
                    if constexpr (W > 1) {  // ... or from the wave below (its ring warm-up rows)
                        const int i0 = 2 * (n - NL1);
                        const v4f q0 = *lds_ring(lds_rd, i0), q1 = *lds_ring(lds_rd, i0 + 1);
                        ring[s0][0] = v2f{q0.x, q0.y};
                        ring[s0][1] = v2f{q0.z, q0.w};
                        ring[s1][0] = v2f{q1.x, q1.y};
                        ring[s1][1] = v2f{q1.z, q1.w};
                    }
                }
                // rows 2n + HLEN-2 + NV and the next one, DIST A1 rows ahead (frozen at the wave's last row: the prefetch past the end
                // re-reads a cached line instead of fetching new ones)
                {
                    const char *p0, *p1;
                    next_rows(p0, p1);
                    asm_load_s(v[r0], reinterpret_cast<const float*>(p0), xoff);
                    asm_load_s(v[r1], reinterpret_cast<const float*>(p1), xoff);
                }
                // level-1 column pass
                v2f ah[2], vd[2];
#pragma unroll
                for (int p = 0; p < 2; p++) ah[p] = vd[p] = v2f{0.f, 0.f};
                static_for<HLEN>([&](auto J) {
                    constexpr int j = decltype(J)::value;
                    constexpr int s = (2 * u + j) % HLEN;
                    const v2f t = f.t[HLEN - 1 - j];
#pragma unroll
                    for (int p = 0; p < 2; p++) {
                        ah[p] = pk_fma(splat(ring[s][p].x), t, ah[p]);
                        vd[p] = pk_fma(splat(ring[s][p].y), t, vd[p]);
                    }
                });
                {
                    // rows the wave does not own (the recomputed halo) are stored with every lane off
                    asm_store3_sm(b.H1, b.V1, b.D1, off1 + s1off, v2f{ah[0].y, ah[1].y}, v2f{vd[0].x, vd[1].x}, v2f{vd[0].y, vd[1].y},
                                  n < NA ? vmask : 0ull);
                    s1off += (unsigned)Nc2 * 4u;
                    if (s1off == s1end) s1off = 0;
                }
                // level-2 row pass on the A1 pair; ring2 slot = n % HLEN = a
                row_pass2(ah[0].x, ah[1].x, ring2[a]);
                if constexpr (W > 1 && a < HLEN - 2) {
                    // first super-body: the wave above needs the row-pass results of this wave's first HLEN-2 A1 rows
                    if (sb == hand_sb) *lds_ring2(lds_wr, a) = ring2[a];  // (sb == 0 and not the first wave)
                }
            } else {
                if constexpr (W > 1) ring2[a] = *lds_ring2(lds_rd, min(n - NA, HLEN - 3));
            }
            if constexpr (a & 1) {
                // A1 rows n-HLEN+1 .. n complete the window of level-2 row (n-(HLEN-1))/2
                v2f ah2 = {0.f, 0.f}, vd2 = {0.f, 0.f};
                static_for<HLEN>([&](auto J) {
                    constexpr int j = decltype(J)::value;
                    constexpr int s = (a + 1 + j) % HLEN;
                    const v2f t = f.t[HLEN - 1 - j];
                    ah2 = pk_fma(splat(ring2[s].x), t, ah2);
                    vd2 = pk_fma(splat(ring2[s].y), t, vd2);
                });
                const bool own = (unsigned)(n - (HLEN - 1)) < (unsigned)(2 * rows2);  // level-2 row (n - (HLEN-1)) / 2 of the wave's rows2
                asm_store4_sm(b.A2, b.H2, b.V2, b.D2, off2 + s2off, ah2.x, ah2.y, vd2.x, vd2.y, own ? vmask : 0ull);
                s2off += own ? (unsigned)Nc4 * 4u : 0u;
            }
    };
    // ---- straight-line wave programs (W = 16, one A1 row of prefetch) -------------------------------------------------------------
    // A wave of this form lives for 16-odd A1 rows: it is all prologue and epilogue, and in the loop below every A1 row pays ~45
    // scalar instructions for predicates and cursors (where does the row come from, is it owned, has the image wrapped, is the wave
    // past its last row) on a kernel that is bound by instruction issue (+40 scalar instructions per A1 row = +3.5 us).
    // For the row counts the default geometry produces (non-last waves with R2 level-2 rows, the last wave with its reduced count)
    // the whole wave program is instantiated with the A1 row index as a compile-time constant: every predicate folds away, stores
    // of rows the wave does not own and the arithmetic that only feeds them are not emitted, the vmcnt waits are constants of the
    // position, and the only wave-dependent addressing left is a 64-bit row pointer (+ 2 rows per A1 row) and two running store
    // offsets.  The two places where a wave's rows wrap around the image are compile-time positions as well: the first wave of
    // the top workgroups (local input row 3C, level-1 row C) and the last wave of the bottom workgroups (local row 4*R2 + 3C).
    // Same arithmetic, same order: bit-identical to the loop (tests: test_forward_cascade_row_cursors).
    // (a wave program has consumed every row it loaded: only stores are in flight at its end, and nothing has to wait for them)
    auto epilogue = [&]() __attribute__((always_inline)) {
        CASC_TRACE(5);  // loop left
#ifdef PDWT_CASC_TRACE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        CASC_TRACE(6);  // everything this wave issued has retired
        CASC_TRACE_STORE(trash, blockIdx.x * W + kw, ((unsigned long long)NA1 << 32) | (unsigned)rows2);
    };
    if constexpr (SPEC) {
        static_assert(W == 16 && (NV == 2 || NV == 4), "wave programs exist for the 16-wave workgroups");
        const unsigned xoff_odd = xoff + (unsigned)strideB;  // second row of a pair: same scalar base, the row stride in the lane offset
        auto spec = [&](auto R2c, auto LASTc) __attribute__((always_inline)) {
            constexpr int R2 = decltype(R2c)::value;
            constexpr bool LV = decltype(LASTc)::value;
            using S = CascSpec<HLEN, R2, LV, DIST>;
            constexpr int sNA = S::NA, sNA1 = S::NA1, sNL1 = S::NL1;
            constexpr int WR = LV ? 4 * R2 + 3 * C : 3 * C;  // local input row at which the image wraps, for the waves that wrap at all
            const bool wraps = LV ? (j0 + R2 == Nr4) : (j0 == 0);
            const size_t wrapB = wraps ? (size_t)Nr * strideB : (size_t)0;
            const unsigned s1wrap = (!LV && j0 == 0) ? s1end : 0u;
            const char* sp = lp;  // local row HLEN (already wrapped: the prologue's row arithmetic)
            static_for<sNA1>([&](auto Nn) {
                constexpr int n = decltype(Nn)::value;
                constexpr int a = n % HLEN;
                constexpr bool from_mem = LV ? true : (n < sNL1);
                constexpr bool computed = LV ? true : (n < sNA);
                constexpr bool store1 = n < sNA;
                constexpr int s0 = (2 * n + HLEN - 2) % HLEN, s1 = (2 * n + HLEN - 1) % HLEN;
                if constexpr (computed) {
                    if constexpr (from_mem) {
                        constexpr int q0 = (2 * n) % NV, q1 = q0 + 1;
                        // (the first DIST A1 rows use the rows the prologue loaded and drained)
                        if constexpr (n >= DIST) asm_wait2<S::wait_cnt(n)>(v[q0], v[q1]);
                        row_pass1(v[q0], ring[s0]);
                        row_pass1(v[q1], ring[s1]);
                        if constexpr (n + DIST < S::LIM) {
                            constexpr int r0 = 2 * (n + DIST) + HLEN - 2;  // sp points at local row r0
                            // (the row registers start a new life here: without this the tied loads make hipcc carry the dead old values
                            // into whatever registers it picked for the new ones, two v_mov_b64 per load)
                            asm volatile("" : "=v"(v[q0]));
                            asm volatile("" : "=v"(v[q1]));
                            asm_load_s(v[q0], reinterpret_cast<const float*>(sp), xoff);
                            if constexpr (r0 + 1 == WR) {
                                asm_load_s(v[q1], reinterpret_cast<const float*>(sp + strideB - wrapB), xoff);
                            } else {
                                asm_load_s(v[q1], reinterpret_cast<const float*>(sp), xoff_odd);
                            }
                            sp += 2 * strideB;
                            if constexpr (r0 + 1 == WR || r0 + 2 == WR) sp -= wrapB;
                        }
                    } else {
                        constexpr int i0 = 2 * (n - sNL1);
                        const v4f q0 = *lds_ring(lds_rd, i0), q1 = *lds_ring(lds_rd, i0 + 1);
                        ring[s0][0] = v2f{q0.x, q0.y};
                        ring[s0][1] = v2f{q0.z, q0.w};
                        ring[s1][0] = v2f{q1.x, q1.y};
                        ring[s1][1] = v2f{q1.z, q1.w};
                    }
                    v2f ah[2], vd[2];
#pragma unroll
                    for (int p = 0; p < 2; p++) ah[p] = vd[p] = v2f{0.f, 0.f};
                    if constexpr (HLEN == 8) {
                        constexpr int q = (2 * n) % 8;  // ring slot of window position 0
#define PDWT_R(j, p) ring[(q + (j)) % 8][p]
                        col_pass8x2<store1>(ah[0], ah[1], vd[0], vd[1], PDWT_R(0, 0), PDWT_R(1, 0), PDWT_R(2, 0), PDWT_R(3, 0), PDWT_R(4, 0), PDWT_R(5, 0), PDWT_R(6, 0),
                                            PDWT_R(7, 0), PDWT_R(0, 1), PDWT_R(1, 1), PDWT_R(2, 1), PDWT_R(3, 1), PDWT_R(4, 1), PDWT_R(5, 1), PDWT_R(6, 1), PDWT_R(7, 1),
                                            f.t[7], f.t[6], f.t[5], f.t[4], f.t[3], f.t[2], f.t[1], f.t[0]);
#undef PDWT_R
                    } else {
                        static_for<HLEN>([&](auto J) {
                            constexpr int j = decltype(J)::value;
                            constexpr int s = (2 * n + j) % HLEN;
                            const v2f t = f.t[HLEN - 1 - j];
#pragma unroll
                            for (int p = 0; p < 2; p++) {
                                ah[p] = pk_fma_vbcast<0, j == 0>(ring[s][p], t, ah[p]);
                                if constexpr (store1) vd[p] = pk_fma_vbcast<1, j == 0>(ring[s][p], t, vd[p]);
                            }
                        });
                    }
                    if constexpr (store1) {
                        asm_store3_sm(b.H1, b.V1, b.D1, off1 + s1off, v2f{ah[0].y, ah[1].y}, v2f{vd[0].x, vd[1].x}, v2f{vd[0].y, vd[1].y}, vmask);
                        s1off += (unsigned)Nc2 * 4u;
                        if constexpr (n == C - 1) s1off -= s1wrap;
                    }
                    row_pass2(ah[0].x, ah[1].x, ring2[a]);
                    if constexpr (n < HLEN - 2) {
                        if (kw > 0) *lds_ring2(lds_wr, n) = ring2[a];
                    }
                } else {
                    ring2[a] = *lds_ring2(lds_rd, n - sNA);
                }
                if constexpr ((n & 1) && n >= HLEN - 1) {
                    v2f ah2 = {0.f, 0.f}, vd2 = {0.f, 0.f};
                    if constexpr (HLEN == 8) {
                        constexpr int q = (a + 1) % 8;
                        col_pass8x1(ah2, vd2, ring2[q], ring2[(q + 1) % 8], ring2[(q + 2) % 8], ring2[(q + 3) % 8], ring2[(q + 4) % 8], ring2[(q + 5) % 8],
                                    ring2[(q + 6) % 8], ring2[(q + 7) % 8], f.t[7], f.t[6], f.t[5], f.t[4], f.t[3], f.t[2], f.t[1], f.t[0]);
                    } else {
                        static_for<HLEN>([&](auto J) {
                            constexpr int j = decltype(J)::value;
                            constexpr int s = (a + 1 + j) % HLEN;
                            const v2f t = f.t[HLEN - 1 - j];
                            ah2 = pk_fma_vbcast<0, j == 0>(ring2[s], t, ah2);
                            vd2 = pk_fma_vbcast<1, j == 0>(ring2[s], t, vd2);
                        });
                    }
                    asm_store4_sm(b.A2, b.H2, b.V2, b.D2, off2 + s2off, ah2.x, ah2.y, vd2.x, vd2.y, vmask);
                    s2off += (unsigned)Nc4 * 4u;
                }
                // the two hand-off barriers of the first super-body (see the loop below)
                if constexpr (n == HLEN / 2 - 1 || n == HLEN - 1) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

#ifdef PDWT_CASC_TRACE
                if constexpr (n == HLEN / 2 - 1) CASC_TRACE(3);
                if constexpr (n == HLEN - 1) CASC_TRACE(4);
#endif
            });
        };
        // The wave programs live in a kernel of their own: next to the loop in ONE kernel, hipcc's control-flow structuriser chains the arms
        // (arm 1 -> flag -> arm 2 ... -> loop) and the register allocator keeps every arm's entry state alive, in scratch, around the others.
        // Every arm but the last ends the program itself (s_endpgm instead of a return to a common exit) for the same reason.
        const int variant = __builtin_amdgcn_readfirstlane(last ? 3 : (rows2 == 5 ? 2 : 1));
        if (variant == 1) {
            spec(std::integral_constant<int, 4>{}, std::false_type{});
            epilogue();
            __builtin_amdgcn_endpgm();
        }
        if (variant == 2) {
            spec(std::integral_constant<int, 5>{}, std::false_type{});
            epilogue();
            __builtin_amdgcn_endpgm();
        }
        spec(std::integral_constant<int, 1>{}, std::true_type{});
        epilogue();
    } else {
    for (int sb = 0;; sb++) {
        static_for<HLEN / 2>([&](auto U) { a1_row(std::integral_constant<int, decltype(U)::value>{}, sb); });
        // Hand-off order (W > 1): the ring rows are written in the prologue and first read at A1 row NL1 >= HLEN/2, the ring2 rows
        // are written at A1 rows 0..HLEN-3 and first read at A1 row NA >= HLEN -- one barrier after each half of the FIRST
        // super-body (every wave runs both halves: NA1 >= HLEN for rows2 >= 1).  LDS only: the global loads in flight are not drained.
        if constexpr (W > 1) {
            if (sb == 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
#ifdef PDWT_CASC_TRACE
        if (sb == 0) CASC_TRACE(3);  // first half super-body done
#endif
        if (sb * HLEN + HLEN / 2 >= NA1) break;
        static_for<HLEN / 2>([&](auto U) { a1_row(std::integral_constant<int, decltype(U)::value + HLEN / 2>{}, sb); });
        if constexpr (W > 1) {
            if (sb == 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
#ifdef PDWT_CASC_TRACE
        if (sb == 0) CASC_TRACE(4);  // first super-body done
#endif
        if (sb * HLEN + HLEN >= NA1) break;
    }
    static_for<NV>([&](auto K) { asm_drain1(v[decltype(K)::value]); });
    epilogue();
    }
}

// =================================================================================================
// inverse: levels l+1 and l in one launch
// =================================================================================================
// lane = 1 coefficient column of level l+1 -> 2 columns of A_l (which never goes to memory) = the lane's 2
// coefficient columns of level l -> 4 output columns.  One STEP = one new coefficient row of level l+1:
//   level-(l+1) column synthesis for both tap parities -> DPP halo of (t1,t2) -> row synthesis -> two rows of
//   A_l; each of them enters the level-l ring together with the H,V,D row loaded from memory and yields two
//   output rows (column synthesis, DPP halo, row synthesis, one 16-byte store each) -- the arithmetic of
//   k_inv2d_stream, twice.
// A super-body = H2 steps brings both rings back to slot 0, so every slot is a compile-time constant.  The
// row registers of a step are re-issued for the same step ONE SUPER-BODY AHEAD; all loads/stores are inline
// asm with exact s_waitcnt counts (14 VMEM per step: 4 loads, then per A_l row 3 loads + 2 stores).
// The recomputed halo lands on the coarse level (4x less data), so it is cheap in this direction.
// PFD = prefetch distance in steps (row registers are re-issued for the step PFD steps ahead; H2 % PFD == 0)
template <int HLEN, int PFD>
__global__ __launch_bounds__(256) void k_inv2d_casc(CascInvBands b, float* __restrict__ out, int Nr, int Nc, int VL,
                                                     float* __restrict__ trash, CascMap cm, Taps2<float> f)
{
    using G = CascInvGeom<HLEN>;
    constexpr int H2 = G::H2, C = G::C, SHIFT = G::SHIFT, NB1 = G::NB1, NB2 = G::NB2, NBT = G::NBT, WIN1 = G::WIN1, WIN2 = G::WIN2;
    constexpr int VM_PF = PFD * (4 + 2 * (3 + 2));  // VMEM instructions per prefetch period
    static_assert(VM_PF - 3 <= 63, "vmcnt is a 6-bit counter");
    static_assert(H2 % PFD == 0 || PFD % H2 == 0, "register slots must be compile-time constants");
    const int lane = threadIdx.x & 63;
    const int xcd = blockIdx.x & 7;
    // the wave index is uniform, but only readfirstlane tells the compiler: everything derived from it (rows, row
    // bases, trip counts, predicates) then lives on the scalar unit
    const int wi = (blockIdx.x >> 3) * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wi >= cm.cpx * cm.strips) return;
    const int cy = xcd * cm.cpx + wi / cm.strips;
    const int strip = wi % cm.strips;
    const int Nr1 = Nr >> 1, Nc1 = Nc >> 1, Nr2 = Nr >> 2, Nc2 = Nc >> 2;
    const int nchunks = 8 * cm.cpx;
    // level-l coefficient rows [ya, ye) of the chunk (mod Nr1): even heights, common start parity BASE
    const int ya = G::BASE + 2 * (int)(((long long)cy * (Nr1 >> 1)) / nchunks);
    const int ye = G::BASE + 2 * (int)(((long long)(cy + 1) * (Nr1 >> 1)) / nchunks);
    const int rows1 = ye - ya;
    if (rows1 <= 0) return;
    const int cx1 = strip * VL * 2 + 2 * (lane - NBT);  // first of the lane's two level-l coefficient columns
    const bool valid = (lane >= NBT) && (lane < NBT + VL) && (cx1 < Nc1);
    const int cx1w = wrapi(cx1, Nc1);
    const int cx2w = cx1w >> 1;
    const int gA = ya - C;              // A_l row of stream index r1 = 0
    const int P0 = (gA + SHIFT) >> 1;   // level-(l+1) window position of step 0 (exact division: see BASE)
    const int r2base = P0 - C;          // level-(l+1) coefficient row of r2 = 0
    const int nrows1 = rows1 + H2 - 1 + SHIFT;
    const int nsteps = (nrows1 + 1) >> 1;

    const float* const pA2 = b.A2 + cx2w;
    const float* const pH2 = b.H2 + cx2w;
    const float* const pV2 = b.V2 + cx2w;
    const float* const pD2 = b.D2 + cx2w;
    const float* const pH1 = b.H1 + cx1w;
    const float* const pV1 = b.V1 + cx1w;
    const float* const pD1 = b.D1 + cx1w;
    auto off2 = [&](int r2) { return (size_t)wrap1(r2base + r2, Nr2) * Nc2; };
    auto off1 = [&](int r1) { return (size_t)wrap1(gA + r1, Nr1) * Nc1; };
    // steady state: uniform row bases (scalar unit) + loop-invariant per-lane byte offsets (stream_dev.hpp)
    const unsigned voff2 = (unsigned)cx2w * 4u, voff1 = (unsigned)cx1w * 4u, voffo = (unsigned)(valid ? cx1 : 0) * 8u;
    const lanemask_t vmask = __ballot(valid);

    v2f r2av[H2], r2hd[H2];               // level l+1 ring: (A,V) and (H,D) of the lane's column
    v2f ra[H2], rh[H2], rv[H2], rd[H2];   // level l ring: the lane's two columns of each band
#pragma unroll
    for (int k = 0; k < H2; k++) ra[k] = rh[k] = rv[k] = rd[k] = v2f{0.f, 0.f};
    float q2[PFD][4];                     // row registers in flight, level l+1 (one per step of a prefetch period)
    v2f q1[2 * PFD][3];                   // row registers in flight, level l (two per step)
    {
#pragma unroll
        for (int r = 0; r < H2 - 1; r++) {
            const size_t o = off2(r);
            r2av[r] = v2f{pA2[o], pV2[o]};
            r2hd[r] = v2f{pH2[o], pD2[o]};
        }
        r2av[H2 - 1] = r2hd[H2 - 1] = v2f{0.f, 0.f};
#pragma unroll
        for (int p = 0; p < PFD; p++) {
            const size_t o = off2(H2 - 1 + p);
            q2[p][0] = pA2[o];
            q2[p][1] = pH2[o];
            q2[p][2] = pV2[o];
            q2[p][3] = pD2[o];
        }
#pragma unroll
        for (int q = 0; q < 2 * PFD; q++) {
            const size_t o = off1(q);
            q1[q][0] = *reinterpret_cast<const v2f*>(pH1 + o);
            q1[q][1] = *reinterpret_cast<const v2f*>(pV1 + o);
            q1[q][2] = *reinterpret_cast<const v2f*>(pD1 + o);
        }
    }

    float* const tr = trash + (size_t)(blockIdx.x & 7) * Nc;  // a trash ROW (the dispatcher checks the area holds 8 of them)

    // one output row of level l from the ring window starting at slot S0 with tap parity OFF (cf. k_inv2d_stream::emit)
    auto emit = [&](auto S0, auto OFF, int g) {
        constexpr int s0 = decltype(S0)::value, off = decltype(OFF)::value;
        // rows of the ring warm-up / beyond the chunk: the store is still issued (to a trash row, the VMEM count must not
        // change) but the arithmetic is skipped -- a uniform branch, 8 of the ~46 emits of a wave
        const bool own = (g >= 0) && (g < 2 * rows1);
        float o4[4] = {0.f, 0.f, 0.f, 0.f};
        if (own) {
            v2f sa = {0.f, 0.f}, sh = {0.f, 0.f}, sv = {0.f, 0.f}, sd = {0.f, 0.f};
            static_for<H2>([&](auto J) {
                constexpr int j = decltype(J)::value;
                constexpr int s = (s0 + j) % H2;
                constexpr int k = HLEN - 1 - (2 * j + off);
                constexpr int kp = k & ~1;  // tap k as one half of its aligned pair: no splat copies in scalar registers (dwt_stream.hip)
                const v2f pl2 = v2f{f.a[kp], f.a[kp + 1]}, ph2 = v2f{f.b[kp], f.b[kp + 1]};
                sa = pk_fma_sbcast<k & 1, false>(ra[s], pl2, sa);
                sh = pk_fma_sbcast<k & 1, false>(rh[s], ph2, sh);
                sv = pk_fma_sbcast<k & 1, false>(rv[s], pl2, sv);
                sd = pk_fma_sbcast<k & 1, false>(rd[s], ph2, sd);
            });
            const v2f t1o = sa + sh, t2o = sv + sd;
            float t1[WIN1], t2[WIN1];
            t1[NB1 * 2] = t1o.x;
            t1[NB1 * 2 + 1] = t1o.y;
            t2[NB1 * 2] = t2o.x;
            t2[NB1 * 2 + 1] = t2o.y;
    #pragma unroll
            for (int k = 0; k < NB1; k++) {
                const int dl = (NB1 - 1 - k) * 2, sl = (NB1 - k) * 2, dr = (NB1 + 1 + k) * 2, sr = (NB1 + k) * 2;
    #pragma unroll
                for (int cc = 0; cc < 2; cc++) {
                    t1[dl + cc] = dpp_shr1(t1[sl + cc]);
                    t2[dl + cc] = dpp_shr1(t2[sl + cc]);
                    t1[dr + cc] = dpp_shl1(t1[sr + cc]);
                    t2[dr + cc] = dpp_shl1(t2[sr + cc]);
                }
            }
            auto pair_out = [&](auto E0) {
                constexpr int e0 = decltype(E0)::value;
                constexpr int pl = (e0 + SHIFT) >> 1;
                v2f s1 = {0.f, 0.f}, s2 = {0.f, 0.f};
    #pragma unroll
                for (int j = 0; j < H2; j++) {
                    const int m = HLEN - 1 - 2 * j;
                    s1 = pk_fma(splat(t1[NB1 * 2 + pl - C + j]), v2f{f.a[m - 1], f.a[m]}, s1);
                    s2 = pk_fma(splat(t2[NB1 * 2 + pl - C + j]), v2f{f.b[m - 1], f.b[m]}, s2);
                }
                const v2f o = s1 + s2;
                o4[e0] = o.x;
                o4[e0 + 1] = o.y;
            };
            auto single_out = [&](auto E) {
                constexpr int eo = decltype(E)::value;
                constexpr int gp = eo + SHIFT;
                constexpr int pl = gp >> 1, offx = 1 - (gp & 1);
                float s1 = 0.f, s2 = 0.f;
    #pragma unroll
                for (int j = 0; j < H2; j++) {
                    const int k = HLEN - 1 - (2 * j + offx);
                    s1 = __builtin_fmaf(t1[NB1 * 2 + pl - C + j], f.a[k], s1);
                    s2 = __builtin_fmaf(t2[NB1 * 2 + pl - C + j], f.b[k], s2);
                }
                o4[eo] = s1 + s2;
            };
            if constexpr (SHIFT == 0) {
                pair_out(std::integral_constant<int, 0>{});
                pair_out(std::integral_constant<int, 2>{});
            } else {
                single_out(std::integral_constant<int, 0>{});
                pair_out(std::integral_constant<int, 1>{});
                single_out(std::integral_constant<int, 3>{});
            }
        }
        asm_store_sm(own ? out + (size_t)wrap1(2 * ya + g, Nr) * Nc : tr, voffo, v4f{o4[0], o4[1], o4[2], o4[3]}, vmask);
    };

    // one row of A_l (the lane's two columns) from the level-(l+1) ring window starting at slot S0, tap parity OFF
    auto synth2 = [&](auto S0, auto OFF, float& a0, float& a1) {
        constexpr int s0 = decltype(S0)::value, off = decltype(OFF)::value;
        v2f sav = {0.f, 0.f}, shd = {0.f, 0.f};
        static_for<H2>([&](auto J) {
            constexpr int j = decltype(J)::value;
            constexpr int s = (s0 + j) % H2;
            constexpr int k = HLEN - 1 - (2 * j + off);
            constexpr int kp = k & ~1;  // tap k as one half of its aligned pair: no splat copies in scalar registers (dwt_stream.hip)
            sav = pk_fma_sbcast<k & 1, false>(r2av[s], v2f{f.a[kp], f.a[kp + 1]}, sav);
            shd = pk_fma_sbcast<k & 1, false>(r2hd[s], v2f{f.b[kp], f.b[kp + 1]}, shd);
        });
        const v2f t = sav + shd;  // (t1, t2) of the lane's column
        float t1[WIN2], t2[WIN2];
        t1[NB2] = t.x;
        t2[NB2] = t.y;
#pragma unroll
        for (int k = 0; k < NB2; k++) {
            t1[NB2 - 1 - k] = dpp_shr1(t1[NB2 - k]);
            t2[NB2 - 1 - k] = dpp_shr1(t2[NB2 - k]);
            t1[NB2 + 1 + k] = dpp_shl1(t1[NB2 + k]);
            t2[NB2 + 1 + k] = dpp_shl1(t2[NB2 + k]);
        }
        float o2[2];
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const int gp = e + SHIFT;
            const int pl = gp >> 1, offx = 1 - (gp & 1);
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int j = 0; j < H2; j++) {
                const int k = HLEN - 1 - (2 * j + offx);
                s1 = __builtin_fmaf(t1[NB2 + pl - C + j], f.a[k], s1);
                s2 = __builtin_fmaf(t2[NB2 + pl - C + j], f.b[k], s2);
            }
            o2[e] = s1 + s2;
        }
        a0 = o2[0];
        a1 = o2[1];
    };

    constexpr int kWait2 = VM_PF - 4, kWait1 = VM_PF - 3;
    auto step = [&](auto Pp, int sb) {
        constexpr int p = decltype(Pp)::value;
        const int s = sb * H2 + p;
        // ---- level l+1: coefficient row r2 = H2-1+s completes the window of step s ----
        constexpr int pr = p % PFD;  // register slot of this step
        asm_wait4<kWait2>(q2[pr][0], q2[pr][1], q2[pr][2], q2[pr][3]);
        r2av[(H2 - 1 + p) % H2] = v2f{asm_copy(q2[pr][0]), asm_copy(q2[pr][2])};
        r2hd[(H2 - 1 + p) % H2] = v2f{asm_copy(q2[pr][1]), asm_copy(q2[pr][3])};
        {
            const size_t o = off2(min(H2 - 1 + s + PFD, H2 - 2 + nsteps));  // the row needed PFD steps ahead (clamped to the last one)
            asm_load_s(q2[pr][0], b.A2 + o, voff2);
            asm_load_s(q2[pr][1], b.H2 + o, voff2);
            asm_load_s(q2[pr][2], b.V2 + o, voff2);
            asm_load_s(q2[pr][3], b.D2 + o, voff2);
        }
        static_for<2>([&](auto I) {
            constexpr int idx = decltype(I)::value;  // 0: tap parity 1 (A_l row 2P-SHIFT), 1: parity 0 (the next row)
            constexpr int q = 2 * p + idx;           // position of the A_l row in the super-body
            float a0, a1;
            synth2(std::integral_constant<int, p % H2>{}, std::integral_constant<int, 1 - idx>{}, a0, a1);
            // ---- level l: stream row r1 enters the ring with its H,V,D row ----
            const int r1 = 2 * s + idx;
            constexpr int qr = q % (2 * PFD);  // register slot of this row
            asm_wait3<kWait1>(q1[qr][0], q1[qr][1], q1[qr][2]);
            ra[q % H2] = v2f{a0, a1};
            rh[q % H2] = asm_copy(q1[qr][0]);
            rv[q % H2] = asm_copy(q1[qr][1]);
            rd[q % H2] = asm_copy(q1[qr][2]);
            {
                const size_t o = off1(min(r1 + 2 * PFD, 2 * nsteps - 1));
                asm_load_s(q1[qr][0], b.H1 + o, voff1);
                asm_load_s(q1[qr][1], b.V1 + o, voff1);
                asm_load_s(q1[qr][2], b.D1 + o, voff1);
            }
            const int g1 = 2 * (r1 - (H2 - 1)) - SHIFT;
            emit(std::integral_constant<int, (q + 1) % H2>{}, std::integral_constant<int, 1>{}, g1);
            emit(std::integral_constant<int, (q + 1) % H2>{}, std::integral_constant<int, 0>{}, g1 + 1);
        });
    };

    static_for<PFD>([&](auto K) {
        constexpr int k = decltype(K)::value;
        asm_drain1(q2[k][0]);
        asm_drain1(q2[k][1]);
        asm_drain1(q2[k][2]);
        asm_drain1(q2[k][3]);
    });
    static_for<2 * PFD>([&](auto K) {
        constexpr int k = decltype(K)::value;
        asm_drain1(q1[k][0]);
        asm_drain1(q1[k][1]);
        asm_drain1(q1[k][2]);
    });
    for (int sb = 0;; sb++) {
        bool fin = false;
        static_for<H2>([&](auto Pp) {
            if (!fin) {
                step(Pp, sb);
                fin = (sb * H2 + decltype(Pp)::value + 1 >= nsteps);
            }
        });
        if (fin) break;
    }
    static_for<PFD>([&](auto K) {
        constexpr int k = decltype(K)::value;
        asm_drain1(q2[k][0]);
        asm_drain1(q2[k][1]);
        asm_drain1(q2[k][2]);
        asm_drain1(q2[k][3]);
    });
    static_for<2 * PFD>([&](auto K) {
        constexpr int k = decltype(K)::value;
        asm_drain1(q1[k][0]);
        asm_drain1(q1[k][1]);
        asm_drain1(q1[k][2]);
    });
}

// =================================================================================================
// host side
// =================================================================================================
#define PDWT_CHECK_LAUNCH() PDWT_HIP_TRY(hipGetLastError())

static bool casc_enabled() { return knob(KN_CASC) == 1; }

template <int HLEN>
static int launch_fwd_casc(const float* in, const CascBands& b, float* trash, int nr, int nc, const Taps2<float>& f2, const CascBatchF* d_tbl, int nimg)
{
    TapsLH f;
    for (int k = 0; k < PDWT_MAX_FILTER_WIDTH; k++) f.t[k] = v2f{f2.a[k], f2.b[k]};
    constexpr int MAXVL = CascGeom<HLEN>::MAXVL;
    const int strips = idiv_up(nc, MAXVL * 4);
    const int VL = idiv_up(nc / 4, strips);
    // default prefetch distance (row registers in flight); banks of more than 16 taps: 4 -- HLEN / 2 prologue rows plus a deeper prefetch
    // would pass the 6-bit vmcnt counter
    constexpr int NVD = HLEN > 16 ? 4 : (HLEN % 4 == 0 ? HLEN / 2 : HLEN);
    KTimer kt(K_FWD2D_CASC, true);
    // ---- workgroup form: W waves stacked in one strip hand their ring warm-up rows to the wave above through LDS, so only the
    // LAST wave of a workgroup re-reads and recomputes the 3(hlen-2) halo input rows (C2: 14 x 18 workgroups of 4 waves
    // stream 4096 + 14*18 rows per strip instead of 4096 + 56*18)
    int W = knob(KN_CASC_WG);
    if (W == 0) W = 16;  // C2 forward: 25.1 us independent waves, 24.6 W=8, 23.2 W=16 (252 workgroups each)
    if (W != 1 && W != 4 && W != 8 && W != 16) W = 16;
    constexpr size_t REG = casc_fwd_region_bytes<HLEN>();
    while (W > 1 && (size_t)(W - 1) * REG > 150 * 1024) W /= 2;  // the hand-off area must fit the 160 KiB of LDS
    if (W == 2) W = 1;
    if (W > 1) {
        const int nr4 = nr / 4;
        // one workgroup per CU; if the image is too short for W waves of >= HLEN/2 level-2 rows each at that count, fewer waves
        // per workgroup rather than fewer workgroups (2048^2 db4: 20.8 us with 72 workgroups of 16, 15.9 with 252 of 4)
        const int wgs = knob(KN_CASC_WAVES) > 0 ? idiv_up(knob(KN_CASC_WAVES), W) : 256;
        int gy = std::max(1, wgs / strips);
        while (W > 4 && (nr4 / gy) / W < HLEN / 2) W /= 2;
        while (gy > 1 && (nr4 / gy) / W < HLEN / 2) gy--;  // every wave needs >= HLEN/2 level-2 rows (see the kernel)
        if (gy >= 1 && (nr4 / gy) / W >= HLEN / 2) {
            int nwg = gy * strips;
            const size_t lds = (size_t)(W - 1) * REG;  // hand-off regions
            // (HLEN row registers in flight instead of HLEN/2 measured slower in this form too: 24.8 vs 24.6 us at W = 8)
            void (*k)(const float*, CascBands, int, int, int, float*, CascMap, TapsLH);
            k = (W == 4) ? k_fwd2d_casc<HLEN, NVD, 4> : (W == 8) ? k_fwd2d_casc<HLEN, NVD, 8> : k_fwd2d_casc<HLEN, NVD, 16>;
            // 16 waves: TWO row registers per wave (one A1 row of prefetch) -- the launch starts with 8 instead of 10 rows per wave in flight
            // (4032 waves: 33 instead of 41 MB of read-only start): C2 forward 23.5-23.8 -> 23.1-23.2 us (casc_nv = hlen/2: the round-2 depth)
            const bool nv2 = W == 16 && (knob(KN_CASC_NV) == 0 || knob(KN_CASC_NV) == 2);
            if (nv2) k = k_fwd2d_casc<HLEN, 2, 16>;
            // the straight-line wave programs (kernel form SPEC) exist for waves of 4 or 5 level-2 rows whose workgroup's last wave has 1:
            // the kernel's split, replayed on the two chunk sizes that occur (C2: 73 or 74 level-2 rows over 16 waves)
            bool spec = nv2 && (knob(KN_CASC_SPEC) & 1) && 2 * 4 >= HLEN;
            auto spec_ok = [&](int gyq, int xwq) {
                const int cpxq = idiv_up(gyq * strips, 8);
                for (int g = 0; g < gyq; g++)
                    for (int st = 0; st < strips; st++) {
                        const int R = casc_chunk_start(g + 1, st, nr4, gyq, strips, cpxq, xwq) - casc_chunk_start(g, st, nr4, gyq, strips, cpxq, xwq);
                        if (R < 16) return false;
                        const int E = std::min((3 * (HLEN - 2) + 3) / 4, R / 16 - 1);
                        const int base = (R + E) / 16, rem = (R + E) % 16;
                        if (base != 4 || R - (15 * base + std::min(15, rem)) != 1) return false;
                    }
                return true;
            };
            // Tall images (8192 rows and more: 18+ level-2 rows per wave with one workgroup per CU) run SEVERAL rounds of workgroups of the C2
            // height instead, which have wave programs -- the situation of a batch of 4096-row images, whose workgroups are out of phase
            // after the first round (tools/tall_sweep.py, db4 L3 pair, interleaved on one box: 8192^2 268 -> 255 us, 12288^2 592 -> 546, 16384^2
            // 1135 -> 1031; 6144^2, 2.3 rounds: 135 -> 137, hence three rounds and more only).  knob casc_spec bit 2.
            // Forward only: the inverse's per-workgroup prologue (three ring warm-ups) makes the same trade a loss (8192^2 pair +27 us).
            if (spec && (knob(KN_CASC_SPEC) & 4) && knob(KN_CASC_WAVES) <= 0 && !d_tbl && (nr4 / gy) / 16 >= 8) {
                const int gy2 = (nr4 + 36) / 73;
                if (gy2 >= 3 * gy && spec_ok(gy2, 0)) {
                    gy = gy2;
                    nwg = gy * strips;
                }
            }
            // XCD-weighted split (casc_chunk_start): only with the wave programs, and only if every workgroup still has them
            int xw = spec ? knob(KN_CASC_XCDW) : 0;
            if (spec && xw != 0 && !spec_ok(gy, xw)) xw = 0;  // the weighted split leaves the instantiated row counts: even split
            spec = spec && spec_ok(gy, xw);
            if (!spec) xw = 0;
            if (spec) {
                k = k_fwd2d_casc<HLEN, 2, 16, true>;
                stat_casc_spec(0);
            }
            const CascMap cm = {idiv_up(nwg, 8), strips, gy, xw, d_tbl};
            const dim3 grid((unsigned)(8 * cm.cpx), (unsigned)(d_tbl ? nimg : 1));
            if (lds > 64 * 1024) {  // opt-in once per (kernel, device), not per launch
                int rc = (W == 4) ? lds_opt_in<k_fwd2d_casc<HLEN, NVD, 4>>()
                         : (W == 8) ? lds_opt_in<k_fwd2d_casc<HLEN, NVD, 8>>() : lds_opt_in<k_fwd2d_casc<HLEN, NVD, 16>>();
                if (rc == PDWT_OK && nv2) rc = lds_opt_in<k_fwd2d_casc<HLEN, 2, 16>>();
                if (rc == PDWT_OK && spec) rc = lds_opt_in<k_fwd2d_casc<HLEN, 2, 16, true>>();
                if (rc != PDWT_OK) return rc;
            }
            PDWT_LAUNCH_KT(kt, k, grid, dim3(64 * W), lds, in, b, nr, nc, VL, trash, cm, f);
            PDWT_CHECK_LAUNCH();
            return PDWT_OK;
        }
    }
    if (d_tbl) return 1;  // (the independent-wave kernels have no batched form)
    // ---- independent waves.  chunk rows: a multiple of 8 (one band per XCD), ~PDWT_CASC_WAVES waves in total, at least 4
    // level-2 rows each.  One wave per SIMD (1024) is the optimum: every extra chunk row recomputes 3(hlen-2) input rows of halo
    // (measured 27.2 us @1024, 30.9 @2048, 35 @4096 for 4096^2 db4).
    int cpx = (knob(KN_CASC_WAVES) > 0 ? knob(KN_CASC_WAVES) : 1024) / (8 * strips);
    if (cpx > nr / 4 / 4 / 8) cpx = nr / 4 / 4 / 8;
    if (cpx < 1) cpx = 1;
    const CascMap cm = {cpx, strips, 0};
    const dim3 grid((unsigned)(8 * idiv_up(cpx * strips, 4)));
    // row registers in flight (= prefetch distance): HLEN/2 measured best (26.2 us vs 26.8 @HLEN, 28.5 @2*HLEN for 4096^2 db4);
    // a shorter pipeline fills and drains faster, and every wave fills and drains at the same time
    const int nv = knob(KN_CASC_NV) > 0 ? knob(KN_CASC_NV) : NVD;
    if (nv == 2)
        PDWT_LAUNCH_KT(kt, (k_fwd2d_casc<HLEN, 2, 1>), grid, dim3(256), 0, in, b, nr, nc, VL, trash, cm, f);
    else if (nv < HLEN && HLEN % 4 == 0)
        PDWT_LAUNCH_KT(kt, (k_fwd2d_casc<HLEN, NVD, 1>), grid, dim3(256), 0, in, b, nr, nc, VL, trash, cm, f);
    else
        PDWT_LAUNCH_KT(kt, (k_fwd2d_casc<HLEN, (HLEN > 16 ? NVD : HLEN), 1>), grid, dim3(256), 0, in, b, nr, nc, VL, trash, cm, f);
    PDWT_CHECK_LAUNCH();
    return PDWT_OK;
}

#ifdef PDWT_CASC_ONLY8  // (quick ISA inspection builds)
#define PDWT_CASC_FWD_HLENS(X) X(8)
#else
#define PDWT_CASC_FWD_HLENS(X) X(4) X(6) X(8) X(10) X(12) X(14) X(16) X(18) X(20)
#endif

int fwd2d_casc_f32(const float* in, float* H1, float* V1, float* D1, float* A2, float* H2, float* V2, float* D2, float* trash, int nr,
                   int nc, int hlen, const Taps2<float>& f, const CascBatchF* d_tbl, int nimg)
{
    if (!casc_enabled() || !stream_enabled() || !trash) return 1;
    if ((nr & 3) || (nc & 3) || nc < 256 || nr < 16 * hlen) return 1;
    if ((long long)nr * nc < (long long)knob(KN_CASC_MIN)) return 1;
    if ((long long)nr * nc >= (1LL << 32)) return 1;  // 32-bit row offsets inside a level-1 band (nr/2 x nc/2 floats): the level kernels take larger images
    if (!al16(in) || !al16(H1) || !al16(V1) || !al16(D1) || !al16(A2) || !al16(H2) || !al16(V2) || !al16(D2)) return 1;
    const CascBands b = {H1, V1, D1, A2, H2, V2, D2};
    switch (hlen) {
#define X(H) \
    case H: return launch_fwd_casc<H>(in, b, trash, nr, nc, f, d_tbl, nimg);
        PDWT_CASC_FWD_HLENS(X)
#undef X
        default: return 1;
    }
}

template <int HLEN>
static int launch_inv_casc(const CascInvBands& b, float* out, float* trash, int nr, int nc, const Taps2<float>& f)
{
    constexpr int MAXVL = CascInvGeom<HLEN>::MAXVL;
    const int nc1 = nc / 2;
    const int strips = idiv_up(nc1, MAXVL * 2);
    const int VL = idiv_up(nc1 / 2, strips);
    int cpx = (knob(KN_CASC_IWAVES) > 0 ? knob(KN_CASC_IWAVES) : 2048) / (8 * strips);
    if (cpx > nr / 2 / 8 / 8) cpx = nr / 2 / 8 / 8;  // at least 8 level-l coefficient rows per chunk
    if (cpx < 1) cpx = 1;
    const CascMap cm = {cpx, strips, 0};
    const dim3 grid((unsigned)(8 * idiv_up(cpx * strips, 4)));
    KTimer kt(K_INV2D_CASC, true);
    constexpr int H2 = HLEN / 2;
    // prefetch distance in steps: 1 measured best (28.2 us vs 29.5 @2, 31.5 @H2 for 4096^2 db4, 2048 waves)
    const int pfd = knob(KN_CASC_IPFD);
    bool launched = false;
    if constexpr (H2 % 2 == 0) {
        if (pfd == 2) {
            PDWT_LAUNCH_KT(kt, (k_inv2d_casc<HLEN, 2>), grid, dim3(256), 0, b, out, nr, nc, VL, trash, cm, f);
            launched = true;
        }
    }
    if constexpr (H2 * 14 - 3 <= 63 && H2 > 2) {  // a whole ring period ahead only while vmcnt can count that far
        if (pfd >= H2 && !launched) {
            PDWT_LAUNCH_KT(kt, (k_inv2d_casc<HLEN, H2>), grid, dim3(256), 0, b, out, nr, nc, VL, trash, cm, f);
            launched = true;
        }
    }
    if (!launched) PDWT_LAUNCH_KT(kt, (k_inv2d_casc<HLEN, 1>), grid, dim3(256), 0, b, out, nr, nc, VL, trash, cm, f);
    PDWT_CHECK_LAUNCH();
    return PDWT_OK;
}

// (hlen >= 12: two single-level launches are faster in this direction -- 37.9 vs 41.3 us db6, 52.4 vs 59.6 us db8 at 4096^2)
#ifdef PDWT_CASC_ONLY8
#define PDWT_CASC_INV_HLENS(X) X(8)
#else
#define PDWT_CASC_INV_HLENS(X) X(4) X(6) X(8) X(10)
#endif

int inv2d_casc_f32(const float* A2, const float* H2, const float* V2, const float* D2, const float* H1, const float* V1, const float* D1,
                   float* out, float* trash, int nr, int nc, int hlen, const Taps2<float>& f)
{
    if (!casc_enabled() || !stream_enabled() || !trash) return 1;
    if ((nr & 3) || (nc & 3) || nc < 256 || nr < 32 * hlen) return 1;
    if ((long long)nr * nc < (long long)knob(KN_CASC_MIN)) return 1;
    if (!al16(out) || !al16(H1) || !al16(V1) || !al16(D1) || !al16(A2) || !al16(H2) || !al16(V2) || !al16(D2) || !al16(trash)) return 1;
    const CascInvBands b = {A2, H2, V2, D2, H1, V1, D1};
    switch (hlen) {
#define X(H) \
    case H: return launch_inv_casc<H>(b, out, trash, nr, nc, f);
        PDWT_CASC_INV_HLENS(X)
#undef X
        default: return 1;
    }
}

}  // namespace pdwt
