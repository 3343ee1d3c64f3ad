// dwt_casc_inv3.hip -- inverse 2D DWT, THREE levels per launch, float32, workgroup form, all three levels STREAMED.
//
// k_inv2d_cascw<.., L3 = true> (dwt_casc_invw.hip) folds the third level in with a prologue: every wave synthesises all the
// A_{l+1} rows it owns from the level-(l+2) bands into private LDS rows before its first step.  In-kernel timestamps
// (tools/casc_trace.py, profiles/r03_c2_timeline.md) show what that costs at C2: 6.2-6.8 us pass before ANY wave emits its first
// row (63 loads per lane, ~630 VALU, an LDS round trip, every wave in the same phase) in a kernel whose steady-state loop
// moves 7.6 TB/s -- a quarter of the launch is dead time.  Here the third level is a third ring:
//
//   level l+2 ring (H2 rows of (A,V) and (H,D), lane = ONE level-(l+2) column)  -- one new row every SECOND step
//     -> column synthesis, DPP halo, row synthesis: the lane's two A_{l+1} columns (same arithmetic as level l+1 -> A_l)
//     -> two ds_bpermute + a select move them to the main mapping (lane = one level-(l+1) column)
//   level l+1 ring  -- the A part of the new row is that value, V / H / D come from memory      -- one new row per step
//   level l   ring  -- as in k_inv2d_cascw                                                       -- two new rows per step
//
// so the prologue shrinks to the three ring warm-ups (one round trip of loads, three small syntheses), no private LDS rows
// exist (W = 16 waves fit: 138 KB of hand-off regions) and the A_{l+1} rows a wave gets from the wave below arrive through the
// existing hand-off (they carry their A part).  A_{l+1} rows come in pairs from one level-(l+2) window (tap parity 1, then 0);
// chunks start at EVEN level-(l+1) rows, so the pair phase is a compile-time constant and a step's parity decides whether the
// level-(l+2) ring advances (even steps) -- H2 must be even for the unrolled super-body to see a constant parity: db2 and
// db4 / sym4 (hlen 4 and 8); other lengths keep the prologue form.
// Results are bit-identical to the per-level kernels (same taps, same FMA order per output).
// Reference code replaced: three iterations of the level loop of w_inverse_separable (src/separable.cu:332-364), two kernels
// each (:246-328).
#include <algorithm>

#include "casc_dev.hpp"
#include "dwt_stream.hpp"
#include "stream_dev.hpp"

namespace pdwt {

// lambdas of the kernel: inlined wherever they are called (a call in front of s_endpgm counts as cold and would stay a CALL, with the wave's
// whole state passed through scratch)
#define PDWT_AI __attribute__((always_inline))

template <int HLEN>
constexpr int casc_inv3_region_bytes() { return (HLEN / 2 - 1) * 64 * (16 + 32); }

// Column syntheses of a 4-tap half bank (hlen 8) as ONE asm statement each.  As separate statements (pk_fma_sbcast, one per tap) hipcc puts a
// wait state between any two that depend on each other -- it cannot see what is inside -- ~100 s_nop per wave of ~1900 instructions in
// a kernel that is bound by instruction issue.  HALF: which half of the aligned SGPR tap pairs (F[2i], F[2i+1]) is broadcast (tap parity);
// window position j meets pair 3 - j.  Four independent accumulators interleaved; every scalar still sums its taps in ascending j.
#define PDWT_PKS_FIRST0 " op_sel_hi:[1,0,0]"
#define PDWT_PKS_NEXT0 " op_sel_hi:[1,0,1]"
#define PDWT_PKS_FIRST1 " op_sel:[0,1,0] op_sel_hi:[1,1,0]"
#define PDWT_PKS_NEXT1 " op_sel:[0,1,0] op_sel_hi:[1,1,1]"
template <int HALF>
__device__ __forceinline__ void col_synth4x4(v2f& sa, v2f& sh, v2f& sv, v2f& sd, v2f a0, v2f a1, v2f a2, v2f a3, v2f h0, v2f h1, v2f h2, v2f h3, v2f v0,
                                             v2f v1, v2f v2, v2f v3, v2f d0, v2f d1, v2f d2, v2f d3, v2f fa3, v2f fa2_, v2f fa1, v2f fa0, v2f fb3, v2f fb2_,
                                             v2f fb1, v2f fb0)
{
#define PDWT_CS4(F, N)                                                                                                                    \
    asm("v_pk_fma_f32 %0, %4, %20, 0" F "\n\tv_pk_fma_f32 %1, %8, %24, 0" F "\n\tv_pk_fma_f32 %2, %12, %20, 0" F "\n\tv_pk_fma_f32 %3, %16, %24, 0" F \
        "\n\tv_pk_fma_f32 %0, %5, %21, %0" N "\n\tv_pk_fma_f32 %1, %9, %25, %1" N "\n\tv_pk_fma_f32 %2, %13, %21, %2" N "\n\tv_pk_fma_f32 %3, %17, %25, %3" N \
        "\n\tv_pk_fma_f32 %0, %6, %22, %0" N "\n\tv_pk_fma_f32 %1, %10, %26, %1" N "\n\tv_pk_fma_f32 %2, %14, %22, %2" N "\n\tv_pk_fma_f32 %3, %18, %26, %3" N \
        "\n\tv_pk_fma_f32 %0, %7, %23, %0" N "\n\tv_pk_fma_f32 %1, %11, %27, %1" N "\n\tv_pk_fma_f32 %2, %15, %23, %2" N "\n\tv_pk_fma_f32 %3, %19, %27, %3" N \
        : "=&v"(sa), "=&v"(sh), "=&v"(sv), "=&v"(sd)                                                                                       \
        : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(h0), "v"(h1), "v"(h2), "v"(h3), "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(d0), "v"(d1), "v"(d2),  \
          "v"(d3), "s"(fa3), "s"(fa2_), "s"(fa1), "s"(fa0), "s"(fb3), "s"(fb2_), "s"(fb1), "s"(fb0))
    if constexpr (HALF == 0) PDWT_CS4(PDWT_PKS_FIRST0, PDWT_PKS_NEXT0);
    else PDWT_CS4(PDWT_PKS_FIRST1, PDWT_PKS_NEXT1);
#undef PDWT_CS4
}
template <int HALF>
__device__ __forceinline__ void col_synth2x4(v2f& sav, v2f& shd, v2f a0, v2f a1, v2f a2, v2f a3, v2f h0, v2f h1, v2f h2, v2f h3, v2f fa3, v2f fa2_, v2f fa1,
                                             v2f fa0, v2f fb3, v2f fb2_, v2f fb1, v2f fb0)
{
#define PDWT_CS2(F, N)                                                                                                      \
    asm("v_pk_fma_f32 %0, %2, %10, 0" F "\n\tv_pk_fma_f32 %1, %6, %14, 0" F "\n\tv_pk_fma_f32 %0, %3, %11, %0" N "\n\tv_pk_fma_f32 %1, %7, %15, %1" N \
        "\n\tv_pk_fma_f32 %0, %4, %12, %0" N "\n\tv_pk_fma_f32 %1, %8, %16, %1" N "\n\tv_pk_fma_f32 %0, %5, %13, %0" N "\n\tv_pk_fma_f32 %1, %9, %17, %1" N \
        : "=&v"(sav), "=&v"(shd)                                                                                            \
        : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(h0), "v"(h1), "v"(h2), "v"(h3), "s"(fa3), "s"(fa2_), "s"(fa1), "s"(fa0), "s"(fb3), "s"(fb2_),  \
          "s"(fb1), "s"(fb0))
    if constexpr (HALF == 0) PDWT_CS2(PDWT_PKS_FIRST0, PDWT_PKS_NEXT0);
    else PDWT_CS2(PDWT_PKS_FIRST1, PDWT_PKS_NEXT1);
#undef PDWT_CS2
}

// Bookkeeping of a straight-line wave program (see the kernel): NQ level-(l+1) rows, last wave of its workgroup or not.  Everything is a
// function of the step index s: which parts of a step run, which loads it issues, how many VMEM instructions lie between a load
// and its use.  Order of the VMEM instructions of a step (the loop's order): [level l+2: wait, 4 loads] [level l+1: wait, NL2 loads]
// then for each of the two A_l rows: [wait, 3 loads] [up to 2 stores].
template <int HLEN, int NQ, bool LV, bool L3>
struct CascInvSpec {
    using G = CascInvGeom<HLEN>;
    static constexpr int H2 = G::H2, XS = H2 / 2, NSTEPS = NQ + XS, NP = 2 * NQ;
    static constexpr int NL2 = L3 ? 3 : 4;
    static constexpr bool l2act(int s) { return s >= 0 && s < NSTEPS && (LV || s < NQ); }        // the level-(l+1) part runs
    static constexpr bool need2(int s) { return l2act(s) && (LV || s + H2 - 1 < NQ); }          // its new row is the wave's own (not from the hand-off)
    static constexpr bool adv3(int s) { return L3 && !(s & 1) && (need2(s) || need2(s + 1)); }  // the level-(l+2) ring advances
    static constexpr int n3(int s) { return (adv3(s) && adv3(s + 2)) ? 4 : 0;  }                // loads a step issues ...
    static constexpr int n2(int s) { return (l2act(s) && need2(s + 1)) ? NL2 : 0; }
    static constexpr int n1(int s) { return (l2act(s) && l2act(s + 1)) ? 3 : 0; }               // ... per A_l row
    static constexpr bool own(int s, int idx) { return 2 * s + idx - (H2 - 1) >= 0 && 2 * s + idx - (H2 - 1) < NP; }
    static constexpr int st(int s, int idx) { return own(s, idx) ? 2 : 0; }
    static constexpr int total(int s) { return n3(s) + n2(s) + 2 * n1(s) + st(s, 0) + st(s, 1); }
    // VMEM instructions issued after the load of a row register and before its use
    static constexpr int wait3(int s) { return n2(s - 2) + 2 * n1(s - 2) + st(s - 2, 0) + st(s - 2, 1) + total(s - 1); }
    static constexpr int wait2(int s) { return 2 * n1(s - 1) + st(s - 1, 0) + st(s - 1, 1) + n3(s); }
    static constexpr int wait1(int s, int idx)
    {
        return idx == 0 ? st(s - 1, 0) + n1(s - 1) + st(s - 1, 1) + n3(s) + n2(s) : st(s - 1, 1) + n3(s) + n2(s) + n1(s) + st(s, 0);
    }
};

// L3 = false: the same kernel on TWO levels (the A parts of the level-(l+1) rows are loaded like their H, V, D parts; no third ring)
// SPEC = true: the kernel consists of the straight-line wave programs only (the host launches it when every wave of the geometry has one)
template <int HLEN, int W, bool L3, bool SPEC = false>
__global__ __launch_bounds__(64 * W) void k_inv2d_casc3(CascInvBands b, CascInv3B b3, float* out, int Nr, int Nc, int VL,
                                                         float* __restrict__ trash, CascMap cm, Taps2<float> f)
{
    using G = CascInvGeom<HLEN>;
    constexpr int H2 = G::H2, C = G::C, SHIFT = G::SHIFT, NBT = G::NBT;
    static_assert(H2 % 2 == 0, "the parity of a step must be a compile-time constant of the unrolled super-body");
    constexpr int XS = H2 / 2;   // extra (level-l only) steps that drain the last H2-1 level-l windows
    constexpr int PHI = SHIFT;   // chunks start at even level-(l+1) rows: stream row s2 is output (s2 + PHI) & 1 of level-(l+2) step (s2 + PHI) >> 1
    constexpr int T0 = (H2 - 1 + PHI) >> 1;  // level-(l+2) step of loop step 0 (whose new row s2 = H2-1 is the FIRST of a pair)
    static_assert(((H2 - 1 + PHI) & 1) == 0, "loop step 0 starts a level-(l+2) window");
    constexpr int NW3 = H2 + T0 - 1;         // level-(l+2) rows the warm-up windows t = 0 .. T0-1 span
    // VMEM instructions between a load and its use (stream_dev.hpp); E = the step is even (4 level-(l+2) loads are issued first)
    constexpr int NL2 = L3 ? 3 : 4;            // level-(l+1) loads per step
    constexpr int kStep = NL2 + 2 * (3 + 2);   // (odd) step: the level-(l+1) loads + per A_l row 3 loads and 2 stores
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    CASC_TRACE_DECL;
    CASC_TRACE(0);
    if (cm.tbl) {  // batched launch: this workgroup's image -- read through the constant address space: scalar loads, pointers in SGPRs
        static_assert(sizeof(CascBatchI) == 12 * sizeof(void*), "twelve pointers per image");
        typedef const unsigned long long __attribute__((address_space(4))) * tbl_t;
        const tbl_t q = (tbl_t)(const unsigned long long*)cm.tbl + 12 * (size_t)blockIdx.y;
        b = CascInvBands{(const float*)q[0], (const float*)q[1], (const float*)q[2], (const float*)q[3], (const float*)q[4], (const float*)q[5], (const float*)q[6]};
        b3 = CascInv3B{(const float*)q[7], (const float*)q[8], (const float*)q[9], (const float*)q[10]};
        out = (float*)q[11];
    }
    const int lane = threadIdx.x & 63;
    const int kw = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int Nr1 = Nr >> 1, Nc1 = Nc >> 1, Nr2 = Nr >> 2, Nc2 = Nc >> 2, Nr3 = Nr >> 3, Nc3 = Nc >> 3;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int wg = xcd * cm.cpx + slot;
    if (slot >= cm.cpx || wg >= cm.gy * cm.strips) return;  // (uniform over the workgroup: nobody is left at a barrier)
    const int gy = wg / cm.strips, strip = wg % cm.strips;
    // rows are dealt in PAIRS (see PHI): workgroup chunk [J0, J0 + R), wave chunk [Q0, Q0 + nQ), all even
    const int Np = Nr2 >> 1;
    const int Jp = casc_chunk_start(gy, strip, Np, cm.gy, cm.strips, cm.cpx, cm.flags);
    const int Rp = casc_chunk_start(gy + 1, strip, Np, cm.gy, cm.strips, cm.cpx, cm.flags) - Jp;
    // every wave runs nQ + XS steps (the last XS of the last wave are full steps: it recomputes its halo; the others only run the
    // level-l part on rows from the hand-off), so an even split of the pairs is the balanced one (timeline: r03_c2_timeline.md)
    const int Ep = 0;
    const int basep = (Rp + Ep) / W, remp = (Rp + Ep) % W;
    const int startp = kw * basep + min(kw, remp);
    const bool last = (kw == W - 1);
    const int nQ = 2 * (last ? Rp - startp : basep + (kw < remp ? 1 : 0));
    const int Q0 = 2 * (Jp + startp);
    const int nsteps = nQ + XS;
    const int P0 = 2 * (Q0 + C) - SHIFT;  // A_l row of stream index r1 = 0
    const int O0 = 2 * (P0 + C) - SHIFT;  // output row of stream index g = 0
    const int nP = 2 * nQ;                // level-l windows (= output row pairs) the wave owns
    const int last2 = last ? H2 - 2 + nsteps : nQ - 1;  // last stream rows the wave loads (or synthesises) itself
    const int last1 = last ? 2 * nsteps - 1 : nP - 1;
    const int P3_0 = (Q0 + SHIFT) >> 1;                 // level-(l+2) pair index of level-(l+2) step t = 0
    const int last3 = ((last2 + PHI) >> 1) + H2 - 1;    // last level-(l+2) stream row (i = 0 <-> row P3_0 - C) the wave needs

    const int X0 = strip * VL - NBT;               // level-(l+1) column of lane 0 (may be negative: periodic)
    const int cx1 = 2 * (X0 + lane);               // first of the lane's two level-l coefficient columns
    const bool valid = (lane >= NBT) && (lane < NBT + VL) && (cx1 < Nc1);
    const int cx1w = wrapi(cx1, Nc1);
    const int cx2w = cx1w >> 1;
    // level l+2: lane <-> column c3 = ((X0 + 1) >> 1) - C + lane, which produces the natural pair (A[2 c3 - 1], A[2 c3]).  The lane
    // of the main mapping that holds A_{l+1} column X = X0 + lane takes it from level-(l+2) lane ((X + 1) >> 1) - ((X0 + 1) >> 1) + C:
    // the second output when X is even, the first when it is odd.
    const int c3w = wrapi(((X0 + 1) >> 1) - C + lane, Nc3);
    const int bp_addr = 4 * (((X0 + lane + 1) >> 1) - ((X0 + 1) >> 1) + C);
    const bool bp_odd = ((X0 + lane) & 1) != 0;

    auto off3 = [&](int i) PDWT_AI { return (size_t)CASC_DIAG_LD(wrapi(P3_0 - C + i, Nr3)) * Nc3; };
    auto off2 = [&](int s2) PDWT_AI { return (size_t)CASC_DIAG_LD(wrap1(Q0 + s2, Nr2)) * Nc2; };
    auto off1 = [&](int r1) PDWT_AI { return (size_t)CASC_DIAG_LD(wrap1(P0 + r1, Nr1)) * Nc1; };
    const unsigned voff3 = (unsigned)c3w * 4u, voff2 = (unsigned)cx2w * 4u, voff1 = (unsigned)cx1w * 4u, voffo = (unsigned)(valid ? cx1 : 0) * 8u;
    const lanemask_t vmask = __ballot(valid);
    // Scalar bookkeeping of the loop (cf. k_fwd2d_casc): every stream of rows is walked by a cursor -- a 32-bit byte offset inside its band
    // (bands of 4 GiB and more never get here), folded into the lane offset by one v_add per row group, advanced by the band's row stride,
    // frozen at the last row the wave needs, wrapped on the offset itself -- instead of wrap + 64-bit multiply + 64-bit add per band and row;
    // the output rows are walked by a 64-bit pointer, and rows the wave does not own are stored with EXEC = 0 (no trash select).
    constexpr int kFold = (PDWT_CASC_DIAG & 2) ? 32 : 0;  // (diagnostic builds: loads folded onto 32 rows)
    const unsigned str3 = (unsigned)Nc3 * 4u, str2 = (unsigned)Nc2 * 4u, str1 = (unsigned)Nc1 * 4u;
    const unsigned end3 = (unsigned)(kFold ? min(kFold, Nr3) : Nr3) * str3, end2 = (unsigned)(kFold ? min(kFold, Nr2) : Nr2) * str2,
                   end1 = (unsigned)(kFold ? min(kFold, Nr1) : Nr1) * str1;
    int c3 = min(T0 + H2, last3), c2 = min(H2, last2), c1 = min(2, last1);  // stream rows of the next loads
    unsigned o3 = (unsigned)CASC_DIAG_LD(wrapi(P3_0 - C + c3, Nr3)) * str3, o2 = (unsigned)CASC_DIAG_LD(wrap1(Q0 + c2, Nr2)) * str2,
             o1 = (unsigned)CASC_DIAG_LD(wrap1(P0 + c1, Nr1)) * str1;
    auto advance = [&](int& c, unsigned& o, int lastc, unsigned str, unsigned end) {
        int cq = c;
        asm("" : "+s"(cq));  // (a scalar compare of its own, cf. k_fwd2d_casc)
        if (cq < lastc) {
            c++;
            o += str;
            if (o == end) o = 0;
        }
    };
    int orow = CASC_DIAG_ST(wrap1(O0, Nr));  // output row of the next OWN emit (own windows are consecutive: g = 0, 1, 2, ...)
    float* op = out + (size_t)orow * Nc;
    const int oend = (PDWT_CASC_DIAG & 1) ? min(32, Nr) : Nr;
    (void)trash;

    // LDS: the hand-off regions
    constexpr int REG = casc_inv3_region_bytes<HLEN>();
    unsigned char* const lds_rd = lds_raw + (size_t)kw * REG;                    // written by wave kw+1
    unsigned char* const lds_wr = lds_raw + (size_t)(kw > 0 ? kw - 1 : 0) * REG;  // read by wave kw-1
    auto lds_l2 = [&](unsigned char* reg, int r) PDWT_AI { return reinterpret_cast<v4f*>(reg + ((size_t)r * 64 + lane) * 16); };
    auto lds_l1 = [&](unsigned char* reg, int r, int h) PDWT_AI {
        return reinterpret_cast<v4f*>(reg + (size_t)(H2 - 1) * 64 * 16 + (((size_t)r * 2 + h) * 64 + lane) * 16);
    };

    v2f r3av[H2], r3hd[H2];               // level l+2 ring, oldest row in slot 0 (rotated by moves: it advances every second step)
    v2f r2av[H2], r2hd[H2];               // level l+1 ring: (A,V) and (H,D) of the lane's column
    v2f ra[H2], rh[H2], rv[H2], rd[H2];   // level l ring: the lane's two columns of each band
#pragma unroll
    for (int k = 0; k < H2; k++) ra[k] = rh[k] = rv[k] = rd[k] = v2f{0.f, 0.f};

    // Row synthesis on NATURAL pairs.  With SHIFT = 1 window position p (t columns p-C .. p-C+H2-1) yields the outputs 2p-1 and 2p,
    // and the two share every product's t operand: (o[2p-1], o[2p]) += splat(t[p-C+j]) * (F[m-1], F[m]), m = HLEN-1-2j -- one
    // v_pk_fma_f32 per tap with the taps as an aligned SGPR pair.  The kernels of dwt_casc_invw.hip compute the pairs a lane OWNS
    // (2c, 2c+1), which straddle two windows: scalar FMA chains that hipcc then re-packs with a v_mov per operand.  Here a lane
    // computes natural pairs only and the neighbour's half travels by one DPP move.  Same products, same order per output.
    static_assert(SHIFT == 1, "even H2");
    // the taps live once, as aligned SGPR pairs (F[2i], F[2i+1]): the row passes use a pair as it is, the column passes broadcast
    // one half of it (pk_fma_sbcast, stream_dev.hpp)
    v2f fa2[H2], fb2[H2];
#pragma unroll
    for (int i = 0; i < H2; i++) {
        fa2[i] = v2f{f.a[2 * i], f.a[2 * i + 1]};
        fb2[i] = v2f{f.b[2 * i], f.b[2 * i + 1]};
    }
    auto nat_pair = [&](const float* t1w, const float* t2w) PDWT_AI {  // t?w[j] = t[p-C+j]
        v2f s1 = {0.f, 0.f}, s2 = {0.f, 0.f};
#pragma unroll
        for (int j = 0; j < H2; j++) {
            const int m = HLEN - 1 - 2 * j;  // taps (F[m-1], F[m]) = pair (m-1)/2
            s1 = pk_fma(splat(t1w[j]), fa2[(m - 1) / 2], s1);
            s2 = pk_fma(splat(t2w[j]), fb2[(m - 1) / 2], s2);
        }
        return s1 + s2;
    };
    // one synthesis level on a lane that holds ONE coefficient column c: the natural pair (A[2c-1], A[2c]) of window p = c from the
    // ring window av / hd starting at slot S0 with tap parity OFF (column synthesis -> DPP halo of (t1, t2) -> row synthesis)
    auto synth_col = [&](const v2f (&av)[H2], const v2f (&hd)[H2], auto S0, auto OFF) {
        constexpr int s0 = decltype(S0)::value, off = decltype(OFF)::value;
        v2f sav = {0.f, 0.f}, shd = {0.f, 0.f};
        if constexpr (H2 == 4) {
            constexpr int half = (HLEN - 1 - off) & 1;  // (the parity of tap k = HLEN-1-(2j+off) does not depend on j)
            col_synth2x4<half>(sav, shd, av[s0 % 4], av[(s0 + 1) % 4], av[(s0 + 2) % 4], av[(s0 + 3) % 4], hd[s0 % 4], hd[(s0 + 1) % 4], hd[(s0 + 2) % 4],
                               hd[(s0 + 3) % 4], fa2[3], fa2[2], fa2[1], fa2[0], fb2[3], fb2[2], fb2[1], fb2[0]);
        } else {
            static_for<H2>([&](auto J) {
                constexpr int j = decltype(J)::value;
                constexpr int sl = (s0 + j) % H2;
                constexpr int k = HLEN - 1 - (2 * j + off);
                sav = pk_fma_sbcast<k & 1, j == 0>(av[sl], fa2[k >> 1], sav);
                shd = pk_fma_sbcast<k & 1, j == 0>(hd[sl], fb2[k >> 1], shd);
            });
        }
        const v2f t = sav + shd;  // (t1, t2) of the lane's column
        float t1w[H2], t2w[H2];   // window of p = c: columns c-C .. c-C+H2-1
        t1w[C] = t.x;
        t2w[C] = t.y;
#pragma unroll
        for (int k = 1; k <= C; k++) {
            t1w[C - k] = dpp_shr1(t1w[C - k + 1]);
            t2w[C - k] = dpp_shr1(t2w[C - k + 1]);
        }
#pragma unroll
        for (int k = 1; k <= H2 - 1 - C; k++) {
            t1w[C + k] = dpp_shl1(t1w[C + k - 1]);
            t2w[C + k] = dpp_shl1(t2w[C + k - 1]);
        }
        return nat_pair(t1w, t2w);
    };
    // the lane's two columns (2c, 2c+1) of the level below: its own second output and the first output of the lane to the right
    auto own_pair = [&](v2f p0) PDWT_AI { return v2f{p0.y, dpp_shl1(p0.x)}; };
    // one A_{l+1} value per lane of the main mapping from a level-(l+2) window: column X is the second output of level-(l+2)
    // column X/2 when X is even and the first output of column (X+1)/2 when it is odd
    auto a2_from = [&](const v2f (&av)[H2], const v2f (&hd)[H2], auto OFF) {
        const v2f p0 = synth_col(av, hd, std::integral_constant<int, 0>{}, OFF);
        const int glo = __builtin_amdgcn_ds_bpermute(bp_addr, __float_as_int(p0.x));
        const int ghi = __builtin_amdgcn_ds_bpermute(bp_addr, __float_as_int(p0.y));
        return __int_as_float(bp_odd ? glo : ghi);
    };

    // ---- ring warm-up rows of all three levels and the first row registers, issued together ----------------------------
    float q3[4];   // row registers in flight, level l+2 (the row the NEXT even step inserts)
    float q2[4];   // level l+1 (prefetch distance: one step); q2[0] is unused (the A part is synthesised)
    v2f q1[2][3];  // level l (two A_l rows per step)
    {
        v2f w3av[NW3], w3hd[NW3];
        if constexpr (L3) {
#pragma unroll
            for (int i = 0; i < NW3; i++) {
                const size_t o = off3(min(i, last3)) + c3w;
                w3av[i] = v2f{b3.A3[o], b3.V3[o]};
                w3hd[i] = v2f{b3.H3[o], b3.D3[o]};
            }
            const size_t o = off3(min(NW3, last3)) + c3w;
            q3[0] = b3.A3[o];
            q3[1] = b3.H3[o];
            q3[2] = b3.V3[o];
            q3[3] = b3.D3[o];
        } else {
            q3[0] = q3[1] = q3[2] = q3[3] = 0.f;
        }
#pragma unroll
        for (int r = 0; r < H2 - 1; r++) {
            const size_t o = off2(r) + cx2w;
            r2av[r] = v2f{L3 ? 0.f : b.A2[o], b.V2[o]};
            r2hd[r] = v2f{b.H2[o], b.D2[o]};
        }
        r2av[H2 - 1] = r2hd[H2 - 1] = v2f{0.f, 0.f};
        {
            const size_t o = off2(H2 - 1) + cx2w;
            q2[0] = L3 ? 0.f : b.A2[o];
            q2[1] = b.H2[o];
            q2[2] = b.V2[o];
            q2[3] = b.D2[o];
        }
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const size_t o = off1(q) + cx1w;
            q1[q][0] = *reinterpret_cast<const v2f*>(b.H1 + o);
            q1[q][1] = *reinterpret_cast<const v2f*>(b.V1 + o);
            q1[q][2] = *reinterpret_cast<const v2f*>(b.D1 + o);
        }
        // the A parts of the level-(l+1) warm-up rows s2 = 0 .. H2-2: output (s2 + PHI) & 1 of level-(l+2) window (s2 + PHI) >> 1
        if constexpr (L3) static_for<H2 - 1>([&](auto S2) {
            constexpr int s2 = decltype(S2)::value;
            constexpr int t = (s2 + PHI) >> 1, idx3 = (s2 + PHI) & 1;
            v2f av[H2], hd[H2];
#pragma unroll
            for (int j = 0; j < H2; j++) {
                av[j] = w3av[t + j];
                hd[j] = w3hd[t + j];
            }
            r2av[s2].x = a2_from(av, hd, std::integral_constant<int, 1 - idx3>{});
        });
        // the ring holds the window of step T0 - 1: loop step 0 rotates it and inserts q3
        if constexpr (L3) {
#pragma unroll
            for (int j = 0; j < H2; j++) {
                r3av[j] = w3av[T0 - 1 + j];
                r3hd[j] = w3hd[T0 - 1 + j];
            }
        }
    }
    CASC_TRACE(1);  // warm-up syntheses done
    // hand over the level-(l+1) ring warm-up rows: the wave above completes its last H2-1 windows with them
    if (kw > 0) {
#pragma unroll
        for (int r = 0; r < H2 - 1; r++) *lds_l2(lds_wr, r) = v4f{r2av[r].x, r2av[r].y, r2hd[r].x, r2hd[r].y};
    }


    // one output row of level l from the ring window starting at slot S0 with tap parity OFF: the lane holds the coefficient columns
    // (c0, c0+1) and owns the outputs 2 c0 .. 2 c0 + 3 = second output of window p = c0 (computed by the lane to the LEFT as the
    // second half of its last pair), the natural pair of p = c0+1, first output of p = c0+2
    auto emit = [&](auto S0, auto OFF, auto own_, int g) PDWT_AI {
        constexpr int s0 = decltype(S0)::value, off = decltype(OFF)::value;
        constexpr bool kOwnStatic = std::is_same<decltype(own_), std::true_type>::value;  // (wave programs: the row is known to be owned)
        const bool own = own_;
        // rows of the ring warm-up / beyond the wave's windows: the store is still issued (to a trash row, the VMEM count must
        // not change) but the arithmetic is skipped -- a uniform branch
        v4f o4;
        asm("" : "=v"(o4));  // (a row that goes to the trash carries whatever these registers hold: no instruction spent on it)
        if (own) {
            v2f sa = {0.f, 0.f}, sh = {0.f, 0.f}, sv = {0.f, 0.f}, sd = {0.f, 0.f};
            if constexpr (H2 == 4) {
                constexpr int half = (HLEN - 1 - off) & 1;
                constexpr int j0 = s0 % 4, j1 = (s0 + 1) % 4, j2 = (s0 + 2) % 4, j3 = (s0 + 3) % 4;
                col_synth4x4<half>(sa, sh, sv, sd, ra[j0], ra[j1], ra[j2], ra[j3], rh[j0], rh[j1], rh[j2], rh[j3], rv[j0], rv[j1], rv[j2], rv[j3], rd[j0],
                                   rd[j1], rd[j2], rd[j3], fa2[3], fa2[2], fa2[1], fa2[0], fb2[3], fb2[2], fb2[1], fb2[0]);
            } else {
                static_for<H2>([&](auto J) {
                    constexpr int j = decltype(J)::value;
                    constexpr int sl = (s0 + j) % H2;
                    constexpr int k = HLEN - 1 - (2 * j + off);
                    sa = pk_fma_sbcast<k & 1, j == 0>(ra[sl], fa2[k >> 1], sa);
                    sh = pk_fma_sbcast<k & 1, j == 0>(rh[sl], fb2[k >> 1], sh);
                    sv = pk_fma_sbcast<k & 1, j == 0>(rv[sl], fa2[k >> 1], sv);
                    sd = pk_fma_sbcast<k & 1, j == 0>(rd[sl], fb2[k >> 1], sd);
                });
            }
            const v2f t1o = sa + sh, t2o = sv + sd;
            // t columns c0 + 1 - C .. c0 + 2 - C + H2 - 1 (both windows), index 0 = column c0 + 1 - C; the lane's own two sit at C-1, C
            constexpr int NT = H2 + 1;
            float t1w[NT], t2w[NT];
            t1w[C - 1] = t1o.x;
            t1w[C] = t1o.y;
            t2w[C - 1] = t2o.x;
            t2w[C] = t2o.y;
#pragma unroll
            for (int i = C - 2; i >= 0; i--) {  // from the lane to the left (two columns per lane)
                t1w[i] = dpp_shr1(t1w[i + 2]);
                t2w[i] = dpp_shr1(t2w[i + 2]);
            }
#pragma unroll
            for (int i = C + 1; i < NT; i++) {  // from the lane to the right
                t1w[i] = dpp_shl1(t1w[i - 2]);
                t2w[i] = dpp_shl1(t2w[i - 2]);
            }
            const v2f pa = nat_pair(t1w, t2w);          // outputs 2 c0 + 1, 2 c0 + 2
            const v2f pb = nat_pair(t1w + 1, t2w + 1);  // outputs 2 c0 + 3, 2 c0 + 4 (the second belongs to the lane to the right)
            o4 = v4f{dpp_shr1(pb.y), pa.x, pa.y, pb.x};
        }
        (void)g;
        if constexpr (kOwnStatic) {
            asm_store_sm(op, voffo, o4, vmask);
            op += Nc;  // (the one place where a wave program's rows wrap is handled by the caller)
        } else {
            asm_store_sm(op, voffo, o4, own ? vmask : 0ull);
            if (own) {
                op += Nc;
                if (++orow == oend) {
                    orow = 0;
                    op = out;
                }
            }
        }
    };

    auto step = [&](auto Pp, int sb) {
        constexpr int p = decltype(Pp)::value;
        constexpr bool even = L3 && (p & 1) == 0;  // (H2 is even: the parity of the step is the parity of p)
        constexpr int EX = even ? 4 : 0;           // VMEM instructions an even step issues before everything else
        const int s = sb * H2 + p;
        const bool l2act = last || (s < nQ);  // the level-(l+1) part runs (afterwards: level-l rows from the hand-off only)
        if (l2act) {
            const int s2 = H2 - 1 + s;  // the level-(l+1) stream row that completes the window starting at stream row s
            // ---- level l+2: every second step the ring advances by the row that was loaded two steps ago ----
            if constexpr (even) {
                asm_wait4<2 * kStep>(q3[0], q3[1], q3[2], q3[3]);
#pragma unroll
                for (int j = 0; j < H2 - 1; j++) {
                    r3av[j] = r3av[j + 1];
                    r3hd[j] = r3hd[j + 1];
                }
                r3av[H2 - 1] = v2f{asm_copy(q3[0]), asm_copy(q3[2])};
                r3hd[H2 - 1] = v2f{asm_copy(q3[1]), asm_copy(q3[3])};
                {
                    // the row the even step after this one inserts: stream row T0 + s/2 + H2 (clamped to the last one needed)
                    const unsigned vo = voff3 + o3;
                    asm_load_s(q3[0], b3.A3, vo);
                    asm_load_s(q3[1], b3.H3, vo);
                    asm_load_s(q3[2], b3.V3, vo);
                    asm_load_s(q3[3], b3.D3, vo);
                    advance(c3, o3, last3, str3, end3);
                }
            }
            // the A part of stream row s2: first (even step: tap parity 1) or second output of the current level-(l+2) window
            float a2 = 0.f;
            if constexpr (L3) a2 = a2_from(r3av, r3hd, std::integral_constant<int, even ? 1 : 0>{});
            // ---- level l+1 ----
            constexpr int sl = (H2 - 1 + p) % H2;
            // (no control flow between a counted wait and the re-issue of its registers: the wait, the copies out of the row
            // registers and the next loads always run; WHICH value enters the ring is a select on finished values)
            if constexpr (L3) {
                asm_wait3<2 * (3 + 2) + EX>(q2[1], q2[2], q2[3]);
            } else {
                asm_wait4<2 * (3 + 2)>(q2[0], q2[1], q2[2], q2[3]);
                a2 = asm_copy(q2[0]);
            }
            v4f e = v4f{a2, asm_copy(q2[2]), asm_copy(q2[1]), asm_copy(q2[3])};
            if (!(last || s2 < nQ)) e = *lds_l2(lds_rd, s2 - nQ);  // ... from the wave below (its ring warm-up rows)
            r2av[sl] = v2f{e.x, e.y};
            r2hd[sl] = v2f{e.z, e.w};
            {
                const unsigned vo = voff2 + o2;  // stream row s2 + 1, needed one step ahead (frozen at the last one the wave loads)
                if constexpr (!L3) asm_load_s(q2[0], b.A2, vo);
                asm_load_s(q2[1], b.H2, vo);
                asm_load_s(q2[2], b.V2, vo);
                asm_load_s(q2[3], b.D2, vo);
                advance(c2, o2, last2, str2, end2);
            }
        }
        static_for<2>([&](auto I) {
            constexpr int idx = decltype(I)::value;  // 0: tap parity 1 (first A_l row of the pair), 1: parity 0 (the next row)
            constexpr int q = 2 * p + idx;           // position of the A_l row in the super-body
            const int r1 = 2 * s + idx;
            constexpr int sl = q % H2;
            if (l2act) {
                const v2f a01 = own_pair(synth_col(r2av, r2hd, std::integral_constant<int, p % H2>{}, std::integral_constant<int, 1 - idx>{}));
                // ---- level l: stream row r1 enters the ring with its H,V,D row ----
                asm_wait3<2 + (3 + 2) + NL2 + EX>(q1[idx][0], q1[idx][1], q1[idx][2]);
                ra[sl] = a01;
                rh[sl] = asm_copy(q1[idx][0]);
                rv[sl] = asm_copy(q1[idx][1]);
                rd[sl] = asm_copy(q1[idx][2]);
                {
                    const unsigned vo = voff1 + o1;  // stream row r1 + 2 (frozen at the last one)
                    asm_load_s(q1[idx][0], b.H1, vo);
                    asm_load_s(q1[idx][1], b.V1, vo);
                    asm_load_s(q1[idx][2], b.D1, vo);
                    advance(c1, o1, last1, str1, end1);
                }
                if constexpr (q < H2 - 1) {
                    // first steps: the wave above needs this wave's first H2-1 level-l ring rows
                    if (sb == 0 && kw > 0) {
                        *lds_l1(lds_wr, q, 0) = v4f{ra[sl].x, ra[sl].y, rh[sl].x, rh[sl].y};
                        *lds_l1(lds_wr, q, 1) = v4f{rv[sl].x, rv[sl].y, rd[sl].x, rd[sl].y};
                    }
                }
            } else if (r1 - nP < H2 - 1) {
                const v4f e0 = *lds_l1(lds_rd, r1 - nP, 0), e1 = *lds_l1(lds_rd, r1 - nP, 1);
                ra[sl] = v2f{e0.x, e0.y};
                rh[sl] = v2f{e0.z, e0.w};
                rv[sl] = v2f{e1.x, e1.y};
                rd[sl] = v2f{e1.z, e1.w};
            }
            // the window that ends with stream row r1 starts at r1-(H2-1): the wave owns the starts [0, nP)
            const int ws = r1 - (H2 - 1);
            const bool own = (ws >= 0) && (ws < nP);
            emit(std::integral_constant<int, (q + 1) % H2>{}, std::integral_constant<int, 1>{}, own, 2 * ws);
            emit(std::integral_constant<int, (q + 1) % H2>{}, std::integral_constant<int, 0>{}, own, 2 * ws + 1);
        });
    };

    asm_drain1(q3[0]);
    asm_drain1(q3[1]);
    asm_drain1(q3[2]);
    asm_drain1(q3[3]);
    asm_drain1(q2[0]);
    asm_drain1(q2[1]);
    asm_drain1(q2[2]);
    asm_drain1(q2[3]);
    static_for<2>([&](auto K) {
        constexpr int k = decltype(K)::value;
        asm_drain1(q1[k][0]);
        asm_drain1(q1[k][1]);
        asm_drain1(q1[k][2]);
    });
    CASC_TRACE(2);  // first row registers landed

    // ---- straight-line wave programs (cf. k_fwd2d_casc) ---------------------------------------------------------------------------
    // A wave of the default geometry lives for 6 or 8 steps; in the loop below every step pays for predicates (does the level-(l+1) part
    // run, is its row the wave's own or the neighbour's, which windows are owned), cursors that freeze at the wave's last row and wrap
    // with the image, and for loads, waits and syntheses whose results a select then drops (the last H2-1 level-(l+1) rows of a wave
    // that is not the last of its workgroup come from the hand-off, yet their A parts are synthesised and their V, H, D parts loaded).
    // Here the step index is a compile-time constant: parts that do not run are not emitted, stores of rows the wave does not own do
    // not exist, the vmcnt waits are constants of the position (CascInvSpec) and the only addressing left is three running 32-bit band
    // offsets and the output row pointer.  The image wraps inside a wave program only for the last wave of the bottom workgroups, at
    // compile-time positions.  Same arithmetic, same order: bit-identical to the loop.
    auto spec = [&](auto NQc, auto LASTc) PDWT_AI {
        {
            constexpr int NQ = decltype(NQc)::value;
            constexpr bool LV = decltype(LASTc)::value;
            using S = CascInvSpec<HLEN, NQ, LV, L3>;
            // stream rows at which the image wraps, for the (only) waves whose rows do: the last wave of a bottom workgroup
            constexpr int WR3 = NQ / 2 + C, WR2 = NQ, WR1 = 2 * NQ - 2 * C + SHIFT, WRO = 4 * NQ - 6 * C + 3 * SHIFT;
            const bool bottom = LV && (Q0 + NQ == Nr2);
            const unsigned w3 = bottom ? end3 : 0u, w2 = bottom ? end2 : 0u, w1 = bottom ? end1 : 0u;
            const size_t wo = bottom ? (size_t)Nr * Nc : (size_t)0;
            static_for<S::NSTEPS>([&](auto Ss) {
                constexpr int st = decltype(Ss)::value;  // step
                constexpr int p = st % H2;
                constexpr bool even = L3 && (st & 1) == 0;
                if constexpr (S::l2act(st)) {
                    constexpr int s2 = H2 - 1 + st;
                    constexpr int sl = s2 % H2;
                    if constexpr (S::adv3(st)) {
                        if constexpr (st >= 2) asm_wait4<S::wait3(st)>(q3[0], q3[1], q3[2], q3[3]);
#pragma unroll
                        for (int j = 0; j < H2 - 1; j++) {
                            r3av[j] = r3av[j + 1];
                            r3hd[j] = r3hd[j + 1];
                        }
                        r3av[H2 - 1] = v2f{asm_copy(q3[0]), asm_copy(q3[2])};
                        r3hd[H2 - 1] = v2f{asm_copy(q3[1]), asm_copy(q3[3])};
                        if constexpr (S::n3(st) != 0) {
                            constexpr int row = T0 + H2 + st / 2;  // level-(l+2) stream row this load fetches
                            const unsigned vo = voff3 + o3;
                            asm volatile("" : "=v"(q3[0]));
                            asm volatile("" : "=v"(q3[1]));
                            asm volatile("" : "=v"(q3[2]));
                            asm volatile("" : "=v"(q3[3]));
                            asm_load_s(q3[0], b3.A3, vo);
                            asm_load_s(q3[1], b3.H3, vo);
                            asm_load_s(q3[2], b3.V3, vo);
                            asm_load_s(q3[3], b3.D3, vo);
                            o3 += str3;
                            if constexpr (row + 1 == WR3) o3 -= w3;
                        }
                    }
                    if constexpr (S::need2(st)) {
                        float a2 = 0.f;
                        if constexpr (L3) a2 = a2_from(r3av, r3hd, std::integral_constant<int, even ? 1 : 0>{});
                        if constexpr (L3) {
                            if constexpr (st >= 1) asm_wait3<S::wait2(st)>(q2[1], q2[2], q2[3]);
                        } else {
                            if constexpr (st >= 1) asm_wait4<S::wait2(st)>(q2[0], q2[1], q2[2], q2[3]);
                            a2 = asm_copy(q2[0]);
                        }
                        r2av[sl] = v2f{a2, asm_copy(q2[2])};
                        r2hd[sl] = v2f{asm_copy(q2[1]), asm_copy(q2[3])};
                    } else {
                        const v4f e = *lds_l2(lds_rd, s2 - NQ);  // from the wave below (its ring warm-up rows)
                        r2av[sl] = v2f{e.x, e.y};
                        r2hd[sl] = v2f{e.z, e.w};
                    }
                    if constexpr (S::n2(st) != 0) {
                        const unsigned vo = voff2 + o2;  // stream row s2 + 1
                        if constexpr (!L3) {
                            asm volatile("" : "=v"(q2[0]));
                            asm_load_s(q2[0], b.A2, vo);
                        }
                        asm volatile("" : "=v"(q2[1]));
                        asm volatile("" : "=v"(q2[2]));
                        asm volatile("" : "=v"(q2[3]));
                        asm_load_s(q2[1], b.H2, vo);
                        asm_load_s(q2[2], b.V2, vo);
                        asm_load_s(q2[3], b.D2, vo);
                        o2 += str2;
                        if constexpr (s2 + 2 == WR2) o2 -= w2;
                    }
                }
                static_for<2>([&](auto I) {
                    constexpr int idx = decltype(I)::value;
                    constexpr int r1 = 2 * st + idx;
                    constexpr int sl = r1 % H2;
                    if constexpr (S::l2act(st)) {
                        const v2f a01 = own_pair(synth_col(r2av, r2hd, std::integral_constant<int, p % H2>{}, std::integral_constant<int, 1 - idx>{}));
                        if constexpr (st >= 1) asm_wait3<S::wait1(st, idx)>(q1[idx][0], q1[idx][1], q1[idx][2]);
                        ra[sl] = a01;
                        rh[sl] = asm_copy(q1[idx][0]);
                        rv[sl] = asm_copy(q1[idx][1]);
                        rd[sl] = asm_copy(q1[idx][2]);
                        if constexpr (S::n1(st) != 0) {
                            const unsigned vo = voff1 + o1;  // stream row r1 + 2
                            asm volatile("" : "=v"(q1[idx][0]));
                            asm volatile("" : "=v"(q1[idx][1]));
                            asm volatile("" : "=v"(q1[idx][2]));
                            asm_load_s(q1[idx][0], b.H1, vo);
                            asm_load_s(q1[idx][1], b.V1, vo);
                            asm_load_s(q1[idx][2], b.D1, vo);
                            o1 += str1;
                            if constexpr (r1 + 3 == WR1) o1 -= w1;
                        }
                        if constexpr (r1 < H2 - 1) {
                            if (kw > 0) {
                                *lds_l1(lds_wr, r1, 0) = v4f{ra[sl].x, ra[sl].y, rh[sl].x, rh[sl].y};
                                *lds_l1(lds_wr, r1, 1) = v4f{rv[sl].x, rv[sl].y, rd[sl].x, rd[sl].y};
                            }
                        }
                    } else if constexpr (r1 - S::NP < H2 - 1) {
                        const v4f e0 = *lds_l1(lds_rd, r1 - S::NP, 0), e1 = *lds_l1(lds_rd, r1 - S::NP, 1);
                        ra[sl] = v2f{e0.x, e0.y};
                        rh[sl] = v2f{e0.z, e0.w};
                        rv[sl] = v2f{e1.x, e1.y};
                        rd[sl] = v2f{e1.z, e1.w};
                    }
                    if constexpr (S::own(st, idx)) {
                        constexpr int ws = r1 - (H2 - 1);
                        static_for<2>([&](auto Pq) {
                            constexpr int par = decltype(Pq)::value;  // first the row with tap parity 1, then parity 0
                            constexpr int g = 2 * ws + par;            // the wave's g-th output row
                            // (one output row at a time: four interleaved row syntheses do not fit the 128 registers of a 16-wave workgroup)
                            __builtin_amdgcn_sched_barrier(0);
                            emit(std::integral_constant<int, (r1 + 1) % H2>{}, std::integral_constant<int, 1 - par>{}, std::true_type{}, g);
                            if constexpr (g + 1 == WRO) op -= wo;
                            __builtin_amdgcn_sched_barrier(0);
                        });
                    }
                });
                if constexpr (st < XS) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#ifdef PDWT_CASC_TRACE
                if constexpr (st == H2 - 1) CASC_TRACE(3);
                if constexpr (st == 2 * H2 - 1) CASC_TRACE(4);
#endif
            });
        }
    };
    auto epilogue = [&]() PDWT_AI {
        CASC_TRACE(5);  // loop left
#ifdef PDWT_CASC_TRACE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        CASC_TRACE(6);  // everything this wave issued has retired
        CASC_TRACE_STORE(trash + kCascTraceOff, blockIdx.x * W + kw, ((unsigned long long)nsteps << 32) | (unsigned)nQ);
    };
    // The wave programs live in a kernel of their own (SPEC): next to the loop in ONE kernel, hipcc's control-flow structuriser chains the
    // arms (arm 1 -> flag -> arm 2 ... -> loop) and the register allocator then keeps every arm's entry state alive, in scratch, around the
    // others -- spills of row registers whose loads are in flight included.  Every arm but the last ends the program itself (s_endpgm, not a
    // return to a common exit) for the same reason.
    if constexpr (SPEC) {
        static_assert(W == 16, "wave programs exist for the 16-wave workgroups");
        const int variant = __builtin_amdgcn_readfirstlane(last ? 3 : (nQ == 6 ? 2 : 1));
        if (variant == 1) {
            spec(std::integral_constant<int, 4>{}, std::false_type{});
            epilogue();
            __builtin_amdgcn_endpgm();
        }
        if (variant == 2) {
            spec(std::integral_constant<int, 6>{}, std::false_type{});
            epilogue();
            __builtin_amdgcn_endpgm();
        }
        spec(std::integral_constant<int, 4>{}, std::true_type{});
        epilogue();
    } else {
    for (int sb = 0;; sb++) {
        bool fin = false;
        static_for<H2>([&](auto Pp) {
            constexpr int p = decltype(Pp)::value;
            if (!fin) {
                step(Pp, sb);
                // hand-off order: level-(l+1) rows are written in the prologue and first read at step nQ-(H2-1) >= 1,
                // level-l rows are written during steps 0 .. XS-1 and first read at step nQ >= XS: one barrier after each of the
                // first XS steps (every wave runs them: nsteps > XS).  LDS only -- the global loads in flight are not drained.
                if constexpr (p < XS || p == 0) {
                    if (sb == 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                }
                fin = (sb * H2 + p + 1 >= nsteps);
            }
        });
#ifdef PDWT_CASC_TRACE
        if (sb == 0) CASC_TRACE(3);  // first super-body (H2 steps) done
        if (sb == 1) CASC_TRACE(4);  // second
#endif
        if (fin) break;
    }
    {
        asm_drain1(q3[0]);
        asm_drain1(q3[1]);
        asm_drain1(q3[2]);
        asm_drain1(q3[3]);
        asm_drain1(q2[0]);
        asm_drain1(q2[1]);
        asm_drain1(q2[2]);
        asm_drain1(q2[3]);
        static_for<2>([&](auto K) {
            constexpr int k = decltype(K)::value;
            asm_drain1(q1[k][0]);
            asm_drain1(q1[k][1]);
            asm_drain1(q1[k][2]);
        });
    }
        epilogue();
    }
}

// =================================================================================================
// host side
// =================================================================================================
#define PDWT_CHECK_LAUNCH() PDWT_HIP_TRY(hipGetLastError())

template <int HLEN, bool L3>
static int launch_inv_casc3(const CascInvBands& b, const CascInv3B& b3, float* out, float* trash, int nr, int nc, const Taps2<float>& f,
                            const CascBatchI* d_tbl, int nimg)
{
    using G = CascInvGeom<HLEN>;
    constexpr int H2 = G::H2;
    const int nc1 = nc / 2, np = nr / 8;  // level-(l+1) row PAIRS
    const int strips = idiv_up(nc1, G::MAXVL * 2);
    const int VL = idiv_up(nc1 / 2, strips);
    int Wk = knob(KN_CASC_IWG);
    if (Wk != 4 && Wk != 8 && Wk != 12 && Wk != 16) Wk = 16;
    constexpr size_t REG = casc_inv3_region_bytes<HLEN>();
    auto lds_bytes = [&](int w) { return (size_t)(w - 1) * REG; };
    const int wgs = knob(KN_CASC_IWAVES) > 0 ? idiv_up(knob(KN_CASC_IWAVES), Wk) : 256;  // default: one workgroup per CU
    // (W, gy) fits when every wave but the last gets >= H2 rows (the hand-off is first read at step nQ-(H2-1) >= 1) and the last
    // one at least one pair; the kernel's split, replayed on the two chunk sizes that occur
    auto fits = [&](int w, int g) {
        if (lds_bytes(w) > 150 * 1024) return false;
        for (int Rp : {np / g, idiv_up(np, g)}) {
            const int Ep = 0;  // (the kernel's split)
            const int basep = (Rp + Ep) / w, remp = (Rp + Ep) % w;
            if (2 * basep < H2) return false;
            const int lastp = Rp - ((w - 1) * basep + std::min(w - 1, remp));
            if (lastp < 1) return false;
        }
        return true;
    };
    int W = 0, gy = 0;
    for (int w : {Wk, 16, 12, 8, 4}) {
        for (int g = std::max(1, wgs / strips); g >= std::max(1, wgs / strips / 2) && !W; g--)
            if (fits(w, g)) {
                W = w;
                gy = g;
            }
        if (W) break;
    }
    if (!W) return 1;
    const int nwg = gy * strips;
    
    const size_t lds = lds_bytes(W);
    void (*k)(CascInvBands, CascInv3B, float*, int, int, int, float*, CascMap, Taps2<float>);
    k = (W == 4) ? k_inv2d_casc3<HLEN, 4, L3> : (W == 8) ? k_inv2d_casc3<HLEN, 8, L3> : (W == 12) ? k_inv2d_casc3<HLEN, 12, L3> : k_inv2d_casc3<HLEN, 16, L3>;
    // the straight-line wave programs (kernel form SPEC) exist for waves of 4 or 6 level-(l+1) rows whose workgroup's last wave has 4:
    // the kernel's split, replayed on the two chunk sizes that occur (C2: 36 or 37 row pairs over 16 waves)
    bool spec = W == 16 && ((knob(KN_CASC_SPEC) >> 1) & 1) && H2 <= 4;
    // XCD-weighted split (casc_chunk_start): only with the wave programs, and only if every workgroup still has them
    int xw = spec ? knob(KN_CASC_XCDW) : 0;
    for (int pass = 0; pass < 2; pass++) {
        bool ok = spec;
        for (int g = 0; g < gy && ok; g++)
            for (int st = 0; st < strips && ok; st++) {
                const int Rp = casc_chunk_start(g + 1, st, np, gy, strips, idiv_up(nwg, 8), xw) - casc_chunk_start(g, st, np, gy, strips, idiv_up(nwg, 8), xw);
                ok = Rp / 16 == 2;
            }
        if (ok || xw == 0) {
            spec = ok;
            break;
        }
        xw = 0;
    }
    if (!spec) xw = 0;
    if (spec) {
        k = k_inv2d_casc3<HLEN, 16, L3, true>;
        stat_casc_spec(1);
    }
    const CascMap cm = {idiv_up(nwg, 8), strips, gy, xw, d_tbl};
    const dim3 grid((unsigned)(8 * cm.cpx), (unsigned)(d_tbl ? nimg : 1));
    if (lds > 64 * 1024) {  // opt-in once per (kernel, device), not per launch
        const int rc = spec ? lds_opt_in<k_inv2d_casc3<HLEN, 16, L3, true>>()
                       : (W == 4) ? lds_opt_in<k_inv2d_casc3<HLEN, 4, L3>>() : (W == 8) ? lds_opt_in<k_inv2d_casc3<HLEN, 8, L3>>() : (W == 12) ? lds_opt_in<k_inv2d_casc3<HLEN, 12, L3>>() : lds_opt_in<k_inv2d_casc3<HLEN, 16, L3>>();
        if (rc != PDWT_OK) return rc;
    }
    KTimer kt(K_INV2D_CASC, true);
    PDWT_LAUNCH_KT(kt, k, grid, dim3(64 * W), lds, b, b3, out, nr, nc, VL, trash, cm, f);
    PDWT_CHECK_LAUNCH();
    return PDWT_OK;
}

// Three levels, all streamed (A3 != NULL), or two (A3 == NULL: A2 is read).  PDWT_OK when launched, 1 when the geometry / filter
// length is outside this path (the caller falls back to dwt_casc_invw.hip).
int inv2d_casc3_f32(const float* A2, const float* H2, const float* V2, const float* D2, const float* H1, const float* V1, const float* D1,
                    const float* A3, const float* H3, const float* V3, const float* D3, float* out, float* trash, int nr, int nc, int hlen,
                    const Taps2<float>& f, const CascBatchI* d_tbl, int nimg)
{
    const bool l3 = A3 != nullptr;
    if ((nr & 7) || (nc & 7) || nc < 256 || nr < 32 * hlen) return 1;
    if ((long long)nr * nc >= (1LL << 32)) return 1;  // 32-bit row offsets inside a level-l band: larger images take one launch per level
    if (!al16(out) || !al16(H1) || !al16(V1) || !al16(D1) || !al16(H2) || !al16(V2) || !al16(D2) || !al16(trash)) return 1;
    if (!l3 && !al16(A2)) return 1;
    const CascInvBands b = {A2, H2, V2, D2, H1, V1, D1};
    const CascInv3B b3 = {A3, H3, V3, D3};
    switch (hlen) {
        case 4: return l3 ? launch_inv_casc3<4, true>(b, b3, out, trash, nr, nc, f, d_tbl, nimg) : launch_inv_casc3<4, false>(b, b3, out, trash, nr, nc, f, d_tbl, nimg);
        case 8: return l3 ? launch_inv_casc3<8, true>(b, b3, out, trash, nr, nc, f, d_tbl, nimg) : launch_inv_casc3<8, false>(b, b3, out, trash, nr, nc, f, d_tbl, nimg);
        case 12: return l3 ? launch_inv_casc3<12, true>(b, b3, out, trash, nr, nc, f, d_tbl, nimg) : launch_inv_casc3<12, false>(b, b3, out, trash, nr, nc, f, d_tbl, nimg);
        case 20: return l3 ? launch_inv_casc3<20, true>(b, b3, out, trash, nr, nc, f, d_tbl, nimg) : launch_inv_casc3<20, false>(b, b3, out, trash, nr, nc, f, d_tbl, nimg);
        case 16: return l3 ? launch_inv_casc3<16, true>(b, b3, out, trash, nr, nc, f, d_tbl, nimg) : launch_inv_casc3<16, false>(b, b3, out, trash, nr, nc, f, d_tbl, nimg);
        default: return 1;
    }
}

}  // namespace pdwt
