// dwt_casc_invw.hip -- inverse 2D DWT, TWO or THREE levels per launch, float32, workgroup form.
//
// Same per-wave arithmetic as k_inv2d_casc (dwt_casc.hip: level-(l+1) column synthesis -> DPP halo -> row synthesis ->
// two rows of A_l in registers -> level-l ring with the H,V,D rows -> output rows), organised differently:
//
//  * W waves of a workgroup are stacked vertically in ONE strip.  A wave OWNS the level-(l+1) coefficient rows
//    [Q0, Q0+nQ) it loads first-hand and emits exactly the rows whose synthesis windows START there:
//      level-(l+1) window start q   ->  A_l rows      2(q+C)-SHIFT, +1      (stream r1 = 0 <-> A_l row P0 = 2(Q0+C)-SHIFT)
//      level-l     window start p   ->  output rows   2(p+C)-SHIFT, +1      (stream g  = 0 <-> output row O0 = 2(P0+C)-SHIFT)
//    The last H2-1 windows of either level reach into the rows of the wave BELOW: that wave has them in its rings anyway
//    (its first H2-1 ring rows) and drops them in LDS during its first steps; a barrier after each of those steps orders
//    the hand-off, and a wave picks its bottom halo up at the END of its chunk.  Only the last wave of a workgroup loads
//    (level l+1: H2-1 rows + the windows of XS extra steps) and recomputes the halo, as every wave of k_inv2d_casc does.
//  * L3: the approximation band of level l+1 is not read from memory at all.  Every wave synthesises the A_{l+1} rows it
//    owns from the level-(l+2) bands in a prologue (redundant only in the H2-1 window rows: level l+2 is 1/16 of the data)
//    into a private LDS area and reads them from there -- the level-(l+2) inverse launch (4.8 us at C2, latency-bound) and
//    the round trip of A_{l+1} through memory disappear.
// Reference code replaced: two / three iterations of the level loop of w_inverse_separable (src/separable.cu:332-364) with
// their two kernels each (:246-328).
#include "casc_dev.hpp"
#include "dwt_stream.hpp"
#include <algorithm>

#include "stream_dev.hpp"

namespace pdwt {

struct CascInv3 {
    const float *A3, *H3, *V3, *D3;
};

// LDS hand-off region of one CONSUMER wave: (H2-1) level-(l+1) ring rows (A,V,H,D of the lane's column: 16 B) and
// (H2-1) level-l ring rows (the lane's two columns of A,H,V,D: 32 B)
template <int HLEN>
constexpr int casc_inv_region_bytes() { return (HLEN / 2 - 1) * 64 * (16 + 32); }
constexpr int kInvA2Rows = 16;  // three-level form: private A_{l+1} rows per wave (stream rows, 256 B each)
constexpr int kInvR3Max = 12;   // ... and level-(l+2) rows a wave may load for them

template <int HLEN, int W, bool L3>
__global__ __launch_bounds__(64 * W) void k_inv2d_cascw(CascInvBands b, CascInv3 b3, float* __restrict__ out, int Nr, int Nc, int VL,
                                                         float* __restrict__ trash, CascMap cm, Taps2<float> f)
{
    using G = CascInvGeom<HLEN>;
    constexpr int H2 = G::H2, C = G::C, SHIFT = G::SHIFT, NB1 = G::NB1, NB2 = G::NB2, NBT = G::NBT, WIN1 = G::WIN1, WIN2 = G::WIN2;
    constexpr int XS = H2 / 2;                 // extra (level-l only) steps that drain the last H2-1 level-l windows
    constexpr int NL2 = L3 ? 3 : 4;            // level-(l+1) loads per step
    constexpr int kWait2 = 2 * (3 + 2);        // VMEM between a step's level-(l+1) loads and their use one step later
    constexpr int kWait1 = 2 + (3 + 2) + NL2;  // ... between an A_l row's loads and their use one step later
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    CASC_TRACE_DECL;
    CASC_TRACE(0);
    const int lane = threadIdx.x & 63;
    const int kw = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int Nr1 = Nr >> 1, Nc1 = Nc >> 1, Nr2 = Nr >> 2, Nc2 = Nc >> 2;
    // workgroup -> (workgroup-chunk row, strip); XCD x owns the logical workgroups [x*cpx, (x+1)*cpx)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int wg = xcd * cm.cpx + slot;
    if (slot >= cm.cpx || wg >= cm.gy * cm.strips) return;  // (uniform over the workgroup: nobody is left at a barrier)
    const int gy = wg / cm.strips, strip = wg % cm.strips;
    const int J0 = (int)(((long long)gy * Nr2) / cm.gy);
    const int R = (int)(((long long)(gy + 1) * Nr2) / cm.gy) - J0;
    // split of the R level-(l+1) rows: the last wave runs XS steps more than it owns rows (the recomputed halo);
    // the host guarantees R / W >= H2
    const int E = min(XS, R / W - 1);
    const int base = (R + E) / W, rem = (R + E) % W;
    const int start = kw * base + min(kw, rem);
    const int nQ = (kw < W - 1) ? base + (kw < rem ? 1 : 0) : R - start;
    const int Q0 = J0 + start;
    const bool last = (kw == W - 1);
    const int nsteps = nQ + XS;
    const int P0 = 2 * (Q0 + C) - SHIFT;  // A_l row of stream index r1 = 0
    const int O0 = 2 * (P0 + C) - SHIFT;  // output row of stream index g = 0
    const int nP = 2 * nQ;                // level-l windows (= output row pairs) the wave owns
    // last stream rows the wave loads itself
    const int last2 = last ? H2 - 2 + nsteps : nQ - 1;
    const int last1 = last ? 2 * nsteps - 1 : nP - 1;

    const int cx1 = strip * VL * 2 + 2 * (lane - NBT);  // first of the lane's two level-l coefficient columns
    const bool valid = (lane >= NBT) && (lane < NBT + VL) && (cx1 < Nc1);
    const int cx1w = wrapi(cx1, Nc1);
    const int cx2w = cx1w >> 1;
    const float* const pA2 = b.A2 + cx2w;
    const float* const pH2 = b.H2 + cx2w;
    const float* const pV2 = b.V2 + cx2w;
    const float* const pD2 = b.D2 + cx2w;
    const float* const pH1 = b.H1 + cx1w;
    const float* const pV1 = b.V1 + cx1w;
    const float* const pD1 = b.D1 + cx1w;
    // single conditional wraps: Q0 < Nr2, P0 < Nr1 + 2C, O0 < Nr + 6C and a wave's rows are few (the host keeps chunks < Nr2/2)
    auto off2 = [&](int s2) { return (size_t)wrap1(Q0 + s2, Nr2) * Nc2; };
    auto off1 = [&](int r1) { return (size_t)wrap1(P0 + r1, Nr1) * Nc1; };
    const unsigned voff2 = (unsigned)cx2w * 4u, voff1 = (unsigned)cx1w * 4u, voffo = (unsigned)(valid ? cx1 : 0) * 8u;
    const lanemask_t vmask = __ballot(valid);

    // LDS: [hand-off regions][private A_{l+1} rows]
    constexpr int REG = casc_inv_region_bytes<HLEN>();
    unsigned char* const lds_rd = lds_raw + (size_t)kw * REG;                    // written by wave kw+1
    unsigned char* const lds_wr = lds_raw + (size_t)(kw > 0 ? kw - 1 : 0) * REG;  // read by wave kw-1
    auto lds_l2 = [&](unsigned char* reg, int r) { return reinterpret_cast<v4f*>(reg + ((size_t)r * 64 + lane) * 16); };
    auto lds_l1 = [&](unsigned char* reg, int r, int h) {
        return reinterpret_cast<v4f*>(reg + (size_t)(H2 - 1) * 64 * 16 + (((size_t)r * 2 + h) * 64 + lane) * 16);
    };
    float* const lds_a2 = reinterpret_cast<float*>(lds_raw + (size_t)(W - 1) * REG + (size_t)kw * kInvA2Rows * 256);

    v2f r2av[H2], r2hd[H2];               // level l+1 ring: (A,V) and (H,D) of the lane's column
    v2f ra[H2], rh[H2], rv[H2], rd[H2];   // level l ring: the lane's two columns of each band
#pragma unroll
    for (int k = 0; k < H2; k++) ra[k] = rh[k] = rv[k] = rd[k] = v2f{0.f, 0.f};

    // one row of A_l (the lane's two columns) from a level-(l+1)-style ring window starting at slot S0, tap parity OFF
    auto synth_pair = [&](const v2f (&av)[H2], const v2f (&hd)[H2], auto S0, auto OFF, float& a0, float& a1) {
        constexpr int s0 = decltype(S0)::value, off = decltype(OFF)::value;
        v2f sav = {0.f, 0.f}, shd = {0.f, 0.f};
        static_for<H2>([&](auto J) {
            constexpr int j = decltype(J)::value;
            constexpr int s = (s0 + j) % H2;
            constexpr int k = HLEN - 1 - (2 * j + off);
            constexpr int kp = k & ~1;  // tap k as one half of its aligned pair: no splat copies in scalar registers (dwt_stream.hip)
            sav = pk_fma_sbcast<k & 1, false>(av[s], v2f{f.a[kp], f.a[kp + 1]}, sav);
            shd = pk_fma_sbcast<k & 1, false>(hd[s], v2f{f.b[kp], f.b[kp + 1]}, shd);
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

    // ring warm-up rows and the first row registers: issued BEFORE the three-level prologue so that their latency overlaps it
    float q2[4];   // row registers in flight, level l+1 (prefetch distance: one step)
    v2f q1[2][3];  // row registers in flight, level l (two A_l rows per step)
    {
#pragma unroll
        for (int r = 0; r < H2 - 1; r++) {
            const size_t o = off2(r);
            r2av[r] = v2f{L3 ? 0.f : pA2[o], pV2[o]};
            r2hd[r] = v2f{pH2[o], pD2[o]};
        }
        r2av[H2 - 1] = r2hd[H2 - 1] = v2f{0.f, 0.f};
        {
            const size_t o = off2(H2 - 1);
            q2[0] = L3 ? 0.f : pA2[o];
            q2[1] = pH2[o];
            q2[2] = pV2[o];
            q2[3] = pD2[o];
        }
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const size_t o = off1(q);
            q1[q][0] = *reinterpret_cast<const v2f*>(pH1 + o);
            q1[q][1] = *reinterpret_cast<const v2f*>(pV1 + o);
            q1[q][2] = *reinterpret_cast<const v2f*>(pD1 + o);
        }
    }
    // ---- three-level form: the wave's A_{l+1} rows from the level-(l+2) bands, into its private LDS rows ----------
    if constexpr (L3) {
        const int Nr3 = Nr >> 3, Nc3 = Nc >> 3;
        const int nA2 = min(last ? H2 - 1 + nsteps : nQ, kInvA2Rows);  // stream rows s2 in [0, nA2) are read from here
        // pair index P3 <-> A_{l+1} rows 2*P3-SHIFT, +1  (window start P3 - C)
        const int P3a = (Q0 + SHIFT) >> 1, P3b = (Q0 + nA2 - 1 + SHIFT) >> 1;
        const int np = P3b - P3a + 1;    // pairs to synthesise
        const int nr3 = np + H2 - 1;     // level-(l+2) rows needed (<= kInvR3Max, checked by the host)
        // lane <-> level-(l+2) column; its two A_{l+1} columns are 2*c3, 2*c3+1; lanes [C, 64-C) are valid after the DPP halo
        const int a2c0 = strip * VL - NBT;                    // A_{l+1} column of lane 0 in the main mapping (may be < 0)
        const int c3 = (a2c0 >> 1) - C + lane;                // (arithmetic shift: floor)
        const int c3w = wrapi(c3, Nc3);
        v2f w3av[kInvR3Max], w3hd[kInvR3Max];
#pragma unroll
        for (int r = 0; r < kInvR3Max; r++) {
            const size_t o = (size_t)wrapi(P3a - C + min(r, nr3 - 1), Nr3) * Nc3 + c3w;
            w3av[r] = v2f{b3.A3[o], b3.V3[o]};
            w3hd[r] = v2f{b3.H3[o], b3.D3[o]};
        }
        const int colbase = 2 * c3 - a2c0;  // LDS column of the lane's first A_{l+1} column
        const bool lane_ok = (lane >= C) && (lane < 64 - C);
        static_for<kInvR3Max - H2 + 1>([&](auto PI) {
            constexpr int pi = decltype(PI)::value;
            if (pi < np) {
                v2f av[H2], hd[H2];
#pragma unroll
                for (int j = 0; j < H2; j++) {
                    av[j] = w3av[pi + j];
                    hd[j] = w3hd[pi + j];
                }
#pragma unroll
                for (int idx = 0; idx < 2; idx++) {
                    float a0, a1;
                    if (idx == 0) synth_pair(av, hd, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, a0, a1);
                    else synth_pair(av, hd, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, a0, a1);
                    const int s2 = 2 * (P3a + pi) - SHIFT + idx - Q0;  // stream row of this A_{l+1} row
                    if (s2 >= 0 && s2 < nA2 && lane_ok) {
                        if (colbase >= 0 && colbase < 64) lds_a2[s2 * 64 + colbase] = a0;
                        if (colbase + 1 >= 0 && colbase + 1 < 64) lds_a2[s2 * 64 + colbase + 1] = a1;
                    }
                }
            }
        });
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    }
    auto a2_lds = [&](int s2) { return lds_a2[min(s2, kInvA2Rows - 1) * 64 + lane]; };
    CASC_TRACE(1);  // three-level prologue done (two-level form: nothing happened yet)

    if constexpr (L3) {
        // the A parts of the level-(l+1) ring rows and row registers, now that the private rows exist
#pragma unroll
        for (int r = 0; r < H2 - 1; r++) r2av[r].x = a2_lds(r);
        q2[0] = a2_lds(H2 - 1);
    }
    // hand over the level-(l+1) ring warm-up rows: the wave above completes its last H2-1 windows with them
    if (kw > 0) {
#pragma unroll
        for (int r = 0; r < H2 - 1; r++) *lds_l2(lds_wr, r) = v4f{r2av[r].x, r2av[r].y, r2hd[r].x, r2hd[r].y};
    }

    float* const tr = trash + (size_t)(blockIdx.x & 7) * Nc;  // a trash ROW (the dispatcher checks the area holds 8 of them)

    // one output row of level l from the ring window starting at slot S0 with tap parity OFF (cf. k_inv2d_stream::emit)
    auto emit = [&](auto S0, auto OFF, bool own, int g) {
        constexpr int s0 = decltype(S0)::value, off = decltype(OFF)::value;
        // rows of the ring warm-up / beyond the wave's windows: the store is still issued (to a trash row, the VMEM count must
        // not change) but the arithmetic is skipped -- a uniform branch
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
            auto single_out = [&](auto Ee) {
                constexpr int eo = decltype(Ee)::value;
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
        asm_store_sm(own ? out + (size_t)wrap1(O0 + g, Nr) * Nc : tr, voffo, v4f{o4[0], o4[1], o4[2], o4[3]}, vmask);
    };

    auto step = [&](auto Pp, int sb) {
        constexpr int p = decltype(Pp)::value;
        const int s = sb * H2 + p;
        const bool l2act = last || (s < nQ);  // the level-(l+1) part runs (afterwards: level-l rows from the hand-off only)
        if (l2act) {
            // ---- level l+1: coefficient row s2 = H2-1+s completes the window that starts at stream row s ----
            const int s2 = H2 - 1 + s;
            constexpr int sl = (H2 - 1 + p) % H2;
            // (no control flow between a counted wait and the re-issue of its registers: the wait, the copies out of the row
            // registers and the next loads always run; WHICH value enters the ring is a select on finished values)
            asm_wait4<kWait2>(q2[0], q2[1], q2[2], q2[3]);
            v4f e = v4f{asm_copy(q2[0]), asm_copy(q2[2]), asm_copy(q2[1]), asm_copy(q2[3])};
            if (!(last || s2 < nQ)) e = *lds_l2(lds_rd, s2 - nQ);  // ... from the wave below (its ring warm-up rows)
            r2av[sl] = v2f{e.x, e.y};
            r2hd[sl] = v2f{e.z, e.w};
            {
                const size_t o = off2(min(s2 + 1, last2));  // the row needed one step ahead (clamped to the last one the wave loads)
                if constexpr (L3) q2[0] = a2_lds(min(s2 + 1, last2));
                else asm_load_s(q2[0], b.A2 + o, voff2);
                asm_load_s(q2[1], b.H2 + o, voff2);
                asm_load_s(q2[2], b.V2 + o, voff2);
                asm_load_s(q2[3], b.D2 + o, voff2);
            }
        }
        static_for<2>([&](auto I) {
            constexpr int idx = decltype(I)::value;  // 0: tap parity 1 (first A_l row of the pair), 1: parity 0 (the next row)
            constexpr int q = 2 * p + idx;           // position of the A_l row in the super-body
            const int r1 = 2 * s + idx;
            constexpr int sl = q % H2;
            if (l2act) {
                float a0, a1;
                synth_pair(r2av, r2hd, std::integral_constant<int, p % H2>{}, std::integral_constant<int, 1 - idx>{}, a0, a1);
                // ---- level l: stream row r1 enters the ring with its H,V,D row ----
                asm_wait3<kWait1>(q1[idx][0], q1[idx][1], q1[idx][2]);
                ra[sl] = v2f{a0, a1};
                rh[sl] = asm_copy(q1[idx][0]);
                rv[sl] = asm_copy(q1[idx][1]);
                rd[sl] = asm_copy(q1[idx][2]);
                {
                    const size_t o = off1(min(r1 + 2, last1));
                    asm_load_s(q1[idx][0], b.H1 + o, voff1);
                    asm_load_s(q1[idx][1], b.V1 + o, voff1);
                    asm_load_s(q1[idx][2], b.D1 + o, voff1);
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
    CASC_TRACE(2);  // ring warm-up rows and first row registers landed
    for (int sb = 0;; sb++) {
        bool fin = false;
        static_for<H2>([&](auto Pp) {
            constexpr int p = decltype(Pp)::value;
            if (!fin) {
                step(Pp, sb);
                // hand-off order: level-(l+1) rows are written in the prologue and first read at step nQ-(H2-1) >= 1, level-l
                // rows are written during steps 0 .. XS-1 and first read at step nQ >= XS: one barrier after each of the first
                // XS steps (every wave runs them: nsteps > XS).  LDS only -- the global loads in flight are not drained.
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
    CASC_TRACE(5);  // loop left
    asm_drain1(q2[1]);
    asm_drain1(q2[2]);
    asm_drain1(q2[3]);
    static_for<2>([&](auto K) {
        constexpr int k = decltype(K)::value;
        asm_drain1(q1[k][0]);
        asm_drain1(q1[k][1]);
        asm_drain1(q1[k][2]);
    });
    CASC_TRACE(6);  // everything this wave issued has retired
    CASC_TRACE_STORE(trash + kCascTraceOff, blockIdx.x * W + kw, ((unsigned long long)nsteps << 32) | (unsigned)nQ);
}

// =================================================================================================
// host side
// =================================================================================================
#define PDWT_CHECK_LAUNCH() PDWT_HIP_TRY(hipGetLastError())

template <int HLEN>
static int launch_inv_cascw(const CascInvBands& b, const CascInv3* b3, float* out, float* trash, int nr, int nc, const Taps2<float>& f)
{
    using G = CascInvGeom<HLEN>;
    constexpr int H2 = G::H2, XS = H2 / 2;
    const int nc1 = nc / 2, nr2 = nr / 4;
    const int strips = idiv_up(nc1, G::MAXVL * 2);
    const int VL = idiv_up(nc1 / 2, strips);
    int Wk = knob(KN_CASC_IWG);
    if (Wk != 4 && Wk != 8 && Wk != 16) Wk = 8;
    constexpr size_t REG = casc_inv_region_bytes<HLEN>();
    const bool l3 = b3 != nullptr;
    auto lds_bytes = [&](int w) { return (size_t)(w - 1) * REG + (l3 ? (size_t)w * kInvA2Rows * 256 : 0); };
    const int wgs = knob(KN_CASC_IWAVES) > 0 ? idiv_up(knob(KN_CASC_IWAVES), Wk) : 256;  // default: one workgroup per CU
    // the kernel's split of R level-(l+1) rows over W waves, replayed: rows of the largest middle wave and of the last wave
    auto split = [&](int R, int w, int* mid, int* lastw) {
        const int E = std::min(XS, R / w - 1);
        const int base = (R + E) / w, rem = (R + E) % w;
        *mid = base + (rem > 0 ? 1 : 0);
        *lastw = R - ((w - 1) * base + std::min(w - 1, rem));
    };
    // (W, gy) fits when every wave gets >= H2 rows and, in the three-level form, no wave needs more than kInvA2Rows private
    // A_{l+1} rows (a middle wave: its own rows; the last one: + the recomputed halo) or kInvR3Max level-(l+2) rows
    auto fits = [&](int w, int g) {
        if (lds_bytes(w) > 150 * 1024) return false;
        for (int R : {nr2 / g, idiv_up(nr2, g)}) {
            if (R / w < H2) return false;
            if (l3) {
                int mid, lw;
                split(R, w, &mid, &lw);
                const int na2 = std::max(mid, lw + H2 - 1 + XS);
                if (na2 > kInvA2Rows || (na2 + 1) / 2 + 1 + H2 - 1 > kInvR3Max) return false;
            }
        }
        return true;
    };
    int W = 0, gy = 0;
    for (int w : {Wk, 8, 16, 4}) {
        for (int g = std::max(1, wgs / strips); g >= std::max(1, wgs / strips / 2) && !W; g--)
            if (fits(w, g)) {
                W = w;
                gy = g;
            }
        if (W) break;
    }
    if (!W) return 1;
    const int nwg = gy * strips;
    const CascMap cm = {idiv_up(nwg, 8), strips, gy};
    const dim3 grid((unsigned)(8 * cm.cpx));
    const size_t lds = lds_bytes(W);
    void (*k)(CascInvBands, CascInv3, float*, int, int, int, float*, CascMap, Taps2<float>);
    if (l3) k = (W == 4) ? k_inv2d_cascw<HLEN, 4, true> : (W == 8) ? k_inv2d_cascw<HLEN, 8, true> : k_inv2d_cascw<HLEN, 16, true>;
    else k = (W == 4) ? k_inv2d_cascw<HLEN, 4, false> : (W == 8) ? k_inv2d_cascw<HLEN, 8, false> : k_inv2d_cascw<HLEN, 16, false>;
    if (lds > 64 * 1024) {  // opt-in once per (kernel, device), not per launch
        int rc;
        if (l3) rc = (W == 4) ? lds_opt_in<k_inv2d_cascw<HLEN, 4, true>>() : (W == 8) ? lds_opt_in<k_inv2d_cascw<HLEN, 8, true>>() : lds_opt_in<k_inv2d_cascw<HLEN, 16, true>>();
        else rc = (W == 4) ? lds_opt_in<k_inv2d_cascw<HLEN, 4, false>>() : (W == 8) ? lds_opt_in<k_inv2d_cascw<HLEN, 8, false>>() : lds_opt_in<k_inv2d_cascw<HLEN, 16, false>>();
        if (rc != PDWT_OK) return rc;
    }
    const CascInv3 z3 = l3 ? *b3 : CascInv3{nullptr, nullptr, nullptr, nullptr};
    KTimer kt(K_INV2D_CASC, true);
    PDWT_LAUNCH_KT(kt, k, grid, dim3(64 * W), lds, b, z3, out, nr, nc, VL, trash, cm, f);
    PDWT_CHECK_LAUNCH();
    return PDWT_OK;
}

#define PDWT_CASCW_INV_HLENS(X) X(4) X(6) X(8) X(10) X(12) X(14) X(16) X(18)

// A3 != NULL: three levels (A2 is not read: it is synthesised from the level-(l+2) bands on the fly)
int inv2d_cascw_f32(const float* A2, const float* H2, const float* V2, const float* D2, const float* H1, const float* V1, const float* D1,
                    const float* A3, const float* H3, const float* V3, const float* D3, float* out, float* trash, int nr, int nc, int hlen,
                    const Taps2<float>& f)
{
    if (knob(KN_CASC) != 1 || knob(KN_CASC_IWG) == 1 || !stream_enabled() || !trash) return 1;
    const bool l3 = A3 != nullptr;
    if (l3 && knob(KN_CASC_L3) != 1 && knob(KN_CASC_L3) != 2) return 1;
    if (knob(KN_CASC_L3) == 1 && !(knob(KN_CASC_MIN) > (long long)nr * nc)) {
        // dwt_casc_inv3.hip first: natural-pair arithmetic, all levels streamed (hlen 4 / 8, sizes divisible by 8); this file's
        // kernels take what that one does not (casc_l3 = 2 forces them)
        const int rc = inv2d_casc3_f32(A2, H2, V2, D2, H1, V1, D1, A3, H3, V3, D3, out, trash, nr, nc, hlen, f);
        if (rc <= 0) return rc;
    }
    const int m = l3 ? 7 : 3;
    if ((nr & m) || (nc & m) || nc < 256 || nr < 32 * hlen) return 1;
    if ((long long)nr * nc < (long long)knob(KN_CASC_MIN)) return 1;
    if (!al16(out) || !al16(H1) || !al16(V1) || !al16(D1) || !al16(H2) || !al16(V2) || !al16(D2) || !al16(trash)) return 1;
    if (!l3 && !al16(A2)) return 1;
    const CascInvBands b = {A2, H2, V2, D2, H1, V1, D1};
    const CascInv3 b3 = {A3, H3, V3, D3};
    switch (hlen) {
#define X(H) \
    case H: return launch_inv_cascw<H>(b, l3 ? &b3 : nullptr, out, trash, nr, nc, f);
        PDWT_CASCW_INV_HLENS(X)
#undef X
        default: return 1;
    }
}

}  // namespace pdwt
