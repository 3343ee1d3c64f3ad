// dwt_stream.hip -- the 2D DWT level kernels of the hot path, float32, "streaming" form for gfx950.
//
// One WAVE (64 lanes) owns a vertical strip of the level and walks down it row by row; a workgroup is
// just 4 independent waves (4 adjacent strips).  No LDS, no __syncthreads.
//   forward : lane = 4 consecutive input columns (one 16-byte load per row, 1 KiB per wave-row);
//             the (hlen-2)/2-sample halo on each side comes from the neighbouring lanes through DPP
//             wave shifts (v_mov_b32_dpp wave_shr:1 / wave_shl:1); the row pass leaves lo/hi for
//             2 output columns per lane in a register ring of hlen rows; every second input row the
//             column pass emits one row of A,H,V,D (8-byte stores).
//   inverse : lane = 2 coefficient columns of each band (8-byte loads); a register ring of hlen/2
//             coefficient rows feeds the column synthesis (two output rows per new coefficient row);
//             the row synthesis takes its halo of (t1,t2) from neighbouring lanes by DPP and stores
//             16 bytes per lane.
// Strips overlap by the halo lanes (NB per side; 62 valid lanes of 64 for hlen <= 10), so a wave never
// needs data from another wave and the periodic wrap is just the lane -> column map.
// Memory pipeline: the row registers of one unrolled body (hlen rows forward, hlen/2 inverse) are
// re-issued for the rows ONE BODY AHEAD as soon as the row pass has consumed them, so every wave keeps
// ~hlen KiB of loads in flight while it computes (the per-CU load path, ~10 B/clk, is the resource to
// keep busy -- MI355X_MICROARCH.md), without a second register buffer.
// Re-reads: only the hlen-2 halo rows between vertically adjacent chunks (L2 / Infinity Cache hits).
//
// Arithmetic (tap order, one FMA per tap, pass order) is identical to the tiled kernels in dwt.hip
// and to the CPU oracle, so the outputs are bit-identical to both (tests/test_gpu_parity.py).
// Reference code replaced: w_kern_forward_pass1/2, w_kern_inverse_pass1/2 (src/separable.cu:91-328).
#include "dwt_stream.hpp"

#include <algorithm>
#include "stream_dev.hpp"

namespace pdwt {

// =================================================================================================
// forward
// =================================================================================================
template <int HLEN, int NIN_>
struct FwdGeom {
    static constexpr int NIN = NIN_;                             // input columns per lane (4: 16-byte loads, 2: 8-byte)
    static constexpr int C = HLEN / 2 - 1;                       // halo samples on each side of a lane's columns
    static constexpr int NB = C > 0 ? (C + NIN - 1) / NIN : 0;   // halo lanes per side
    static constexpr int WIN = NIN * (2 * NB + 1);               // window registers
    static constexpr int MAXVL = 64 - 2 * NB;                    // lanes that produce output
};

template <int HLEN, int NIN>
__global__ __launch_bounds__(256) void k_fwd2d_stream(const float* __restrict__ in, float* __restrict__ cA, float* __restrict__ cH,
                                                       float* __restrict__ cV, float* __restrict__ cD, int Nr, int Nc, int R, int VL,
                                                       float* __restrict__ trash, int trash_mask, ChunkMap cm, TapsLH f,
                                                       const StreamBatchF* __restrict__ batch)
{
    if (batch) {  // batched launch: image blockIdx.y (uniform: the table entry arrives by scalar loads)
        const StreamBatchF e = batch[blockIdx.y];
        in = e.in;
        cA = e.cA;
        cH = e.cH;
        cV = e.cV;
        cD = e.cD;
        trash = e.trash;
    }
    using G = FwdGeom<HLEN, NIN>;
    using VIN = typename VecOf<NIN>::type;
    constexpr int NB = G::NB, C = G::C, WIN = G::WIN, P = NIN / 2;
    int cy, bx;
    if (!chunk_of_block(cm, cy, bx)) return;
    const int lane = threadIdx.x & 63;
    const int strip = bx * 4 + (threadIdx.x >> 6);
    const int xs = strip * VL * NIN;  // first input column this strip produces outputs for
    if (xs >= Nc) return;
    const int Nr2 = Nr >> 1, Nc2 = Nc >> 1;
    const int y0 = cy * R;
    const int rows = min(R, Nr2 - y0);
    if (rows <= 0) return;
    const int x = xs + NIN * (lane - NB);
    const bool valid = (lane >= NB) && (lane < NB + VL) && (x < Nc);
    const int xo = wrapi(x, Nc);  // halo / overhanging lanes hold the periodic continuation
    const int yb = 2 * y0 - C;
    const int nin = 2 * rows + HLEN - 2;  // input rows this chunk consumes

    v2f ring[HLEN][P];  // register ring: (lo,hi) row-pass results of the last HLEN input rows

    auto load_row = [&](int r, VIN& v) { v = *reinterpret_cast<const VIN*>(in + (size_t)wrap1(yb + r, Nr) * Nc + xo); };

    auto row_pass = [&](const VIN& v, v2f (&lh)[P]) {
        float w[WIN];
#pragma unroll
        for (int q = 0; q < NIN; q++) w[NB * NIN + q] = v[q];
#pragma unroll
        for (int k = 0; k < NB; k++) {
            const int dl = (NB - 1 - k) * NIN, sl = (NB - k) * NIN, dr = (NB + 1 + k) * NIN, sr = (NB + k) * NIN;
#pragma unroll
            for (int q = 0; q < NIN; q++) {
                w[dl + q] = dpp_shr1(w[sl + q]);
                w[dr + q] = dpp_shl1(w[sr + q]);
            }
        }
#pragma unroll
        for (int p = 0; p < P; p++) {
            v2f acc = {0.f, 0.f};
#pragma unroll
            for (int j = 0; j < HLEN; j++) acc = pk_fma(splat(w[NB * NIN - C + 2 * p + j]), f.t[HLEN - 1 - j], acc);
            lh[p] = acc;
        }
    };

    // issue everything the first body needs in one go: ring prologue rows 0..HLEN-3 and body rows
    VIN v[HLEN];
    {
        VIN pv[HLEN > 2 ? HLEN - 2 : 1];
#pragma unroll
        for (int r = 0; r < HLEN - 2; r++) load_row(r, pv[r]);
#pragma unroll
        for (int u = 0; u < HLEN; u++)
            if (HLEN - 2 + u < nin) load_row(HLEN - 2 + u, v[u]);
        static_for<HLEN - 2>([&](auto Rr) {
            constexpr int r = decltype(Rr)::value;
            row_pass(pv[r], ring[r]);
        });
    }

    const size_t ocol = (size_t)(x >> 1);
    // one unrolled body = HLEN input rows -> HLEN/2 output rows.  FAST: the body and the one after it
    // are both complete -> no wave-level branches, so the compiler can count the loads it leaves in flight.
    auto body = [&](auto FAST, int q0) {
        constexpr bool fast = decltype(FAST)::value;
        const int rnext = 2 * q0 + 2 * HLEN - 2;  // chunk-local input row that v[0] holds in the NEXT body
        static_for<HLEN / 2>([&](auto U) {
            constexpr int u = decltype(U)::value;
            if (fast || q0 + u < rows) {
                constexpr int s0 = (2 * u + HLEN - 2) % HLEN, s1 = (2 * u + HLEN - 1) % HLEN;
                row_pass(v[2 * u], ring[s0]);
                row_pass(v[2 * u + 1], ring[s1]);
                // the two row registers are free: re-issue them for the rows one body ahead
                if (fast || rnext + 2 * u + 1 < nin) {
                    load_row(rnext + 2 * u, v[2 * u]);
                    load_row(rnext + 2 * u + 1, v[2 * u + 1]);
                }
                v2f ah[P], vd[P];  // (A,H) from the lo branch, (V,D) from the hi branch
#pragma unroll
                for (int p = 0; p < P; p++) ah[p] = vd[p] = v2f{0.f, 0.f};
                static_for<HLEN>([&](auto J) {
                    constexpr int j = decltype(J)::value;
                    constexpr int s = (2 * u + j) % HLEN;
                    const v2f t = f.t[HLEN - 1 - j];
#pragma unroll
                    for (int p = 0; p < P; p++) {
                        ah[p] = pk_fma(splat(ring[s][p].x), t, ah[p]);
                        vd[p] = pk_fma(splat(ring[s][p].y), t, vd[p]);
                    }
                });
                float a[P], hh[P], vv[P], dd[P];
#pragma unroll
                for (int p = 0; p < P; p++) {
                    a[p] = ah[p].x;
                    hh[p] = ah[p].y;
                    vv[p] = vd[p].x;
                    dd[p] = vd[p].y;
                }
                if (valid) {
                    const size_t o = (size_t)(y0 + q0 + u) * Nc2 + ocol;
                    vstore(cA + o, a);
                    vstore(cH + o, hh);
                    vstore(cV + o, vv);
                    vstore(cD + o, dd);
                }
            }
        });
    };
    int q0 = 0;
    if (q0 + HLEN <= rows) {
        // ---- steady state: hand-counted pipeline (see asm_load above) ----
        // per output row: wait(2 rows) . 2 row passes . 2 loads (one body ahead) . column pass . 4 stores
        // => VMEM instructions between a row's load and its use one body later:
        constexpr int kAfter = 4 + 6 * (HLEN / 2 - 1);
        using VOUT = typename std::conditional<P == 2, v2f, float>::type;
        // invalid (halo / overhanging) lanes store to a private trash slot and never advance
        float* const tr = trash + (size_t)(blockIdx.x & trash_mask) * 1024 + (threadIdx.x >> 6) * 256 + lane * P;
        const size_t ostep = valid ? (size_t)Nc2 : 0;
        float* pA = valid ? cA + (size_t)y0 * Nc2 + ocol : tr;
        float* pH = valid ? cH + (size_t)y0 * Nc2 + ocol : tr + 64 * P;
        float* pV = valid ? cV + (size_t)y0 * Nc2 + ocol : tr;
        float* pD = valid ? cD + (size_t)y0 * Nc2 + ocol : tr + 64 * P;
        const float* const lbase = in + xo;
        static_for<HLEN>([&](auto K) { asm_drain1(v[decltype(K)::value]); });
        for (; q0 + HLEN <= rows; q0 += HLEN / 2) {
            const int rnext = 2 * q0 + 2 * HLEN - 2;
            static_for<HLEN / 2>([&](auto U) {
                constexpr int u = decltype(U)::value;
                constexpr int s0 = (2 * u + HLEN - 2) % HLEN, s1 = (2 * u + HLEN - 1) % HLEN;
                asm_wait2<kAfter>(v[2 * u], v[2 * u + 1]);
                row_pass(v[2 * u], ring[s0]);
                row_pass(v[2 * u + 1], ring[s1]);
                asm_load(v[2 * u], lbase + (size_t)wrap1(yb + rnext + 2 * u, Nr) * Nc);
                asm_load(v[2 * u + 1], lbase + (size_t)wrap1(yb + rnext + 2 * u + 1, Nr) * Nc);
                v2f ah[P], vd[P];
#pragma unroll
                for (int p = 0; p < P; p++) ah[p] = vd[p] = v2f{0.f, 0.f};
                static_for<HLEN>([&](auto J) {
                    constexpr int j = decltype(J)::value;
                    constexpr int s = (2 * u + j) % HLEN;
                    const v2f t = f.t[HLEN - 1 - j];
#pragma unroll
                    for (int p = 0; p < P; p++) {
                        ah[p] = pk_fma(splat(ring[s][p].x), t, ah[p]);
                        vd[p] = pk_fma(splat(ring[s][p].y), t, vd[p]);
                    }
                });
                const size_t ro = (size_t)(q0 + u) * ostep;
                if constexpr (P == 2) {
                    asm_store(pA + ro, VOUT{ah[0].x, ah[1].x});
                    asm_store(pH + ro, VOUT{ah[0].y, ah[1].y});
                    asm_store(pV + ro, VOUT{vd[0].x, vd[1].x});
                    asm_store(pD + ro, VOUT{vd[0].y, vd[1].y});
                } else {
                    asm_store(pA + ro, ah[0].x);
                    asm_store(pH + ro, ah[0].y);
                    asm_store(pV + ro, vd[0].x);
                    asm_store(pD + ro, vd[0].y);
                }
            });
        }
        static_for<HLEN>([&](auto K) { asm_drain1(v[decltype(K)::value]); });
    }
    for (; q0 < rows; q0 += HLEN / 2) body(std::false_type{}, q0);
}

// =================================================================================================
// inverse
// =================================================================================================
template <int HLEN>
struct InvGeom {
    static constexpr int H2 = HLEN / 2;
    static constexpr int C = H2 / 2;
    static constexpr int SHIFT = (H2 & 1) ? 0 : 1;
    static constexpr int PC = 2;                                 // coefficient columns per lane
    static constexpr int NB = C > 0 ? (C + PC - 1) / PC : 0;     // halo lanes per side
    static constexpr int WIN = PC * (2 * NB + 1);
    static constexpr int NROWS_EXTRA = H2 - 1 + SHIFT;           // coefficient rows streamed beyond the chunk's own
    static constexpr int MAXVL = 64 - 2 * NB;
};

template <int HLEN>
__global__ __launch_bounds__(256) void k_inv2d_stream(const float* __restrict__ cA, const float* __restrict__ cH,
                                                       const float* __restrict__ cV, const float* __restrict__ cD, float* __restrict__ out,
                                                       int Nri, int Nci, int RQ, int VL, ChunkMap cm, Taps2<float> f,
                                                       const StreamBatchI* __restrict__ batch)
{
    if (batch) {  // batched launch: image blockIdx.y
        const StreamBatchI e = batch[blockIdx.y];
        cA = e.cA;
        cH = e.cH;
        cV = e.cV;
        cD = e.cD;
        out = e.out;
    }
    using G = InvGeom<HLEN>;
    constexpr int H2 = G::H2, C = G::C, SHIFT = G::SHIFT, NB = G::NB, WIN = G::WIN;
    int cy, bx;
    if (!chunk_of_block(cm, cy, bx)) return;
    const int lane = threadIdx.x & 63;
    const int strip = bx * 4 + (threadIdx.x >> 6);
    const int col0 = strip * VL * 2;
    if (col0 >= Nci) return;
    const int cx = col0 + (lane - NB) * 2;
    const int cxw = wrapi(cx, Nci);
    const bool valid = (lane >= NB) && (lane < NB + VL) && (cx < Nci);
    const int Nco = 2 * Nci;
    const int y0 = cy * RQ;
    const int rowsq = min(RQ, Nri - y0);
    if (rowsq <= 0) return;
    const int yb = y0 - C;
    const int nrows = rowsq + G::NROWS_EXTRA;

    v2f ra[H2], rhh[H2], rv[H2], rd[H2];  // ring of the last H2 coefficient rows: (col0,col1) of each band

    auto load_row = [&](int r, float2& a, float2& h, float2& v, float2& d) {
        const size_t o = (size_t)wrap1(yb + r, Nri) * Nci + cxw;
        a = *reinterpret_cast<const float2*>(cA + o);
        h = *reinterpret_cast<const float2*>(cH + o);
        v = *reinterpret_cast<const float2*>(cV + o);
        d = *reinterpret_cast<const float2*>(cD + o);
    };
    auto put = [&](auto S, const float2& a, const float2& h, const float2& v, const float2& d) {
        constexpr int s = decltype(S)::value;
        ra[s] = v2f{a.x, a.y};
        rhh[s] = v2f{h.x, h.y};
        rv[s] = v2f{v.x, v.y};
        rd[s] = v2f{d.x, d.y};
    };

    // one output row: column synthesis from the ring window starting at slot S0 with tap parity OFF,
    // DPP exchange of (t1,t2), row synthesis, 16-byte store
    auto emit = [&](auto S0, auto OFF, int gy) {
        constexpr int s0 = decltype(S0)::value, off = decltype(OFF)::value;
        v2f sa = {0.f, 0.f}, sh = {0.f, 0.f}, sv = {0.f, 0.f}, sd = {0.f, 0.f};
        static_for<H2>([&](auto J) {
            constexpr int j = decltype(J)::value;
            constexpr int s = (s0 + j) % H2;
            constexpr int k = HLEN - 1 - (2 * j + off);
            // tap k as one half of the aligned pair it arrives in (pk_fma_sbcast): a splat per tap doubles the scalar registers of the two
            // banks, and from 12 taps on the compiler parks them in VGPR lanes (k_inv2d_stream<16>: 1374 v_readlane in 10 k instructions)
            constexpr int kp = k & ~1;
            const v2f pl2 = v2f{f.a[kp], f.a[kp + 1]}, ph2 = v2f{f.b[kp], f.b[kp + 1]};
            sa = pk_fma_sbcast<k & 1, false>(ra[s], pl2, sa);
            sh = pk_fma_sbcast<k & 1, false>(rhh[s], ph2, sh);
            sv = pk_fma_sbcast<k & 1, false>(rv[s], pl2, sv);
            sd = pk_fma_sbcast<k & 1, false>(rd[s], ph2, sd);
        });
        const v2f t1o = sa + sh, t2o = sv + sd;
        float t1[WIN], t2[WIN];
        t1[NB * 2] = t1o.x;
        t1[NB * 2 + 1] = t1o.y;
        t2[NB * 2] = t2o.x;
        t2[NB * 2 + 1] = t2o.y;
#pragma unroll
        for (int k = 0; k < NB; k++) {
            const int dl = (NB - 1 - k) * 2, sl = (NB - k) * 2, dr = (NB + 1 + k) * 2, sr = (NB + k) * 2;
#pragma unroll
            for (int cc = 0; cc < 2; cc++) {
                t1[dl + cc] = dpp_shr1(t1[sl + cc]);
                t2[dl + cc] = dpp_shr1(t2[sl + cc]);
                t1[dr + cc] = dpp_shl1(t1[sr + cc]);
                t2[dr + cc] = dpp_shl1(t2[sr + cc]);
            }
        }
        // output e of this lane: window start pl_e - C with pl_e = (e+SHIFT)>>1, tap parity 1-((e+SHIFT)&1).
        // Outputs that share a window (SHIFT=0: (0,1),(2,3); SHIFT=1: (1,2)) go through one packed chain
        // with the adjacent-tap pair (F[m-1], F[m]), m = HLEN-1-2j.
        float o4[4];
        auto pair_out = [&](auto E0) {
            constexpr int e0 = decltype(E0)::value;          // parity-1 output; e0+1 is the parity-0 one
            constexpr int pl = (e0 + SHIFT) >> 1;
            v2f s1 = {0.f, 0.f}, s2 = {0.f, 0.f};
#pragma unroll
            for (int j = 0; j < H2; j++) {
                const int m = HLEN - 1 - 2 * j;
                s1 = pk_fma(splat(t1[NB * 2 + pl - C + j]), v2f{f.a[m - 1], f.a[m]}, s1);
                s2 = pk_fma(splat(t2[NB * 2 + pl - C + j]), v2f{f.b[m - 1], f.b[m]}, s2);
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
                s1 = __builtin_fmaf(t1[NB * 2 + pl - C + j], f.a[k], s1);
                s2 = __builtin_fmaf(t2[NB * 2 + pl - C + j], f.b[k], s2);
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
        if (valid) *reinterpret_cast<float4*>(out + (size_t)gy * Nco + 2 * cx) = make_float4(o4[0], o4[1], o4[2], o4[3]);
    };

    // issue the ring prologue rows 0..H2-2 and the first body's rows together
    float2 a[H2], h[H2], v[H2], d[H2];
    {
        float2 pa[H2 > 1 ? H2 - 1 : 1], ph[H2 > 1 ? H2 - 1 : 1], pv[H2 > 1 ? H2 - 1 : 1], pd[H2 > 1 ? H2 - 1 : 1];
#pragma unroll
        for (int r = 0; r < H2 - 1; r++) load_row(r, pa[r], ph[r], pv[r], pd[r]);
#pragma unroll
        for (int u = 0; u < H2; u++)
            if (H2 - 1 + u < nrows) load_row(H2 - 1 + u, a[u], h[u], v[u], d[u]);
        static_for<H2 - 1>([&](auto Rr) {
            constexpr int r = decltype(Rr)::value;
            put(Rr, pa[r], ph[r], pv[r], pd[r]);
        });
    }

    // coefficient row H2-1+t completes the ring window of chunk-local rows t .. t+H2-1 (slots (u+j)%H2) and
    // yields the two output rows that use it:  SHIFT=1: 2t-1 (tap parity 1), 2t (parity 0);  SHIFT=0: 2t, 2t+1
    const int nsteady = nrows - (H2 - 1);
    auto body = [&](auto FAST, int t0) {
        constexpr bool fast = decltype(FAST)::value;
        static_for<H2>([&](auto U) {
            constexpr int u = decltype(U)::value;
            const int t = t0 + u;
            if (fast || t < nsteady) {
                put(std::integral_constant<int, (H2 - 1 + u) % H2>{}, a[u], h[u], v[u], d[u]);
                if (fast || H2 - 1 + t + H2 < nrows) load_row(H2 - 1 + t + H2, a[u], h[u], v[u], d[u]);  // one body ahead
                const int g1 = 2 * t - SHIFT, g0 = g1 + 1;
                if (fast) {  // interior rows: both outputs exist
                    emit(std::integral_constant<int, u % H2>{}, std::integral_constant<int, 1>{}, 2 * y0 + g1);
                    emit(std::integral_constant<int, u % H2>{}, std::integral_constant<int, 0>{}, 2 * y0 + g0);
                } else {
                    if (g1 >= 0 && g1 < 2 * rowsq) emit(std::integral_constant<int, u % H2>{}, std::integral_constant<int, 1>{}, 2 * y0 + g1);
                    if (g0 < 2 * rowsq) emit(std::integral_constant<int, u % H2>{}, std::integral_constant<int, 0>{}, 2 * y0 + g0);
                }
            }
        });
    };
    // t = 0 (g1 = -SHIFT may not exist) and the last rows go through the guarded path
    int t0 = 0;
    body(std::false_type{}, t0);
    t0 += H2;
    if (t0 + 2 * H2 <= nsteady - 1) {  // peeled first fast body: see the forward kernel
        body(std::true_type{}, t0);
        t0 += H2;
        for (; t0 + 2 * H2 <= nsteady - 1; t0 += H2) body(std::true_type{}, t0);
    }
    for (; t0 < nsteady; t0 += H2) body(std::false_type{}, t0);
}

// =================================================================================================
// host dispatch
// =================================================================================================
bool stream_enabled() { return knob(KN_STREAM) == 1 && counted_waits_ok(); }  // (also gates the cascade kernels: dwt_casc*.hip)

// rows of output (forward) / coefficient rows (inverse) per wave: enough waves to fill the chip
// (256 CUs x 4 SIMDs x a few waves), chunks tall enough to amortise the halo rows
static int pick_rows(long long nrows_total, int strips, int unit)
{
    int R = knob(KN_STREAM_R);
    if (R <= 0) {
        const long long target_waves = knob(KN_STREAM_WAVES);
        R = (int)(((long long)nrows_total * strips) / target_waves);
        if (R > 64) R = 64;
        if (R < 2) R = 2;
    }
    (void)unit;
    return R;
}

#define PDWT_CHECK_LAUNCH() PDWT_HIP_TRY(hipGetLastError())

template <int HLEN, int NIN>
static int launch_fwd_n(const float* in, float* cA, float* cH, float* cV, float* cD, float* trash, int trash_mask, int nr, int nc,
                        const Taps2<float>& f2, const StreamBatchF* batch = nullptr, int nimg = 1)
{
    TapsLH f;
    for (int k = 0; k < PDWT_MAX_FILTER_WIDTH; k++) f.t[k] = v2f{f2.a[k], f2.b[k]};
    constexpr int MAXVL = FwdGeom<HLEN, NIN>::MAXVL;
    const int strips = idiv_up(nc, MAXVL * NIN);
    const int VL = idiv_up(nc / NIN, strips);
    const int R = std::min(pick_rows((long long)(nr / 2) * nimg, strips, HLEN / 2), std::max(2, nr / 2));
    dim3 grid;
    const ChunkMap cm = make_map(idiv_up(strips, 4), idiv_up(nr / 2, R), &grid);
    grid.y = (unsigned)nimg;
    KTimer kt(K_FWD2D_STREAM, true);
    PDWT_LAUNCH_KT(kt, (k_fwd2d_stream<HLEN, NIN>), grid, dim3(256), 0, in, cA, cH, cV, cD, nr, nc, R, VL, trash, trash_mask, cm, f, batch);
    PDWT_CHECK_LAUNCH();
    return PDWT_OK;
}
// wide lanes (16-byte loads) for the big levels, narrow lanes (8-byte loads, twice the waves, half the serial
// work per wave) once a level is too small to fill the chip with wide ones
template <int HLEN>
static int launch_fwd(const float* in, float* cA, float* cH, float* cV, float* cD, float* trash, int trash_mask, int nr, int nc,
                      const Taps2<float>& f, const StreamBatchF* batch = nullptr, int nimg = 1)
{
    const long long narrow_below = knob(KN_STREAM_NARROW);
    if constexpr (HLEN > 10) {  // the wide form would need 2*HLEN row + 2*HLEN ring register pairs: narrow lanes only
        return launch_fwd_n<HLEN, 2>(in, cA, cH, cV, cD, trash, trash_mask, nr, nc, f, batch, nimg);
    } else {
        if ((long long)nr * nc * nimg <= narrow_below) return launch_fwd_n<HLEN, 2>(in, cA, cH, cV, cD, trash, trash_mask, nr, nc, f, batch, nimg);
        return launch_fwd_n<HLEN, 4>(in, cA, cH, cV, cD, trash, trash_mask, nr, nc, f, batch, nimg);
    }
}

template <int HLEN>
static int launch_inv(const float* cA, const float* cH, const float* cV, const float* cD, float* out, int nri, int nci, const Taps2<float>& f,
                      const StreamBatchI* batch = nullptr, int nimg = 1)
{
    constexpr int MAXVL = InvGeom<HLEN>::MAXVL;
    const int strips = idiv_up(nci, MAXVL * 2);
    const int VL = idiv_up(nci / 2, strips);
    const int RQ = std::min(pick_rows((long long)nri * nimg, strips, InvGeom<HLEN>::H2), std::max(2, nri));
    dim3 grid;
    const ChunkMap cm = make_map(idiv_up(strips, 4), idiv_up(nri, RQ), &grid);
    grid.y = (unsigned)nimg;
    KTimer kt(K_INV2D_STREAM, true);
    PDWT_LAUNCH_KT(kt, k_inv2d_stream<HLEN>, grid, dim3(256), 0, cA, cH, cV, cD, out, nri, nci, RQ, VL, cm, f, batch);
    PDWT_CHECK_LAUNCH();
    return PDWT_OK;
}

// filter lengths with a streaming instantiation (register budget: one unrolled body holds HLEN rows in flight)
#define PDWT_STREAM_FWD_HLENS(X) X(4) X(6) X(8) X(10) X(12) X(14) X(16)
#define PDWT_STREAM_INV_HLENS(X) X(4) X(6) X(8) X(10) X(12) X(14) X(16)

int fwd2d_stream_f32(const float* in, float* cA, float* cH, float* cV, float* cD, float* trash, size_t trash_floats, int nr, int nc, int hlen,
                     const Taps2<float>& f)
{
    if (!stream_enabled() || !trash || trash_floats < 1024) return 1;
    // trash slots of 1024 floats, one per workgroup modulo a power of two (any overlap is harmless: nobody reads them)
    int slots = 1;
    while (slots < 256 && (size_t)slots * 2 * 1024 <= trash_floats) slots *= 2;
    const int trash_mask = slots - 1;
    if ((nr & 1) || (nc & 3) || nc < 64 || nr < 2 * hlen) return 1;
    if (!al16(in) || !al16(cA) || !al16(cH) || !al16(cV) || !al16(cD)) return 1;
    switch (hlen) {
#define X(H) \
    case H: return launch_fwd<H>(in, cA, cH, cV, cD, trash, trash_mask, nr, nc, f);
        PDWT_STREAM_FWD_HLENS(X)
#undef X
        default: return 1;
    }
}

int inv2d_stream_f32(const float* cA, const float* cH, const float* cV, const float* cD, float* out, int nri, int nci, int nro, int nco,
                     int hlen, const Taps2<float>& f)
{
    if (!stream_enabled()) return 1;
    if ((nci & 1) || nco != 2 * nci || nro != 2 * nri || nci < 32 || nri < 2 * hlen) return 1;
    if (!al16(out) || !al16(cA) || !al16(cH) || !al16(cV) || !al16(cD)) return 1;
    switch (hlen) {
#define X(H) \
    case H: return launch_inv<H>(cA, cH, cV, cD, out, nri, nci, f);
        PDWT_STREAM_INV_HLENS(X)
#undef X
        default: return 1;
    }
}


bool fwd2d_stream_takes(int nr, int nc, int hlen)
{
    if (!stream_enabled() || (nr & 1) || (nc & 3) || nc < 64 || nr < 2 * hlen) return false;
    switch (hlen) {
#define X(H) case H:
        PDWT_STREAM_FWD_HLENS(X)
#undef X
        return true;
        default: return false;
    }
}
bool inv2d_stream_takes(int nri, int nci, int hlen)
{
    if (!stream_enabled() || (nci & 1) || nci < 32 || nri < 2 * hlen) return false;
    switch (hlen) {
#define X(H) case H:
        PDWT_STREAM_INV_HLENS(X)
#undef X
        return true;
        default: return false;
    }
}

int fwd2d_stream_batch_f32(const StreamBatchF* d_tab, int nimg, size_t trash_floats, int nr, int nc, int hlen, const Taps2<float>& f)
{
    if (!d_tab || nimg < 1 || nimg > 65535 || trash_floats < 1024 || !fwd2d_stream_takes(nr, nc, hlen)) return 1;
    int slots = 1;
    while (slots < 256 && (size_t)slots * 2 * 1024 <= trash_floats) slots *= 2;
    switch (hlen) {
#define X(H) \
    case H: return launch_fwd<H>(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, slots - 1, nr, nc, f, d_tab, nimg);
        PDWT_STREAM_FWD_HLENS(X)
#undef X
        default: return 1;
    }
}

int inv2d_stream_batch_f32(const StreamBatchI* d_tab, int nimg, int nri, int nci, int hlen, const Taps2<float>& f)
{
    if (!d_tab || nimg < 1 || nimg > 65535 || !inv2d_stream_takes(nri, nci, hlen)) return 1;
    switch (hlen) {
#define X(H) \
    case H: return launch_inv<H>(nullptr, nullptr, nullptr, nullptr, nullptr, nri, nci, f, d_tab, nimg);
        PDWT_STREAM_INV_HLENS(X)
#undef X
        default: return 1;
    }
}

}  // namespace pdwt
