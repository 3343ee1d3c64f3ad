// rows_tr.hip -- row analysis / synthesis for LONG double-precision filter banks (config C5: db20, hlen 40).
//
// k_ana_rows / k_syn_rows (dwt.hip) read one LDS value per two FMAs (49 % of the LDS cycles are bank conflicts of the
// stride-2 window) and keep 2*hlen taps in SGPRs that do not exist (160 for db20 in double; hipcc parks them in VGPR
// lanes: 2.5 v_readlane per FMA).  Here
//   * a lane computes PO adjacent outputs from ONE register window (2*PO + hlen - 2 samples, read as aligned
//     16-byte LDS slots through an XOR swizzle that makes the 64-byte lane stride conflict-free): 0.07 LDS reads
//     per FMA instead of 0.5;
//   * the taps live in a "tap register" (lane k = tap k), broadcast with v_readlane tap by tap (tapreg.hpp):
//     4 v_readlane per 2*PO FMAs, no SGPR pressure.
// Each accumulator sums its taps in ascending order with one FMA per tap: bit-identical to k_ana_rows / k_syn_rows
// and to the oracle.  Reference code replaced: w_kern_forward_pass1, w_kern_inverse_pass2 (src/separable.cu:91-131,
// 293-328) for the two-pass 2-D path of long filters.
#include "rows_tr.hpp"

#include <type_traits>

#include "tapreg.hpp"

namespace pdwt {

typedef double v2d __attribute__((ext_vector_type(2)));

template <int I, int N, typename F>
__device__ __forceinline__ void rfor_impl(F&& fn)
{
    if constexpr (I < N) {
        fn(std::integral_constant<int, I>{});
        rfor_impl<I + 1, N>(fn);
    }
}
template <int N, typename F>
__device__ __forceinline__ void rfor(F&& fn) { rfor_impl<0, N>(fn); }

// 16-byte slot swizzle: a lane's window starts 4 slots (64 bytes) after its neighbour's, so the 16 lanes of one
// ds_read_b128 pass hit only 4 distinct bank groups; XOR-ing the low two slot bits with bits 4-5 spreads them over all 16
__device__ __forceinline__ int rswz(int slot) { return slot ^ ((slot >> 4) & 3); }

constexpr int kRtTXO = 256;  // outputs per wave-row tile: 64 lanes x 4

// -------------------------------------------------------------------------------------------------
// analysis: in (Nr x Nc) -> lo, hi (Nr x ceil(Nc/2)); block = 4 rows (one per wave) x 256 outputs
// -------------------------------------------------------------------------------------------------
template <int HLEN>
__global__ __launch_bounds__(256) void k_ana_rows_tr(const double* __restrict__ in, double* __restrict__ lo, double* __restrict__ hi, int Nr,
                                                      int Nc, Taps2<double> f)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int PO = 4;
    constexpr int C = HLEN / 2 - 1;
    constexpr int CIN = 2 * kRtTXO + HLEN - 2;     // samples a wave-row tile needs
    constexpr int NSLOT = (CIN + 1) / 2 + 4;        // 16-byte slots (+ swizzle slack)
    constexpr int WIN = 2 * PO + HLEN - 2;          // window samples per lane
    constexpr int NRD = (WIN + 1) / 2;              // 16-byte reads per lane
    static_assert(HLEN <= 64 && (HLEN % 2) == 0, "one tap per lane, even lengths");
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    v2d* const s = reinterpret_cast<v2d*>(smem_raw) + w * NSLOT;
    double* const se = reinterpret_cast<double*>(s);
    const int Nc2 = div2(Nc);
    const int x0 = blockIdx.x * kRtTXO;
    const int xb = 2 * x0 - C;
    const int row = blockIdx.y * 4 + w;
    if (row >= Nr) return;  // (no workgroup barrier below: waves are independent)
    const double* src = in + (size_t)row * Nc;
    // stage the tile: all loads of a lane issued before the first LDS write (index clamped, not predicated); the
    // wrap arithmetic (an integer modulo per sample) only on the tiles that touch an edge
    constexpr int NLD = (CIN + 63) / 64;
    double sv[NLD];
    if (xb >= 0 && xb + CIN <= Nc) {
#pragma unroll
        for (int k = 0; k < NLD; k++) sv[k] = src[xb + min(lane + 64 * k, CIN - 1)];
    } else {
#pragma unroll
        for (int k = 0; k < NLD; k++) sv[k] = src[wrap_ext(xb + min(lane + 64 * k, CIN - 1), Nc)];
    }
#pragma unroll
    for (int k = 0; k < NLD; k++) {
        const int cc = lane + 64 * k;
        if (cc < CIN) se[2 * rswz(cc >> 1) + (cc & 1)] = sv[k];
    }
    double tapA = f.a[min(lane, HLEN - 1)], tapB = f.b[min(lane, HLEN - 1)];
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    double wv[2 * NRD];
    rfor<NRD>([&](auto K) {
        constexpr int k = decltype(K)::value;
        const v2d t = s[rswz(PO * lane + k)];  // sample 2*PO*lane + 2k, +1
        wv[2 * k] = t.x;
        wv[2 * k + 1] = t.y;
    });
    double acc[8];
#pragma unroll
    for (int q = 0; q < 8; q++) acc[q] = 0.0;
    rfor<HLEN>([&](auto J) {
        constexpr int j = decltype(J)::value;
        tap_order(tapA, tapB, acc);
        const double ta = lane_bcast(tapA, HLEN - 1 - j), tb = lane_bcast(tapB, HLEN - 1 - j);
#pragma unroll
        for (int q = 0; q < PO; q++) {
            acc[2 * q] = fma_t(wv[2 * q + j], ta, acc[2 * q]);
            acc[2 * q + 1] = fma_t(wv[2 * q + j], tb, acc[2 * q + 1]);
        }
    });
    const int gx = x0 + PO * lane;
    double* plo = lo + (size_t)row * Nc2 + gx;
    double* phi = hi + (size_t)row * Nc2 + gx;
    if (gx + PO <= Nc2 && ((Nc2 & 1) == 0) && (((uintptr_t)lo | (uintptr_t)hi) & 15) == 0) {
        reinterpret_cast<v2d*>(plo)[0] = v2d{acc[0], acc[2]};
        reinterpret_cast<v2d*>(plo)[1] = v2d{acc[4], acc[6]};
        reinterpret_cast<v2d*>(phi)[0] = v2d{acc[1], acc[3]};
        reinterpret_cast<v2d*>(phi)[1] = v2d{acc[5], acc[7]};
    } else {
#pragma unroll
        for (int q = 0; q < PO; q++)
            if (gx + q < Nc2) {
                plo[q] = acc[2 * q];
                phi[q] = acc[2 * q + 1];
            }
    }
}

// -------------------------------------------------------------------------------------------------
// synthesis: a, d (Nr x Nci) -> out (Nr x Nco), out = a * IL + d * IH (SURVEY A-2); block = 4 rows x 256 coefficient
// positions (512 outputs).  A lane owns 8 adjacent outputs = 4 coefficient positions: one register window of
// 4 + hlen/2 coefficients per band; the outputs of one tap parity share every tap (8 FMAs per tap pair).
// -------------------------------------------------------------------------------------------------
constexpr int kRtTXC = 256;
__device__ __forceinline__ int rswz2(int slot) { return slot ^ ((slot >> 4) & 1); }  // lane stride = 2 slots: 2-way conflict otherwise

template <int HLEN>
__global__ __launch_bounds__(256) void k_syn_rows_tr(const double* __restrict__ a, const double* __restrict__ d, double* __restrict__ out,
                                                      int Nr, int Nci, int Nco, Taps2<double> f)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int H2 = HLEN / 2, C = H2 / 2, SHIFT = (H2 & 1) ? 0 : 1;
    constexpr int CC = kRtTXC + H2 + 1;            // coefficients a wave-row tile needs (window of the last lane included)
    constexpr int NSLOT = (CC + 1) / 2 + 2;         // 16-byte slots per band (+ swizzle slack)
    constexpr int WIN = 4 + H2 + 1;                 // window coefficients per lane and band (positions 4l .. 4l+4+H2)
    constexpr int NRD = (WIN + 1) / 2;
    static_assert(HLEN <= 64 && (HLEN % 2) == 0, "one tap per lane, even lengths");
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    v2d* const sa = reinterpret_cast<v2d*>(smem_raw) + w * 2 * NSLOT;
    v2d* const sd = sa + NSLOT;
    double* const sae = reinterpret_cast<double*>(sa);
    double* const sde = reinterpret_cast<double*>(sd);
    const int x0 = blockIdx.x * kRtTXC;  // first coefficient position of the tile
    const int xb = x0 - C;
    const int row = blockIdx.y * 4 + w;
    if (row >= Nr) return;
    const double* pa = a + (size_t)row * Nci;
    const double* pd = d + (size_t)row * Nci;
    constexpr int NLD = (CC + 63) / 64;
    double va[NLD], vd[NLD];
    if (xb >= 0 && xb + CC <= Nci) {
#pragma unroll
        for (int k = 0; k < NLD; k++) {
            const int sx = xb + min(lane + 64 * k, CC - 1);
            va[k] = pa[sx];
            vd[k] = pd[sx];
        }
    } else {
#pragma unroll
        for (int k = 0; k < NLD; k++) {
            const int sx = wrap_per(xb + min(lane + 64 * k, CC - 1), Nci);
            va[k] = pa[sx];
            vd[k] = pd[sx];
        }
    }
#pragma unroll
    for (int k = 0; k < NLD; k++) {
        const int cc = lane + 64 * k;
        if (cc < CC) {
            sae[2 * rswz2(cc >> 1) + (cc & 1)] = va[k];
            sde[2 * rswz2(cc >> 1) + (cc & 1)] = vd[k];
        }
    }
    double tapA = f.a[min(lane, HLEN - 1)], tapB = f.b[min(lane, HLEN - 1)];
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    double wa[2 * NRD], wd[2 * NRD];  // coefficients xb + 4*lane + m, m = 0 .. WIN-1
    rfor<NRD>([&](auto K) {
        constexpr int k = decltype(K)::value;
        const v2d ta = sa[rswz2(2 * lane + k)], td = sd[rswz2(2 * lane + k)];
        wa[2 * k] = ta.x;
        wa[2 * k + 1] = ta.y;
        wd[2 * k] = td.x;
        wd[2 * k + 1] = td.y;
    });
    // output e (0..7) of the lane: gp = e + SHIFT, window start pl = gp >> 1, tap parity off = 1 - (gp & 1).
    // acc[par][2i], acc[par][2i+1]: a- and d-branch sums of the i-th output of parity class par (par = off).
    double acc[2][8];
#pragma unroll
    for (int p = 0; p < 2; p++)
#pragma unroll
        for (int q = 0; q < 8; q++) acc[p][q] = 0.0;
    rfor<H2>([&](auto J) {
        constexpr int j = decltype(J)::value;
        rfor<2>([&](auto P) {
            constexpr int off = 1 - decltype(P)::value;  // parity 1 first, as k_syn_rows iterates e = 0, 1 with SHIFT folded in
            constexpr int k = HLEN - 1 - (2 * j + off);
            tap_order(tapA, tapB, acc[off]);
            const double ta = lane_bcast(tapA, k), tb = lane_bcast(tapB, k);
            rfor<4>([&](auto I) {
                constexpr int i = decltype(I)::value;
                constexpr int e = 2 * i + ((1 - off + SHIFT) & 1);  // the i-th output whose tap parity 1 - ((e + SHIFT) & 1) is `off`
                constexpr int pl = (e + SHIFT) >> 1;
                acc[off][2 * i] = fma_t(wa[pl + j], ta, acc[off][2 * i]);
                acc[off][2 * i + 1] = fma_t(wd[pl + j], tb, acc[off][2 * i + 1]);
            });
        });
    });
    double res[8];
    rfor<8>([&](auto E) {
        constexpr int e = decltype(E)::value;
        constexpr int off = 1 - ((e + SHIFT) & 1);
        constexpr int i = e / 2;
        res[e] = acc[off][2 * i] + acc[off][2 * i + 1];
    });
    const int gx = 2 * x0 + 8 * lane;
    double* po = out + (size_t)row * Nco + gx;
    if (gx + 8 <= Nco && ((Nco & 1) == 0) && ((uintptr_t)out & 15) == 0) {
#pragma unroll
        for (int q = 0; q < 4; q++) reinterpret_cast<v2d*>(po)[q] = v2d{res[2 * q], res[2 * q + 1]};
    } else {
#pragma unroll
        for (int q = 0; q < 8; q++)
            if (gx + q < Nco) po[q] = res[q];
    }
}

#define PDWT_CHECK_LAUNCH() PDWT_HIP_TRY(hipGetLastError())
#define PDWT_ROWS_TR_HLENS(X) X(20) X(24) X(30) X(40)

template <int HLEN>
static int launch_ana_tr(const double* in, double* lo, double* hi, int Nr, int Nc, const Taps2<double>& f)
{
    constexpr int CIN = 2 * kRtTXO + HLEN - 2;
    constexpr int NSLOT = (CIN + 1) / 2 + 4;
    const size_t lds = 4 * (size_t)NSLOT * 16;
    dim3 grid(idiv_up(div2(Nc), kRtTXO), idiv_up(Nr, 4));
    KTimer kt(K_ANA_ROWS);
    hipLaunchKernelGGL(k_ana_rows_tr<HLEN>, grid, dim3(256), lds, stream(), in, lo, hi, Nr, Nc, f);
    PDWT_CHECK_LAUNCH();
    return PDWT_OK;
}

int ana_rows_tr_f64(const double* in, double* lo, double* hi, int Nr, int Nc, int hlen, const Taps2<double>& f)
{
    const int enabled = knob(KN_ROWS_TR);
    if (!enabled || Nc < 2 * hlen) return 1;
    switch (hlen) {
#define X(H) \
    case H: return launch_ana_tr<H>(in, lo, hi, Nr, Nc, f);
        PDWT_ROWS_TR_HLENS(X)
#undef X
        default: return 1;
    }
}

template <int HLEN>
static int launch_syn_tr(const double* a, const double* d, double* out, int Nr, int Nci, int Nco, const Taps2<double>& f)
{
    constexpr int CC = kRtTXC + HLEN / 2 + 1;
    constexpr int NSLOT = (CC + 1) / 2 + 2;
    const size_t lds = 4 * 2 * (size_t)NSLOT * 16;
    dim3 grid(idiv_up(Nci, kRtTXC), idiv_up(Nr, 4));
    KTimer kt(K_SYN_ROWS);
    hipLaunchKernelGGL(k_syn_rows_tr<HLEN>, grid, dim3(256), lds, stream(), a, d, out, Nr, Nci, Nco, f);
    PDWT_CHECK_LAUNCH();
    return PDWT_OK;
}

int syn_rows_tr_f64(const double* a, const double* d, double* out, int Nr, int Nci, int Nco, int hlen, const Taps2<double>& f)
{
    const int enabled = knob(KN_ROWS_TR);
    if (!enabled || Nci < hlen) return 1;
    switch (hlen) {
#define X(H) \
    case H: return launch_syn_tr<H>(a, d, out, Nr, Nci, Nco, f);
        PDWT_ROWS_TR_HLENS(X)
#undef X
        default: return 1;
    }
}

}  // namespace pdwt
