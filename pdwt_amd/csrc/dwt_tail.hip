// dwt_tail.hip -- the SMALL levels of a multi-level 2D DWT fused into one launch per direction (float32).
//
// After level 1 the remaining levels of config C2 move 31 % of the bytes but cost 4 of the 6 launches of a
// forward+inverse pair, each one latency-bound (launch ramp + one dependent memory round trip + drain:
// 5-10 us for 8-32 MB, see profiles/).  These kernels run levels 2..L (forward: "tail") and L..2 (inverse:
// "head") inside ONE workgroup-tiled launch: a workgroup owns a T x T block of the DEEPEST level and
// recomputes the halo it needs of the intermediate approximations in LDS (overlap-tile), so no
// approximation of an intermediate level ever goes to memory and there is no inter-workgroup dependency.
//   forward tail: stage the level-1 approximation region R0 x R0 (periodic wrap), then per level
//                 row pass -> LDS, column pass -> details of the owned block to HBM, approximation region -> LDS.
//   inverse head: load the deepest approximation + per-level detail regions, per level column synthesis ->
//                 row synthesis -> approximation region of the next finer level in LDS; the last one is the
//                 owned block of the level-1 approximation, written to HBM.
// Region recurrences (h = hlen, c = h/2-1, h2 = h/2, c2 = h2/2, shift = h2 even):
//   forward : n_in = 2*n_out + h - 2, owned offset off_in = 2*off_out + c        (SURVEY A-1)
//   inverse : first coefficient p0 = (g0 + shift)/2 - c2, count = (g_last + shift)/2 - c2 + h2 - p0  (A-2)
// STATUS (round 1): correct and bit-identical, but measured slower on MI355X than the per-level streaming
// launches it replaces (4096^2 db4 L3: tail 26 us vs 13.7 us for levels 2+3 forward, head 36 us vs 17.3 us
// inverse) -- the overlap-tile recompute (x2.6 at level 2) plus scalar LDS traffic cost more than the two
// launch latencies saved.  It is therefore OPT-IN (PDWT_TAIL=1 or pdwt_debug_set("tail", 1)); the default
// path is per-level.  Kept as the starting point for a register-blocked version.
// Arithmetic order per sample is the reference's (row pass then column pass forward; column then row
// synthesis inverse; taps ascending, one FMA each) => bit-identical to the per-level kernels and the oracle.
#include <stdlib.h>

#include "common.hpp"
#include "dwt_tail.hpp"

namespace pdwt {

constexpr int kTailMaxLev = 4;

struct TailGeom {
    int nlev;                    // levels handled by the launch (2..kTailMaxLev)
    int T;                       // owned block edge at the deepest level
    int Nr[kTailMaxLev + 1];     // Nr[k], Nc[k]: size of the INPUT of tail level k (k = 0: the level-1 approximation);
    int Nc[kTailMaxLev + 1];     //               [nlev] = size of the deepest approximation
};
struct TailBandsF {
    float* H[kTailMaxLev];  // detail bands of tail level k (size Nr[k+1] x Nc[k+1])
    float* V[kTailMaxLev];
    float* D[kTailMaxLev];
    float* A;               // deepest approximation
};

__device__ __forceinline__ void tail_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ int wrapm(int s, int n)
{
    s %= n;
    return s < 0 ? s + n : s;
}
// single conditional wrap (regions overhang a level by less than its size: the dispatcher checks it)
__device__ __forceinline__ int wrap1(int s, int n) { return s < 0 ? s + n : (s >= n ? s - n : s); }
// flat index -> (row, col) for a row length n without an integer division: (e + 0.5)/n is at least 0.5/n away
// from an integer, far above float rounding for the few-thousand-element regions used here
__device__ __forceinline__ int fdiv(int e, float inv_n) { return (int)(((float)e + 0.5f) * inv_n); }

// Stage an m x m region of a periodic Nr x Nc band into LDS (row stride m).  All of a thread's loads are issued
// before the first LDS write (index clamped, not predicated): one memory round trip per region instead of one
// per loop iteration.
template <int MAXIT>
__device__ __forceinline__ void stage_region(float* dst, const float* __restrict__ src, int m, int y0, int x0, int Nr, int Nc)
{
    const int n = m * m;
    const float inv = 1.0f / (float)m;
    float v[MAXIT];
#pragma unroll
    for (int it = 0; it < MAXIT; it++) {
        const int e = min((int)threadIdx.x + 256 * it, n - 1);
        const int r = fdiv(e, inv), cc = e - r * m;
        v[it] = src[(size_t)wrap1(y0 + r, Nr) * Nc + wrap1(x0 + cc, Nc)];
    }
#pragma unroll
    for (int it = 0; it < MAXIT; it++) {
        const int e = threadIdx.x + 256 * it;
        if (e < n) dst[e] = v[it];
    }
}

// -------------------------------------------------------------------------------------------------
// forward tail
// -------------------------------------------------------------------------------------------------
template <int HLEN>
__global__ __launch_bounds__(256) void k_fwd2d_tail(const float* __restrict__ in, TailBandsF b, TailGeom g, Taps2<float> f)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int C = HLEN / 2 - 1;
    const int L = g.nlev, T = g.T;
    // region sizes: n_out[L-1] = T, n_in[k] = 2 n_out[k] + HLEN - 2, n_out[k-1] = n_in[k]; owned offsets off[k]
    int n_out[kTailMaxLev], off[kTailMaxLev];
    n_out[L - 1] = T;
    off[L - 1] = 0;
    for (int k = L - 1; k > 0; k--) {
        n_out[k - 1] = 2 * n_out[k] + HLEN - 2;
        off[k - 1] = 2 * off[k] + C;
    }
    const int n_in0 = 2 * n_out[0] + HLEN - 2;
    float* s_in = reinterpret_cast<float*>(smem);      // n_in0 x n_in0 (later levels reuse it for their smaller input)
    float* s_lo = s_in + n_in0 * n_in0;                 // n_in x n_out
    float* s_hi = s_lo + n_in0 * n_out[0];
    float* s_nx = s_hi + n_in0 * n_out[0];              // n_out x n_out approximation of the level just computed

    // owned block of the deepest level and the global origin of every level's region
    const int ty0 = blockIdx.y * T, tx0 = blockIdx.x * T;
    const int own0 = 1 << (L - 1);  // owned edge at level k = T * 2^(L-1-k)
    // global (row, col) of region element (0,0) at the output of level k: own_start_k - off[k]
    // stage the input region: rows/cols 2*(own_start_0 - off[0]) - C ... in the index space of `in`
    {
        const int ys = wrapm(2 * (ty0 * own0 - off[0]) - C, g.Nr[0]), xs = wrapm(2 * (tx0 * own0 - off[0]) - C, g.Nc[0]);
        if (n_in0 * n_in0 <= 256 * 32) stage_region<32>(s_in, in, n_in0, ys, xs, g.Nr[0], g.Nc[0]);
        else stage_region<48>(s_in, in, n_in0, ys, xs, g.Nr[0], g.Nc[0]);
    }
    tail_barrier();

    int nin = n_in0;
    for (int k = 0; k < L; k++) {
        const int no = n_out[k];
        // row pass: (r, i) r < nin, i < no
        const float inv_no = 1.0f / (float)no;
        for (int e = threadIdx.x; e < nin * no; e += 256) {
            const int r = fdiv(e, inv_no), i = e - r * no;
            const float* p = s_in + r * nin + 2 * i;
            float lo = 0.f, hi = 0.f;
#pragma unroll
            for (int j = 0; j < HLEN; j++) {
                const float v = p[j];
                lo = __builtin_fmaf(v, f.a[HLEN - 1 - j], lo);
                hi = __builtin_fmaf(v, f.b[HLEN - 1 - j], hi);
            }
            s_lo[e] = lo;
            s_hi[e] = hi;
        }
        tail_barrier();
        // column pass: (y, x) y,x < no
        const int own = T << (L - 1 - k);
        const int gy0 = ty0 * (own / T) - off[k], gx0 = tx0 * (own / T) - off[k];  // global index of region (0,0)
        const int Nro = g.Nr[k + 1], Nco = g.Nc[k + 1];
        const bool last = (k == L - 1);
        for (int e = threadIdx.x; e < no * no; e += 256) {
            const int y = fdiv(e, inv_no), x = e - y * no;
            const float* pl = s_lo + (2 * y) * no + x;
            const float* ph = s_hi + (2 * y) * no + x;
            float a = 0.f, h = 0.f, v = 0.f, d = 0.f;
#pragma unroll
            for (int j = 0; j < HLEN; j++) {
                const float l = pl[j * no], hh = ph[j * no];
                const float fl = f.a[HLEN - 1 - j], fh = f.b[HLEN - 1 - j];
                a = __builtin_fmaf(l, fl, a);
                h = __builtin_fmaf(l, fh, h);
                v = __builtin_fmaf(hh, fl, v);
                d = __builtin_fmaf(hh, fh, d);
            }
            const int oy = y - off[k], ox = x - off[k];
            if (oy >= 0 && oy < own && ox >= 0 && ox < own) {  // owned block of this level: details (and the last A) go to HBM
                const int gy = gy0 + y, gx = gx0 + x;
                if (gy < Nro && gx < Nco) {
                    const size_t o = (size_t)gy * Nco + gx;
                    b.H[k][o] = h;
                    b.V[k][o] = v;
                    b.D[k][o] = d;
                    if (last) b.A[o] = a;
                }
            }
            if (!last) s_nx[e] = a;
        }
        tail_barrier();
        if (!last) {  // the approximation region becomes the next level's input
            for (int e = threadIdx.x; e < no * no; e += 256) s_in[e] = s_nx[e];
            tail_barrier();
            nin = no;
        }
    }
}

// -------------------------------------------------------------------------------------------------
// inverse head: levels nlev-1 .. 0 of the tail numbering (deepest first); output = the level-1 approximation
// (size Nr[0] x Nc[0]), owned block of edge T * 2^nlev per workgroup ... expressed through the same TailGeom:
// here T is the owned edge of the OUTPUT (finest) block.
// -------------------------------------------------------------------------------------------------
template <int HLEN>
__global__ __launch_bounds__(256) void k_inv2d_head(float* __restrict__ out, TailBandsF b, TailGeom g, Taps2<float> f)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int H2 = HLEN / 2, C2 = H2 / 2, SHIFT = (H2 & 1) ? 0 : 1;
    const int L = g.nlev, T = g.T;
    // Output region of step k (k = 0 finest .. L-1 deepest step): start s[k] (global index at that resolution),
    // count n[k].  Step k consumes coefficient region p0[k], m[k] at the next coarser resolution = output region of step k+1.
    int s_[kTailMaxLev + 1], n_[kTailMaxLev + 1];
    // 1-D recurrences are identical in y and x up to the block origin, so keep origins separate and counts shared
    int sy[kTailMaxLev + 1], sx[kTailMaxLev + 1];
    sy[0] = blockIdx.y * T;
    sx[0] = blockIdx.x * T;
    n_[0] = T;
    for (int k = 0; k < L; k++) {
        // coefficients needed for outputs [s, s+n): p from (s+SHIFT)/2 - C2 to (s+n-1+SHIFT)/2 - C2 + H2 - 1
        const int py = ((sy[k] + SHIFT) >> 1) - C2, px = ((sx[k] + SHIFT) >> 1) - C2;
        const int my = ((sy[k] + n_[k] - 1 + SHIFT) >> 1) - C2 + H2 - py;
        sy[k + 1] = py;
        sx[k + 1] = px;
        n_[k + 1] = my;  // block origins are multiples of T (even), so the count is the same along x
    }
    (void)s_;
    // LDS: approximation region (ping-pong), detail regions, t1/t2
    const int nmax = n_[1];                      // largest coefficient region (finest step)
    float* s_a = reinterpret_cast<float*>(smem);  // approximation input of the current step (<= nmax^2)
    float* s_h = s_a + nmax * nmax;
    float* s_v = s_h + nmax * nmax;
    float* s_d = s_v + nmax * nmax;
    float* s_t1 = s_d + nmax * nmax;             // (n_out) x (m) column-synthesis results
    float* s_t2 = s_t1 + n_[0] * nmax;
    float* s_o = s_t2 + n_[0] * nmax;            // output approximation of the current step (<= nmax^2; the finest goes to HBM)

    // deepest approximation region
    stage_region<8>(s_a, b.A, n_[L], wrapm(sy[L], g.Nr[L]), wrapm(sx[L], g.Nc[L]), g.Nr[L], g.Nc[L]);
    for (int k = L - 1; k >= 0; k--) {
        const int m = n_[k + 1], no = n_[k];
        const int Nr = g.Nr[k + 1], Nc = g.Nc[k + 1];
        {
            const int y0 = wrapm(sy[k + 1], Nr), x0 = wrapm(sx[k + 1], Nc);
            stage_region<8>(s_h, b.H[k], m, y0, x0, Nr, Nc);
            stage_region<8>(s_v, b.V[k], m, y0, x0, Nr, Nc);
            stage_region<8>(s_d, b.D[k], m, y0, x0, Nr, Nc);
        }
        tail_barrier();
        // column synthesis: output rows gy = sy[k] + yl, yl < no; all m coefficient columns
        const float inv_m = 1.0f / (float)m, inv_no = 1.0f / (float)no;
        for (int e = threadIdx.x; e < no * m; e += 256) {
            const int yl = fdiv(e, inv_m), cc = e - yl * m;
            const int gp = sy[k] + yl + SHIFT;
            const int pl = (gp >> 1) - C2 - sy[k + 1], off = 1 - (gp & 1);
            float sa = 0.f, sh = 0.f, sv = 0.f, sd = 0.f;
#pragma unroll
            for (int j = 0; j < H2; j++) {
                const int kk = HLEN - 1 - 2 * j - off;
                const float fl = f.a[kk], fh = f.b[kk];
                const int si = (pl + j) * m + cc;
                sa = __builtin_fmaf(s_a[si], fl, sa);
                sh = __builtin_fmaf(s_h[si], fh, sh);
                sv = __builtin_fmaf(s_v[si], fl, sv);
                sd = __builtin_fmaf(s_d[si], fh, sd);
            }
            s_t1[e] = sa + sh;
            s_t2[e] = sv + sd;
        }
        tail_barrier();
        // row synthesis: (yl, xl) -> out
        const bool last = (k == 0);
        const int Nro = g.Nr[k], Nco = g.Nc[k];
        for (int e = threadIdx.x; e < no * no; e += 256) {
            const int yl = fdiv(e, inv_no), xl = e - yl * no;
            const int gp = sx[k] + xl + SHIFT;
            const int pl = (gp >> 1) - C2 - sx[k + 1], off = 1 - (gp & 1);
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int j = 0; j < H2; j++) {
                const int kk = HLEN - 1 - 2 * j - off;
                s1 = __builtin_fmaf(s_t1[yl * m + pl + j], f.a[kk], s1);
                s2 = __builtin_fmaf(s_t2[yl * m + pl + j], f.b[kk], s2);
            }
            const float r = s1 + s2;
            if (last) {
                const int gy = sy[0] + yl, gx = sx[0] + xl;
                if (gy < Nro && gx < Nco) out[(size_t)gy * Nco + gx] = r;
            } else {
                s_o[e] = r;
            }
        }
        tail_barrier();
        if (!last) {
            for (int e = threadIdx.x; e < no * no; e += 256) s_a[e] = s_o[e];
            // (the barrier after the detail staging of the next step publishes s_a)
        }
    }
}

// =================================================================================================
// host side
// =================================================================================================
#define PDWT_CHECK_LAUNCH() PDWT_HIP_TRY(hipGetLastError())

static int g_tail_enable = -1;
static bool tail_enabled()
{
    if (g_tail_enable < 0) {
        const char* e = getenv("PDWT_TAIL");
        g_tail_enable = (e && e[0] == '1') ? 1 : 0;  // opt-in: measured SLOWER than the per-level launches (see header)
    }
    return g_tail_enable == 1;
}
void tail_set_enabled(int on) { g_tail_enable = on ? 1 : 0; }

template <typename K>
static int set_lds(K kernel, size_t bytes)
{
    if (bytes > 64 * 1024) PDWT_HIP_TRY(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return PDWT_OK;
}

#define PDWT_TAIL_HLENS(X) X(4) X(6) X(8) X(10) X(12) X(14) X(16)

static bool tail_geom_ok(const int* nr, const int* nc, int nlev, int unit)
{
    if (nlev < 2 || nlev > kTailMaxLev) return false;
    for (int k = 0; k <= nlev; k++) {
        if (k < nlev && ((nr[k] & 1) || (nc[k] & 1))) return false;  // even sizes at every tail input
    }
    // the deepest owned blocks must tile the deepest level exactly
    return nr[nlev] % unit == 0 && nc[nlev] % unit == 0;
}

int fwd2d_tail_f32(const float* in, float** c, int first_level, int nlevels, int nr, int nc, int hlen, const Taps2<float>& f)
{
    // tail levels: first_level .. nlevels-1 (0-based level index of the transform); input `in` = approximation of level first_level-1
    if (!tail_enabled()) return 1;
    const int L = nlevels - first_level;
    TailGeom g;
    TailBandsF b;
    if (L < 2 || L > kTailMaxLev) return 1;
    g.nlev = L;
    g.Nr[0] = nr;
    g.Nc[0] = nc;
    for (int k = 1; k <= L; k++) {
        g.Nr[k] = div2(g.Nr[k - 1]);
        g.Nc[k] = div2(g.Nc[k - 1]);
    }
    const int T = (g.Nr[L] % 16 == 0 && g.Nc[L] % 16 == 0) ? 16 : 8;
    if (!tail_geom_ok(g.Nr, g.Nc, L, T)) return 1;
    g.T = T;
    for (int k = 0; k < L; k++) {
        const int lev = first_level + k;
        b.H[k] = c[3 * lev + 1];
        b.V[k] = c[3 * lev + 2];
        b.D[k] = c[3 * lev + 3];
    }
    b.A = c[0];
    // LDS: n_in0^2 + 2*n_in0*n_out0 + n_out0^2
    int no = T;
    for (int k = L - 1; k > 0; k--) no = 2 * no + hlen - 2;
    const int n_in0 = 2 * no + hlen - 2;
    const size_t lds = ((size_t)n_in0 * n_in0 + 2 * (size_t)n_in0 * no + (size_t)no * no) * sizeof(float);
    if (lds > 100 * 1024) return 1;
    if (n_in0 > nr || n_in0 > nc || n_in0 * n_in0 > 256 * 48) return 1;  // single-wrap staging (wrap1) + staging register budget
    void (*k)(const float*, TailBandsF, TailGeom, Taps2<float>) = nullptr;
    switch (hlen) {
#define X(H) case H: k = k_fwd2d_tail<H>; break;
        PDWT_TAIL_HLENS(X)
#undef X
        default: return 1;
    }
    if (set_lds(k, lds) != PDWT_OK) return PDWT_EHIP;
    dim3 grid(g.Nc[L] / T, g.Nr[L] / T);
    KTimer kt(K_FWD2D_FUSED);
    hipLaunchKernelGGL(k, grid, dim3(256), lds, stream(), in, b, g, f);
    PDWT_CHECK_LAUNCH();
    return PDWT_OK;
}

int inv2d_head_f32(float* out, float** c, int first_level, int nlevels, int nr, int nc, int hlen, const Taps2<float>& f)
{
    // reconstructs the approximation of level first_level-1 (size nr x nc) from levels nlevels-1 .. first_level
    if (!tail_enabled()) return 1;
    const int L = nlevels - first_level;
    TailGeom g;
    TailBandsF b;
    if (L < 2 || L > kTailMaxLev) return 1;
    g.nlev = L;
    g.Nr[0] = nr;
    g.Nc[0] = nc;
    for (int k = 1; k <= L; k++) {
        g.Nr[k] = div2(g.Nr[k - 1]);
        g.Nc[k] = div2(g.Nc[k - 1]);
    }
    for (int k = 0; k < L; k++)
        if ((g.Nr[k] & 1) || (g.Nc[k] & 1)) return 1;
    const int T = 64;  // owned block of the OUTPUT
    if (nr % T != 0 || nc % T != 0) return 1;
    g.T = T;
    for (int k = 0; k < L; k++) {
        const int lev = first_level + k;
        b.H[k] = c[3 * lev + 1];
        b.V[k] = c[3 * lev + 2];
        b.D[k] = c[3 * lev + 3];
    }
    b.A = c[0];
    const int h2 = hlen / 2;
    const int n1 = T / 2 + h2 + 1;  // upper bound of the largest coefficient region edge
    const size_t lds = (5 * (size_t)n1 * n1 + 2 * (size_t)T * n1) * sizeof(float);
    if (lds > 100 * 1024) return 1;
    if (n1 > g.Nr[L] || n1 > g.Nc[L] || n1 * n1 > 256 * 8) return 1;  // single-wrap staging (wrap1) + staging register budget
    void (*k)(float*, TailBandsF, TailGeom, Taps2<float>) = nullptr;
    switch (hlen) {
#define X(H) case H: k = k_inv2d_head<H>; break;
        PDWT_TAIL_HLENS(X)
#undef X
        default: return 1;
    }
    if (set_lds(k, lds) != PDWT_OK) return PDWT_EHIP;
    dim3 grid(nc / T, nr / T);
    KTimer kt(K_INV2D_FUSED);
    hipLaunchKernelGGL(k, grid, dim3(256), lds, stream(), out, b, g, f);
    PDWT_CHECK_LAUNCH();
    return PDWT_OK;
}

}  // namespace pdwt
