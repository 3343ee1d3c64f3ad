/*
 * pdwt_oracle_impl.h -- body of the CPU oracle, included twice by pdwt_oracle.c with
 *   T  = float / double         (the reference's DTYPE, src/filters.h:16-30)
 *   SFX = f32 / f64             (function-name suffix)
 *   FMA(a,b,c)                  (fmaf / fma)
 *
 * TEST INFRASTRUCTURE ONLY.  See the header of pdwt_oracle.c.
 *
 * Everything is expressed through four strided 1-D line primitives (decimating analysis,
 * zero-stuffing synthesis, a-trous analysis, a-trous synthesis) applied along rows or columns;
 * the reference instead has one CUDA kernel per (pass, transform) pair.  The arithmetic per output
 * sample -- tap order, accumulation order, one fused multiply-add per tap (nvcc contracts
 * `acc += x*f` to FMA by default, SURVEY.md "Numerics notes") -- is the reference's.
 */

#define CAT_(a, b) a##_##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SFX)
#define FILTERS CAT(orc_filters, SFX)

typedef struct FILTERS {
    int hlen;
    T L[ORC_MAX_FILTER_WIDTH], H[ORC_MAX_FILTER_WIDTH], IL[ORC_MAX_FILTER_WIDTH], IH[ORC_MAX_FILTER_WIDTH];
} FILTERS;

/* ---- decimating analysis of `nlines` lines ------------------------------------------------
 * Follows w_kern_forward_pass1 / _pass2 (src/separable.cu:91-131, 135-176), SURVEY A-1:
 *   out[i] = sum_{j=0}^{hlen-1} xe[2i - c + j] * F[hlen-1-j],   c = hlen/2-1 (even) | hlen/2 (odd)
 * `xe` = periodic extension of x after virtually repeating the last sample when n is odd
 * (src/separable.cu:116-121 -> orc_wrap_ext).  Lines are `ls` apart, samples `es` apart; output
 * lines are `ols` apart, samples `oes` apart.  Two outputs (lo with FL, hi with FH) per input.
 * The loop nest is (line-block, output i, tap j, line) so that for column passes (es = pitch,
 * ls = 1) the innermost loop runs over contiguous memory; per sample the tap order is still
 * j = 0..hlen-1, exactly the reference's.                                                       */
static void FN(ana_lines)(const T* x, ptrdiff_t ls, ptrdiff_t es, int nlines, int n,
                          T* lo, T* hi, ptrdiff_t ols, ptrdiff_t oes,
                          int hlen, const T* FL, const T* FH)
{
    const int c = (hlen & 1) ? hlen / 2 : hlen / 2 - 1;
    const int no = orc_div2(n);
    if (ls == 1) { /* column pass: vectorise across lines */
#pragma omp parallel for schedule(static)
        for (int i = 0; i < no; i++) {
            T* plo = lo + (ptrdiff_t)i * oes;
            T* phi = hi + (ptrdiff_t)i * oes;
            for (int l = 0; l < nlines; l++) { plo[l * ols] = 0; phi[l * ols] = 0; }
            for (int j = 0; j < hlen; j++) {
                const T* px = x + (ptrdiff_t)orc_wrap_ext(2 * i - c + j, n) * es;
                const T fl = FL[hlen - 1 - j], fh = FH[hlen - 1 - j];
                for (int l = 0; l < nlines; l++) {
                    T v = px[l];
                    plo[l * ols] = FMA(v, fl, plo[l * ols]);
                    phi[l * ols] = FMA(v, fh, phi[l * ols]);
                }
            }
        }
    } else {
#pragma omp parallel for schedule(static)
        for (int l = 0; l < nlines; l++) {
            const T* px = x + (ptrdiff_t)l * ls;
            T* plo = lo + (ptrdiff_t)l * ols;
            T* phi = hi + (ptrdiff_t)l * ols;
            for (int i = 0; i < no; i++) {
                T sl = 0, sh = 0;
                const int s0 = 2 * i - c;
                if (s0 >= 0 && s0 + hlen - 1 <= n - 1) { /* interior: no wrap */
                    for (int j = 0; j < hlen; j++) {
                        T v = px[(ptrdiff_t)(s0 + j) * es];
                        sl = FMA(v, FL[hlen - 1 - j], sl);
                        sh = FMA(v, FH[hlen - 1 - j], sh);
                    }
                } else {
                    for (int j = 0; j < hlen; j++) {
                        T v = px[(ptrdiff_t)orc_wrap_ext(s0 + j, n) * es];
                        sl = FMA(v, FL[hlen - 1 - j], sl);
                        sh = FMA(v, FH[hlen - 1 - j], sh);
                    }
                }
                plo[(ptrdiff_t)i * oes] = sl;
                phi[(ptrdiff_t)i * oes] = sh;
            }
        }
    }
}

/* ---- synthesis (upsample by 2 + filter) of `nlines` lines ---------------------------------
 * Follows w_kern_inverse_pass1 / _pass2 (src/separable.cu:246-289, 293-328), SURVEY A-2.
 * Inputs a (low branch) and d (high branch) of length nin, output length nout (2*nin or 2*nin-1):
 *   h2 = hlen/2;  c = h2/2;  g' = g (h2 odd) | g+1 (h2 even);  p = g'/2;  off = 1-(g'&1)
 *   out[g] = sum_{j<h2} a[(p-c+j) mod nin]*IL[hlen-1-(2j+off)]  +  sum_{j<h2} d[..]*IH[..]
 * The two sums are accumulated separately and added once (src/separable.cu:284-287,322-326). */
static void FN(syn_lines)(const T* a, const T* d, ptrdiff_t ls, ptrdiff_t es, int nlines, int nin,
                          T* out, ptrdiff_t ols, ptrdiff_t oes, int nout,
                          int hlen, const T* FIL, const T* FIH)
{
    const int h2 = hlen / 2;
    const int c = h2 / 2;
    const int shift = (h2 & 1) ? 0 : 1;
    if (ls == 1) { /* column pass */
        T* acc = (T*)malloc(sizeof(T) * (size_t)nlines * 2 * (size_t)orc_max_threads());
#pragma omp parallel for schedule(static)
        for (int g = 0; g < nout; g++) {
            T* sa = acc + (size_t)orc_thread_num() * 2 * nlines;
            T* sd = sa + nlines;
            const int gp = g + shift, p = gp / 2, off = 1 - (gp & 1);
            for (int l = 0; l < nlines; l++) { sa[l] = 0; sd[l] = 0; }
            for (int j = 0; j < h2; j++) {
                const ptrdiff_t src = (ptrdiff_t)orc_wrap(p - c + j, nin) * es;
                const T fl = FIL[hlen - 1 - (2 * j + off)], fh = FIH[hlen - 1 - (2 * j + off)];
                const T* pa = a + src;
                const T* pd = d + src;
                for (int l = 0; l < nlines; l++) {
                    sa[l] = FMA(pa[l], fl, sa[l]);
                    sd[l] = FMA(pd[l], fh, sd[l]);
                }
            }
            T* po = out + (ptrdiff_t)g * oes;
            for (int l = 0; l < nlines; l++) po[l * ols] = sa[l] + sd[l];
        }
        free(acc);
    } else {
#pragma omp parallel for schedule(static)
        for (int l = 0; l < nlines; l++) {
            const T* pa = a + (ptrdiff_t)l * ls;
            const T* pd = d + (ptrdiff_t)l * ls;
            T* po = out + (ptrdiff_t)l * ols;
            for (int g = 0; g < nout; g++) {
                const int gp = g + shift, p = gp / 2, off = 1 - (gp & 1);
                T sa = 0, sd = 0;
                for (int j = 0; j < h2; j++) {
                    const ptrdiff_t src = (ptrdiff_t)orc_wrap(p - c + j, nin) * es;
                    sa = FMA(pa[src], FIL[hlen - 1 - (2 * j + off)], sa);
                    sd = FMA(pd[src], FIH[hlen - 1 - (2 * j + off)], sd);
                }
                po[(ptrdiff_t)g * oes] = sa + sd;
            }
        }
    }
}

/* ---- a-trous (undecimated) analysis, level `level` (1-based) -------------------------------
 * Follows w_kern_forward_swt_pass1/2 (src/separable.cu:409-448, 452-493), SURVEY A-3:
 *   f = 2^(level-1); c = (hlen/2-1)*f (even hlen) | (hlen/2)*f (odd)
 *   out[g] = sum_j x[(g - c + f*j) mod n] * F[hlen-1-j]                                        */
static void FN(swt_ana_lines)(const T* x, ptrdiff_t ls, ptrdiff_t es, int nlines, int n,
                              T* lo, T* hi, ptrdiff_t ols, ptrdiff_t oes,
                              int hlen, const T* FL, const T* FH, int level)
{
    const int f = 1 << (level - 1);
    const int c = ((hlen & 1) ? hlen / 2 : hlen / 2 - 1) * f;
    if (ls == 1) {
#pragma omp parallel for schedule(static)
        for (int g = 0; g < n; g++) {
            T* plo = lo + (ptrdiff_t)g * oes;
            T* phi = hi + (ptrdiff_t)g * oes;
            for (int l = 0; l < nlines; l++) { plo[l * ols] = 0; phi[l * ols] = 0; }
            for (int j = 0; j < hlen; j++) {
                const T* px = x + (ptrdiff_t)orc_wrap(g - c + f * j, n) * es;
                const T fl = FL[hlen - 1 - j], fh = FH[hlen - 1 - j];
                for (int l = 0; l < nlines; l++) {
                    plo[l * ols] = FMA(px[l], fl, plo[l * ols]);
                    phi[l * ols] = FMA(px[l], fh, phi[l * ols]);
                }
            }
        }
    } else {
#pragma omp parallel for schedule(static)
        for (int l = 0; l < nlines; l++) {
            const T* px = x + (ptrdiff_t)l * ls;
            for (int g = 0; g < n; g++) {
                T sl = 0, sh = 0;
                for (int j = 0; j < hlen; j++) {
                    T v = px[(ptrdiff_t)orc_wrap(g - c + f * j, n) * es];
                    sl = FMA(v, FL[hlen - 1 - j], sl);
                    sh = FMA(v, FH[hlen - 1 - j], sh);
                }
                lo[(ptrdiff_t)l * ols + (ptrdiff_t)g * oes] = sl;
                hi[(ptrdiff_t)l * ols + (ptrdiff_t)g * oes] = sh;
            }
        }
    }
}

/* ---- a-trous synthesis ---------------------------------------------------------------------
 * Follows w_kern_inverse_swt_pass1/2 (src/separable.cu:553-589, 593-626), SURVEY A-4:
 *   c = (hlen/2)*f;  out[g] = sum_j (a[(g-c+f*j) mod n]*IL[hlen-1-j])/2 + sum_j (d[..]*IH[..])/2
 * The reference writes `res += a * IL[..] / 2` (src/separable.cu:581-584, 621-622): the PRODUCT is rounded, halved (exact) and then
 * added -- the divide sits between the multiply and the add, so nvcc cannot contract them into one FMA.  Restated literally here
 * (round 5; until then the oracle pre-halved the tap and used one FMA, which differs from this by the rounding of the product: up
 * to one ulp per tap, inside the 1e-5 bar and pinned to pywt.iswt2 either way -- but not "bit-identical", as its comment claimed).
 * The HIP kernels keep pre-halved taps in one FMA per tap (deliberate deviation).                          */
static void FN(swt_syn_lines)(const T* a, const T* d, ptrdiff_t ls, ptrdiff_t es, int nlines, int n,
                              T* out, ptrdiff_t ols, ptrdiff_t oes,
                              int hlen, const T* FIL, const T* FIH, int level)
{
    const int f = 1 << (level - 1);
    const int c = (hlen / 2) * f;
    const int ntaps = hlen; /* hL+hR+1 = hlen for both parities (src/separable.cu:561-570) */
    if (ls == 1) {
        T* acc = (T*)malloc(sizeof(T) * (size_t)nlines * 2 * (size_t)orc_max_threads());
#pragma omp parallel for schedule(static)
        for (int g = 0; g < n; g++) {
            T* sa = acc + (size_t)orc_thread_num() * 2 * nlines;
            T* sd = sa + nlines;
            for (int l = 0; l < nlines; l++) { sa[l] = 0; sd[l] = 0; }
            for (int j = 0; j < ntaps; j++) {
                const ptrdiff_t src = (ptrdiff_t)orc_wrap(g - c + f * j, n) * es;
                const T fl = FIL[hlen - 1 - j], fh = FIH[hlen - 1 - j];
                for (int l = 0; l < nlines; l++) {
                    const T pa = a[src + l] * fl, pd = d[src + l] * fh; /* rounded products ... */
                    sa[l] += pa / 2;                                     /* ... halved exactly, then added */
                    sd[l] += pd / 2;
                }
            }
            T* po = out + (ptrdiff_t)g * oes;
            for (int l = 0; l < nlines; l++) po[l * ols] = sa[l] + sd[l];
        }
        free(acc);
    } else {
#pragma omp parallel for schedule(static)
        for (int l = 0; l < nlines; l++) {
            const T* pa = a + (ptrdiff_t)l * ls;
            const T* pd = d + (ptrdiff_t)l * ls;
            for (int g = 0; g < n; g++) {
                T sa = 0, sd = 0;
                for (int j = 0; j < ntaps; j++) {
                    const ptrdiff_t src = (ptrdiff_t)orc_wrap(g - c + f * j, n) * es;
                    const T qa = pa[src] * FIL[hlen - 1 - j], qd = pd[src] * FIH[hlen - 1 - j];
                    sa += qa / 2;
                    sd += qd / 2;
                }
                out[(ptrdiff_t)l * ols + (ptrdiff_t)g * oes] = sa + sd;
            }
        }
    }
}

/* =============================================================================================
 * Level drivers.  Same signature shape as the reference's L2 drivers
 * (src/separable.h:12-28, src/haar.h:9-16): (image, coeffs[], tmp, info) + the filter bank.
 * `coeffs` is a host array of band pointers laid out as src/common.cu:400-445; `tmp` has
 * 2*Nr*Nc elements (src/wt.cu:128-130).  Band 0 must be allocated at level-1 size.
 * ============================================================================================= */

/* w_forward_separable, src/separable.cu:179-209 */
int FN(orc_forward_separable)(const T* image, T** c, T* tmp, orc_info w, const FILTERS* f)
{
    int nr = w.Nr, nc = w.Nc;
    const T* in = image;
    T* t1 = tmp;
    T* t2 = tmp + (size_t)w.Nr * orc_div2(w.Nc);
    T* abuf = tmp + (size_t)2 * w.Nr * orc_div2(w.Nc); /* level input copy is not needed: see below */
    (void)abuf;
    for (int lev = 0; lev < w.nlevels; lev++) {
        const int nc2 = orc_div2(nc), nr2 = orc_div2(nr);
        /* rows: in (nr x nc) -> t1,t2 (nr x nc2) */
        FN(ana_lines)(in, nc, 1, nr, nc, t1, t2, nc2, 1, f->hlen, f->L, f->H);
        /* cols: t1 -> A,H ; t2 -> V,D (nr2 x nc2); pywt cH = H band, SURVEY 8(a) a6 */
        FN(ana_lines)(t1, 1, nc2, nc2, nr, c[0], c[3 * lev + 1], 1, nc2, f->hlen, f->L, f->H);
        FN(ana_lines)(t2, 1, nc2, nc2, nr, c[3 * lev + 2], c[3 * lev + 3], 1, nc2, f->hlen, f->L, f->H);
        in = c[0]; /* next level reads the approximation it will overwrite; rows pass reads all of it first */
        nr = nr2; nc = nc2;
    }
    return 0;
}

/* w_inverse_separable, src/separable.cu:332-364 */
int FN(orc_inverse_separable)(T* image, T** c, T* tmp, orc_info w, const FILTERS* f)
{
    int tNr[64] = {0}, tNc[64] = {0};
    tNr[0] = w.Nr; tNc[0] = w.Nc;
    for (int i = 1; i <= w.nlevels; i++) { tNr[i] = orc_div2(tNr[i - 1]); tNc[i] = orc_div2(tNc[i - 1]); }
    T* t1 = tmp;
    T* t2 = tmp + (size_t)w.Nr * tNc[1];
    for (int i = w.nlevels - 1; i >= 0; i--) {
        const int nri = tNr[i + 1], nci = tNc[i + 1]; /* coefficient size at this level */
        const int nro = tNr[i], nco = tNc[i];         /* output size */
        /* cols: (A,H)->t1, (V,D)->t2 : nri x nci -> nro x nci */
        FN(syn_lines)(c[0], c[3 * i + 1], 1, nci, nci, nri, t1, 1, nci, nro, f->hlen, f->IL, f->IH);
        FN(syn_lines)(c[3 * i + 2], c[3 * i + 3], 1, nci, nci, nri, t2, 1, nci, nro, f->hlen, f->IL, f->IH);
        /* rows: (t1,t2) nro x nci -> nro x nco */
        T* out = (i == 0) ? image : c[0];
        FN(syn_lines)(t1, t2, nci, 1, nro, nci, out, nco, 1, nco, f->hlen, f->IL, f->IH);
    }
    return 0;
}

/* w_forward_separable_1d, src/separable.cu:214-236 (batched 1D = row pass only) */
int FN(orc_forward_separable_1d)(const T* image, T** c, T* tmp, orc_info w, const FILTERS* f)
{
    int nc = w.Nc;
    const T* in = image;
    T* bufs[2] = { tmp, tmp + (size_t)w.Nr * orc_div2(w.Nc) };
    for (int lev = 0; lev < w.nlevels; lev++) {
        const int nc2 = orc_div2(nc);
        T* aout = (lev == w.nlevels - 1) ? c[0] : bufs[lev & 1];
        FN(ana_lines)(in, nc, 1, w.Nr, nc, aout, c[lev + 1], nc2, 1, f->hlen, f->L, f->H);
        in = aout; nc = nc2;
    }
    return 0;
}

/* w_inverse_separable_1d, src/separable.cu:368-395 */
int FN(orc_inverse_separable_1d)(T* image, T** c, T* tmp, orc_info w, const FILTERS* f)
{
    int tNc[64] = {0};
    tNc[0] = w.Nc;
    for (int i = 1; i <= w.nlevels; i++) tNc[i] = orc_div2(tNc[i - 1]);
    T* bufs[2] = { tmp, tmp + (size_t)w.Nr * tNc[1] };
    const T* a = c[0];
    for (int i = w.nlevels - 1; i >= 0; i--) {
        T* out = (i == 0) ? image : bufs[i & 1];
        FN(syn_lines)(a, c[i + 1], tNc[i + 1], 1, w.Nr, tNc[i + 1], out, tNc[i], 1, tNc[i], f->hlen, f->IL, f->IH);
        a = out;
    }
    return 0;
}

/* w_forward_swt_separable, src/separable.cu:496-516 */
int FN(orc_forward_swt_separable)(const T* image, T** c, T* tmp, orc_info w, const FILTERS* f)
{
    const int nr = w.Nr, nc = w.Nc;
    T* t1 = tmp;
    T* t2 = tmp + (size_t)nr * nc;
    const T* in = image;
    for (int lev = 0; lev < w.nlevels; lev++) {
        FN(swt_ana_lines)(in, nc, 1, nr, nc, t1, t2, nc, 1, f->hlen, f->L, f->H, lev + 1);
        FN(swt_ana_lines)(t1, 1, nc, nc, nr, c[0], c[3 * lev + 1], 1, nc, f->hlen, f->L, f->H, lev + 1);
        FN(swt_ana_lines)(t2, 1, nc, nc, nr, c[3 * lev + 2], c[3 * lev + 3], 1, nc, f->hlen, f->L, f->H, lev + 1);
        in = c[0];
    }
    return 0;
}

/* w_inverse_swt_separable, src/separable.cu:629-650 */
int FN(orc_inverse_swt_separable)(T* image, T** c, T* tmp, orc_info w, const FILTERS* f)
{
    const int nr = w.Nr, nc = w.Nc;
    T* t1 = tmp;
    T* t2 = tmp + (size_t)nr * nc;
    for (int i = w.nlevels - 1; i >= 0; i--) {
        FN(swt_syn_lines)(c[0], c[3 * i + 1], 1, nc, nc, nr, t1, 1, nc, f->hlen, f->IL, f->IH, i + 1);
        FN(swt_syn_lines)(c[3 * i + 2], c[3 * i + 3], 1, nc, nc, nr, t2, 1, nc, f->hlen, f->IL, f->IH, i + 1);
        T* out = (i == 0) ? image : c[0];
        FN(swt_syn_lines)(t1, t2, nc, 1, nr, nc, out, nc, 1, f->hlen, f->IL, f->IH, i + 1);
    }
    return 0;
}

/* w_forward_swt_separable_1d, src/separable.cu:520-537 */
int FN(orc_forward_swt_separable_1d)(const T* image, T** c, T* tmp, orc_info w, const FILTERS* f)
{
    const int nr = w.Nr, nc = w.Nc;
    T* bufs[2] = { tmp, tmp + (size_t)nr * nc };
    const T* in = image;
    for (int lev = 0; lev < w.nlevels; lev++) {
        T* aout = (lev == w.nlevels - 1) ? c[0] : bufs[lev & 1];
        FN(swt_ana_lines)(in, nc, 1, nr, nc, aout, c[lev + 1], nc, 1, f->hlen, f->L, f->H, lev + 1);
        in = aout;
    }
    return 0;
}

/* w_inverse_swt_separable_1d, src/separable.cu:654-672 */
int FN(orc_inverse_swt_separable_1d)(T* image, T** c, T* tmp, orc_info w, const FILTERS* f)
{
    const int nr = w.Nr, nc = w.Nc;
    T* bufs[2] = { tmp, tmp + (size_t)nr * nc };
    const T* a = c[0];
    for (int i = w.nlevels - 1; i >= 0; i--) {
        T* out = (i == 0) ? image : bufs[i & 1];
        FN(swt_syn_lines)(a, c[i + 1], nc, 1, nr, nc, out, nc, 1, f->hlen, f->IL, f->IH, i + 1);
        a = out;
    }
    return 0;
}

/* ---- Haar fast path -------------------------------------------------------------------------
 * kern_haar2d_fwd (src/haar.cu:10-37): 2x2 butterfly, indices clamped for odd sizes, NO wrap:
 *   a=x[2y,2x] b=x[2y,2x+1] c=x[2y+1,2x] d=x[2y+1,2x+1]
 *   A=.5((a+c)+(b+d))  V=.5((a+c)-(b+d))  H=.5((a-c)+(b-d))  D=.5((a-c)-(b-d))
 * The reference multiplies by the double literal 0.5 (exact in either precision).               */
static void FN(haar2d_fwd_level)(const T* x, T* cA, T* cH, T* cV, T* cD, int nr, int nc)
{
    const int nr2 = orc_div2(nr), nc2 = orc_div2(nc);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < nr2; y++) {
        const int y0 = 2 * y, y1 = (2 * y + 1 == nr) ? nr - 1 : 2 * y + 1;
        for (int xx = 0; xx < nc2; xx++) {
            const int x0 = 2 * xx, x1 = (2 * xx + 1 == nc) ? nc - 1 : 2 * xx + 1;
            const T a = x[(size_t)y0 * nc + x0], b = x[(size_t)y0 * nc + x1];
            const T cc = x[(size_t)y1 * nc + x0], d = x[(size_t)y1 * nc + x1];
            const size_t o = (size_t)y * nc2 + xx;
            cA[o] = (T)0.5 * ((a + cc) + (b + d));
            cV[o] = (T)0.5 * ((a + cc) - (b + d));
            cH[o] = (T)0.5 * ((a - cc) + (b - d));
            cD[o] = (T)0.5 * ((a - cc) - (b - d));
        }
    }
}

/* kern_haar2d_inv (src/haar.cu:41-58): out (nro x nco) from bands (nri x nci) */
static void FN(haar2d_inv_level)(T* out, const T* cA, const T* cH, const T* cV, const T* cD, int nri, int nci, int nro, int nco)
{
    (void)nri;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < nro; y++) {
        for (int xx = 0; xx < nco; xx++) {
            const size_t o = (size_t)(y / 2) * nci + (xx / 2);
            const T a = cA[o], b = cV[o], cc = cH[o], d = cD[o];
            T r;
            if (!(y & 1)) r = (xx & 1) ? (T)0.5 * ((a + cc) - (b + d)) : (T)0.5 * ((a + cc) + (b + d));
            else          r = (xx & 1) ? (T)0.5 * ((a - cc) - (b - d)) : (T)0.5 * ((a - cc) + (b - d));
            out[(size_t)y * nco + xx] = r;
        }
    }
}

/* haar_forward2d, src/haar.cu:61-86 */
int FN(orc_haar_forward2d)(const T* image, T** c, T* tmp, orc_info w)
{
    int nr = w.Nr, nc = w.Nc;
    const T* in = image;
    T* bufs[2] = { tmp, tmp + (size_t)orc_div2(w.Nr) * orc_div2(w.Nc) };
    for (int lev = 0; lev < w.nlevels; lev++) {
        T* aout = (lev == w.nlevels - 1) ? c[0] : bufs[lev & 1];
        FN(haar2d_fwd_level)(in, aout, c[3 * lev + 1], c[3 * lev + 2], c[3 * lev + 3], nr, nc);
        in = aout; nr = orc_div2(nr); nc = orc_div2(nc);
    }
    return 0;
}

/* haar_inverse2d, src/haar.cu:88-119 */
int FN(orc_haar_inverse2d)(T* image, T** c, T* tmp, orc_info w)
{
    int tNr[64] = {0}, tNc[64] = {0};
    tNr[0] = w.Nr; tNc[0] = w.Nc;
    for (int i = 1; i <= w.nlevels; i++) { tNr[i] = orc_div2(tNr[i - 1]); tNc[i] = orc_div2(tNc[i - 1]); }
    T* bufs[2] = { tmp, tmp + (size_t)tNr[1] * tNc[1] };
    const T* a = c[0];
    for (int i = w.nlevels - 1; i >= 0; i--) {
        T* out = (i == 0) ? image : bufs[i & 1];
        FN(haar2d_inv_level)(out, a, c[3 * i + 1], c[3 * i + 2], c[3 * i + 3], tNr[i + 1], tNc[i + 1], tNr[i], tNc[i]);
        a = out;
    }
    return 0;
}

/* kern_haar1d_fwd / _inv (src/haar.cu:132-160): A = s*(x0+x1), D = s*(x0-x1) with s the DOUBLE
 * literal 0.70710678118654746 (src/haar.cu:128): in the f32 build the product is evaluated in
 * double and rounded once to float (SURVEY 8(a) a16 / numerics note iii).                        */
#define ORC_ONE_SQRT2 0.70710678118654746
static void FN(haar1d_fwd_level)(const T* x, T* cA, T* cD, int nr, int nc)
{
    const int nc2 = orc_div2(nc);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < nr; y++)
        for (int i = 0; i < nc2; i++) {
            const int x1 = (2 * i + 1 == nc) ? nc - 1 : 2 * i + 1;
            const T a = x[(size_t)y * nc + 2 * i], b = x[(size_t)y * nc + x1];
            cA[(size_t)y * nc2 + i] = (T)(ORC_ONE_SQRT2 * (double)(a + b));
            cD[(size_t)y * nc2 + i] = (T)(ORC_ONE_SQRT2 * (double)(a - b));
        }
}
static void FN(haar1d_inv_level)(T* out, const T* cA, const T* cD, int nr, int nci, int nco)
{
#pragma omp parallel for schedule(static)
    for (int y = 0; y < nr; y++)
        for (int g = 0; g < nco; g++) {
            const T a = cA[(size_t)y * nci + g / 2], b = cD[(size_t)y * nci + g / 2];
            out[(size_t)y * nco + g] = (g & 1) ? (T)(ORC_ONE_SQRT2 * (double)(a - b)) : (T)(ORC_ONE_SQRT2 * (double)(a + b));
        }
}

/* haar_forward1d, src/haar.cu:163-186 */
int FN(orc_haar_forward1d)(const T* image, T** c, T* tmp, orc_info w)
{
    int nc = w.Nc;
    const T* in = image;
    T* bufs[2] = { tmp, tmp + (size_t)w.Nr * orc_div2(w.Nc) };
    for (int lev = 0; lev < w.nlevels; lev++) {
        T* aout = (lev == w.nlevels - 1) ? c[0] : bufs[lev & 1];
        FN(haar1d_fwd_level)(in, aout, c[lev + 1], w.Nr, nc);
        in = aout; nc = orc_div2(nc);
    }
    return 0;
}

/* haar_inverse1d, src/haar.cu:193-221 */
int FN(orc_haar_inverse1d)(T* image, T** c, T* tmp, orc_info w)
{
    int tNc[64] = {0};
    tNc[0] = w.Nc;
    for (int i = 1; i <= w.nlevels; i++) tNc[i] = orc_div2(tNc[i - 1]);
    T* bufs[2] = { tmp, tmp + (size_t)w.Nr * tNc[1] };
    const T* a = c[0];
    for (int i = w.nlevels - 1; i >= 0; i--) {
        T* out = (i == 0) ? image : bufs[i & 1];
        FN(haar1d_inv_level)(out, a, c[i + 1], w.Nr, tNc[i + 1], tNc[i]);
        a = out;
    }
    return 0;
}

/* ---- soft threshold: w_call_soft_thresh, src/common.cu:219-249 + kernels :13-52 ------------
 * Type-correct fabs/copysign (the reference calls the float versions even for double, quirk B-3). */
static void FN(soft_band)(T* v, size_t n, T beta)
{
#pragma omp parallel for schedule(static)
    for (ptrdiff_t i = 0; i < (ptrdiff_t)n; i++) {
        T x = v[i];
        T m = ORC_FABS(x) - beta;
        v[i] = ORC_COPYSIGN(m > 0 ? m : (T)0, x);
    }
}

int FN(orc_soft_thresh)(T** c, T beta, orc_info w, int do_thresh_appcoeffs, int normalize)
{
    int nr = w.Nr, nc = w.Nc;
    if (do_thresh_appcoeffs) {
        T beta2 = beta;
        if (normalize > 0) { /* beta / sqrt(2)^nlevels, src/common.cu:231-235 */
            int nl2 = w.nlevels / 2;
            beta2 /= (T)(1 << nl2);
            if (nl2 * 2 != w.nlevels) beta2 = (T)(beta2 / 1.4142135623730951);
        }
        int ar = w.Nr, ac = w.Nc; /* reference thresholds band 0 at LEVEL-1 size (src/common.cu:224-237):   */
        if (!w.do_swt) {          /* elements beyond A_L are scratch; only the first A_L block is meaningful, */
            for (int i = 0; i < w.nlevels; i++) { if (w.ndims > 1) ar = orc_div2(ar); ac = orc_div2(ac); }
        }                         /* so the oracle thresholds exactly the A_L elements.                       */
        FN(soft_band)(c[0], (size_t)ar * ac, beta2);
    }
    for (int i = 0; i < w.nlevels; i++) {
        if (!w.do_swt) { if (w.ndims > 1) nr = orc_div2(nr); nc = orc_div2(nc); }
        if (normalize > 0) beta = (T)(beta / 1.4142135623730951);
        const size_t n = (size_t)nr * nc;
        if (w.ndims > 1) { FN(soft_band)(c[3 * i + 1], n, beta); FN(soft_band)(c[3 * i + 2], n, beta); FN(soft_band)(c[3 * i + 3], n, beta); }
        else FN(soft_band)(c[i + 1], n, beta);
    }
    return 0;
}

/* =============================================================================================
 * Non-separable 2-D transform (reference src/nonseparable.cu), restated sample by sample.
 * Four hlen x hlen kernels K[0..3] = (LL, LH, HL, HH), row-major [y][x].  For the named wavelets they are the
 * outer products of the 1-D banks (w_compute_filters, src/nonseparable.cu:32-83):
 *   LL[y][x] = l[y] l[x],  LH[y][x] = l[y] h[x],  HL[y][x] = h[y] l[x],  HH[y][x] = h[y] h[x]
 * so the band the reference calls H (LH: low-pass along y, high-pass along x) is what its SEPARABLE path calls V
 * and vice versa (the "CHECKME" at :72,78): with do_separable = 0 the H and V bands come out swapped.
 * Accumulation order: jy outer, jx inner, one FMA per tap (nvcc contracts `res += v*k`).
 * ============================================================================================= */
typedef struct FN(orc_filters2d) {
    int hlen;
    T K[4][ORC_MAX_FILTER_WIDTH * ORC_MAX_FILTER_WIDTH];
} FN(orc_filters2d);

/* outer products of a 1-D pair (l, h): w_outer + w_compute_filters, src/nonseparable.cu:16-83 */
void FN(orc_outer_filters)(const T* l, const T* h, int hlen, FN(orc_filters2d)* out)
{
    out->hlen = hlen;
    for (int i = 0; i < hlen; i++)
        for (int j = 0; j < hlen; j++) {
            out->K[0][i * hlen + j] = l[i] * l[j];
            out->K[1][i * hlen + j] = l[i] * h[j];
            out->K[2][i * hlen + j] = h[i] * l[j];
            out->K[3][i * hlen + j] = h[i] * h[j];
        }
}

/* w_kern_forward, src/nonseparable.cu:114-170: one decimated level, in (Nr x Nc) -> 4 bands (Nr2 x Nc2) */
static void FN(ns_forward_level)(const T* img, T* cA, T* cH, T* cV, T* cD, int Nr, int Nc, const FN(orc_filters2d)* f)
{
    const int hlen = f->hlen;
    const int c = (hlen & 1) ? hlen / 2 : hlen / 2 - 1;
    const int Nr2 = orc_div2(Nr), Nc2 = orc_div2(Nc);
#pragma omp parallel for schedule(static)
    for (int gy = 0; gy < Nr2; gy++)
        for (int gx = 0; gx < Nc2; gx++) {
            T ra = 0, rh = 0, rv = 0, rd = 0;
            for (int jy = 0; jy < hlen; jy++) {
                const int iy = orc_wrap_ext(2 * gy - c + jy, Nr);
                for (int jx = 0; jx < hlen; jx++) {
                    const int ix = orc_wrap_ext(2 * gx - c + jx, Nc);
                    const T v = img[(size_t)iy * Nc + ix];
                    const int k = (hlen - 1 - jy) * hlen + (hlen - 1 - jx);
                    ra = FMA(v, f->K[0][k], ra);
                    rh = FMA(v, f->K[1][k], rh);
                    rv = FMA(v, f->K[2][k], rv);
                    rd = FMA(v, f->K[3][k], rd);
                }
            }
            const size_t o = (size_t)gy * Nc2 + gx;
            cA[o] = ra; cH[o] = rh; cV[o] = rv; cD[o] = rd;
        }
}

/* w_kern_inverse, src/nonseparable.cu:176-226: bands (Nr x Nc) -> img (Nr2 x Nc2 = output size) */
static void FN(ns_inverse_level)(T* img, const T* cA, const T* cH, const T* cV, const T* cD, int Nr, int Nc, int Nr2, int Nc2,
                                 const FN(orc_filters2d)* f)
{
    const int hlen = f->hlen, h2 = hlen / 2;
    const int c = h2 / 2;
    const int shift = (h2 & 1) ? 0 : 1;
#pragma omp parallel for schedule(static)
    for (int oy = 0; oy < Nr2; oy++)
        for (int ox = 0; ox < Nc2; ox++) {
            const int gy = oy + shift, gx = ox + shift;  /* "virtual id for shift" */
            const int offy = 1 - (gy & 1), offx = 1 - (gx & 1);
            T ra = 0, rh = 0, rv = 0, rd = 0;
            for (int jy = 0; jy < h2; jy++) {
                const int iy = orc_wrap(gy / 2 - c + jy, Nr);
                for (int jx = 0; jx < h2; jx++) {
                    const int ix = orc_wrap(gx / 2 - c + jx, Nc);
                    const int k = (hlen - 1 - (2 * jy + offy)) * hlen + (hlen - 1 - (2 * jx + offx));
                    const size_t o = (size_t)iy * Nc + ix;
                    ra = FMA(cA[o], f->K[0][k], ra);
                    rh = FMA(cH[o], f->K[1][k], rh);
                    rv = FMA(cV[o], f->K[2][k], rv);
                    rd = FMA(cD[o], f->K[3][k], rd);
                }
            }
            img[(size_t)oy * Nc2 + ox] = ra + rh + rv + rd;
        }
}

/* w_forward, src/nonseparable.cu:233-258 (ping-pong of the approximation; here through tmp, result in band 0) */
int FN(orc_forward_nonseparable)(const T* image, T** c, T* tmp, orc_info w, const FN(orc_filters2d)* f)
{
    int nr = w.Nr, nc = w.Nc;
    const T* in = image;
    T* bufs[2] = { tmp, tmp + (size_t)orc_div2(w.Nr) * orc_div2(w.Nc) };
    for (int i = 0; i < w.nlevels; i++) {
        T* aout = (i == w.nlevels - 1) ? c[0] : bufs[i & 1];
        FN(ns_forward_level)(in, aout, c[3 * i + 1], c[3 * i + 2], c[3 * i + 3], nr, nc, f);
        in = aout;
        nr = orc_div2(nr); nc = orc_div2(nc);
    }
    return 0;
}

/* w_inverse, src/nonseparable.cu:261-292; `f` holds the INVERSE kernels (outer products of IL, IH) */
int FN(orc_inverse_nonseparable)(T* image, T** c, T* tmp, orc_info w, const FN(orc_filters2d)* f)
{
    int tNr[64] = {0}, tNc[64] = {0};
    tNr[0] = w.Nr; tNc[0] = w.Nc;
    for (int i = 1; i <= w.nlevels; i++) { tNr[i] = orc_div2(tNr[i - 1]); tNc[i] = orc_div2(tNc[i - 1]); }
    T* bufs[2] = { tmp, tmp + (size_t)tNr[1] * tNc[1] };
    const T* a = c[0];
    for (int i = w.nlevels - 1; i >= 0; i--) {
        T* out = (i == 0) ? image : bufs[i & 1];
        FN(ns_inverse_level)(out, a, c[3 * i + 1], c[3 * i + 2], c[3 * i + 3], tNr[i + 1], tNc[i + 1], tNr[i], tNc[i], f);
        a = out;
    }
    return 0;
}

/* w_kern_forward_swt / w_kern_inverse_swt + drivers, src/nonseparable.cu:301-452 (tap spacing 2^(level-1)) */
static void FN(ns_swt_level)(const T* in, T* oA, T* oH, T* oV, T* oD, const T* iH, const T* iV, const T* iD, int Nr, int Nc, int level,
                             int inverse, const FN(orc_filters2d)* f)
{
    const int hlen = f->hlen, fac = 1 << (level - 1);
    const int c = (inverse ? ((hlen & 1) ? hlen / 2 : hlen / 2) : ((hlen & 1) ? hlen / 2 : hlen / 2 - 1)) * fac;
    const int ntap = inverse ? ((hlen & 1) ? hlen : hlen) : hlen;  /* hL + hR + 1 = hlen in both directions */
#pragma omp parallel for schedule(static)
    for (int gy = 0; gy < Nr; gy++)
        for (int gx = 0; gx < Nc; gx++) {
            T ra = 0, rh = 0, rv = 0, rd = 0;
            for (int jy = 0; jy < ntap; jy++) {
                const int iy = orc_wrap(gy - c + fac * jy, Nr);
                for (int jx = 0; jx < ntap; jx++) {
                    const int ix = orc_wrap(gx - c + fac * jx, Nc);
                    const int k = (hlen - 1 - jy) * hlen + (hlen - 1 - jx);
                    const size_t o = (size_t)iy * Nc + ix;
                    if (!inverse) {
                        const T v = in[o];
                        ra = FMA(v, f->K[0][k], ra);
                        rh = FMA(v, f->K[1][k], rh);
                        rv = FMA(v, f->K[2][k], rv);
                        rd = FMA(v, f->K[3][k], rd);
                    } else { /* res += c * k / 4  (src/nonseparable.cu:386-389: product rounded, then divided, then added) */
                        ra += in[o] * f->K[0][k] / 4;
                        rh += iH[o] * f->K[1][k] / 4;
                        rv += iV[o] * f->K[2][k] / 4;
                        rd += iD[o] * f->K[3][k] / 4;
                    }
                }
            }
            const size_t o = (size_t)gy * Nc + gx;
            if (!inverse) { oA[o] = ra; oH[o] = rh; oV[o] = rv; oD[o] = rd; }
            else oA[o] = ra + rh + rv + rd;
        }
}

int FN(orc_forward_swt_nonseparable)(const T* image, T** c, T* tmp, orc_info w, const FN(orc_filters2d)* f)
{
    const size_t n = (size_t)w.Nr * w.Nc;
    const T* in = image;
    T* bufs[2] = { tmp, tmp + n };
    for (int i = 0; i < w.nlevels; i++) {
        T* aout = (i == w.nlevels - 1) ? c[0] : bufs[i & 1];
        FN(ns_swt_level)(in, aout, c[3 * i + 1], c[3 * i + 2], c[3 * i + 3], 0, 0, 0, w.Nr, w.Nc, i + 1, 0, f);
        in = aout;
    }
    return 0;
}

int FN(orc_inverse_swt_nonseparable)(T* image, T** c, T* tmp, orc_info w, const FN(orc_filters2d)* f)
{
    const size_t n = (size_t)w.Nr * w.Nc;
    T* bufs[2] = { tmp, tmp + n };
    const T* a = c[0];
    for (int i = w.nlevels - 1; i >= 0; i--) {
        T* out = (i == 0) ? image : bufs[i & 1];
        FN(ns_swt_level)(a, out, 0, 0, 0, c[3 * i + 1], c[3 * i + 2], c[3 * i + 3], w.Nr, w.Nc, i + 1, 1, f);
        a = out;
    }
    return 0;
}



/* ---- norm1: Wavelets::norm1, src/wt.cu:398-418 (sum |c| over all bands incl. A) ------------
 * Accumulated in double, returned as double; callers round to T.                                */
double FN(orc_norm1)(T** c, orc_info w)
{
    int nr = w.Nr, nc = w.Nc;
    double res = 0;
    for (int i = 0; i < w.nlevels; i++) {
        if (!w.do_swt) { if (w.ndims > 1) nr = orc_div2(nr); nc = orc_div2(nc); }
        const size_t n = (size_t)nr * nc;
        const int nb = (w.ndims > 1) ? 3 : 1;
        for (int b = 0; b < nb; b++) {
            const T* v = (w.ndims > 1) ? c[3 * i + 1 + b] : c[i + 1];
            double s = 0;
#pragma omp parallel for reduction(+ : s) schedule(static)
            for (ptrdiff_t k = 0; k < (ptrdiff_t)n; k++) s += (double)ORC_FABS(v[k]);
            res += s;
        }
    }
    {
        const size_t n = (size_t)nr * nc;
        double s = 0;
#pragma omp parallel for reduction(+ : s) schedule(static)
        for (ptrdiff_t k = 0; k < (ptrdiff_t)n; k++) s += (double)ORC_FABS(c[0][k]);
        res += s;
    }
    return res;
}

#undef FN
#undef FILTERS
#undef CAT
#undef CAT_
