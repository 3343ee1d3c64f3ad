// wt.cpp -- host side of the `Wavelets` class (include/wt.h) above the C-ABI (include/pdwt_hip.h).
//
// Mirrors the reference's src/wt.cu: same constructor logic (level clamping src/wt.cu:111-114,
// 155-165; Nr==1 => 1D :133-136), same variant dispatch (:247-266, :283-301), same state machine
// (:237-240, :274-281, :311, :476), same band-size arithmetic for get/set_coeff (:441-465,480-504),
// same memory-footprint report (:527-541).  Plain host C++: compiled by g++, no HIP header, every
// device action is a C-ABI call into libpdwt_hip.so.  Built twice: libpdwt.so (float) and
// libpdwtd.so (-DDOUBLEPRECISION), like the reference Makefile:29-39.
//
// Deliberate fixes ("F" items): per-instance filters (B-1); unknown wavelet
// name or a clamp down to 0 levels is a creation error (B-2); return codes of the drivers are
// checked and mapped onto W_FORWARD_ERROR / W_INVERSE_ERROR / W_THRESHOLD_ERROR (B-10).
#include <limits.h>
#include <string.h>
#include <strings.h>

#include "../../include/pdwt_hip.h"
#include "../../include/wt.h"

static_assert(sizeof(w_info) == sizeof(pdwt_info), "w_info must mirror pdwt_info");

#ifndef DOUBLEPRECISION
#define SFX(name) name##_f32
typedef pdwt_filters_f32 filters_t;
#else
#define SFX(name) name##_f64
typedef pdwt_filters_f64 filters_t;
#endif

// ---- size helpers (reference src/utils.cu:4-34) ------------------------------------------------
int w_iDivUp(int a, int b) { return (a + b - 1) / b; }
int w_ipow2(int a) { return 1 << a; }
int w_ilog2(int i)
{
    int l = 0;
    while (i > 1) {
        i >>= 1;
        l++;
    }
    return l;
}
void w_div2(int* N) { *N = (*N + 1) / 2; }  // ceil-half: odd sizes get one extra element
void w_swap_ptr(DTYPE** a, DTYPE** b)
{
    DTYPE* t = *a;
    *a = *b;
    *b = t;
}

static inline pdwt_info to_pdwt(const w_info& w)
{
    pdwt_info p;
    memcpy(&p, &w, sizeof(p));
    return p;
}
// per-instance private state behind Wavelets::filters_: the 1-D bank first (so that F() is a plain cast), then the
// device copies of the custom non-separable kernels (4*hlen*hlen taps each; NULL unless set_filters_* gave four filters)
struct wstate_t {
    filters_t f;
    DTYPE* d_k2f;  // forward LL, LH, HL, HH
    DTYPE* d_k2i;  // inverse
    void* graph[2];  // recorded launches of forward() / inverse() (PDWT_GRAPH=1), NULL until first use
    int graph_off;   // capture failed once for this instance: plain launches from then on
    int dev;         // the device the instance lives on: the one current when it was constructed (w_set_device / pdwt_set_device)
    // norm1 bookkeeping: soft_threshold() leaves sum|c| of the thresholded bands in d_sum (one pass instead of two); norm1()
    // returns it as long as nothing has touched the coefficients since (sum_valid).  Handing out a raw band pointer
    // (coeff_int_ptr) switches the bookkeeping off for good: the caller may then write the bands behind the class's back.
    double* d_sum;
    int sum_valid;
    int raw_ptr_taken;
    // OPT-IN (set_norm_cache): off, soft_threshold() runs the plain kernel and norm1() always reduces the bands, like the
    // reference -- d_coeffs is a public member (src/wt.h:25), so the class cannot see a caller's kernel writing a band
    int norm_cache;
    int norm_enqueued;  // norm1_begin() has a reduction in flight into d_sum
    int custom;         // set_filters_forward / set_filters_inverse replaced the named bank
};
static inline void coeffs_changed(void* st)
{
    if (st) ((wstate_t*)st)->sum_valid = 0;
}
static inline filters_t* F(void* p) { return &((wstate_t*)p)->f; }
static inline wstate_t* WS(void* p) { return (wstate_t*)p; }

// Multi-device use from one host thread (the reference has none: TODO.txt:15): an instance belongs to the device that was
// current at its construction; every method that touches device memory switches to that device for its duration, so
// instances on different devices can be driven in turn (their work overlaps: launches are asynchronous).
struct DevScope {
    int prev, mine;
    explicit DevScope(const void* st) : prev(-1), mine(st ? ((const wstate_t*)st)->dev : -1)
    {
        if (mine < 0) return;
        prev = pdwt_get_device();
        if (prev != mine) pdwt_set_device(mine);
    }
    ~DevScope()
    {
        if (mine >= 0 && prev >= 0 && prev != mine) pdwt_set_device(prev);
    }
};
#define ON_MY_DEVICE() DevScope dev_scope_(filters_)

int w_set_device(int dev) { return pdwt_set_device(dev); }
int w_get_device(void) { return pdwt_get_device(); }
int w_device_count(void) { return pdwt_device_count(); }

static void report(const char* where, int rc)
{
    printf("ERROR: %s failed (code %d): %s\n", where, rc, pdwt_last_error_string());
}

// ---- constructors / destructor -------------------------------------------------------------------
Wavelets::Wavelets()
    : d_image(NULL), d_coeffs(NULL), d_tmp(NULL), current_shift_r(0), current_shift_c(0), do_separable(1), do_cycle_spinning(0),
      state(W_INIT), filters_(NULL)
{
    wname[0] = 0;
    memset(&winfos, 0, sizeof(winfos));
}

Wavelets::Wavelets(DTYPE* img, int Nr, int Nc, const char* wname_, int levels, int memisonhost, int do_separable_, int do_cycle_spinning_,
                   int do_swt, int ndim)
    : d_image(NULL), d_coeffs(NULL), d_tmp(NULL), current_shift_r(0), current_shift_c(0), do_separable(do_separable_),
      do_cycle_spinning(do_cycle_spinning_), state(W_INIT), filters_(NULL)
{
    winfos.Nr = Nr;
    winfos.Nc = Nc;
    winfos.nlevels = levels;
    winfos.do_swt = do_swt;
    winfos.ndims = ndim;
    winfos.hlen = 0;
    strncpy(this->wname, wname_ ? wname_ : "", 127);
    this->wname[127] = 0;

    if (Nr < 1 || Nc < 1 || !wname_) {
        puts("ERROR: Wavelets(): invalid image size or wavelet name");
        state = W_CREATION_ERROR;
        return;
    }
    if (levels < 1) {
        puts("Warning: cannot initialize wavelet coefficients with nlevels < 1. Forcing nlevels = 1");
        winfos.nlevels = 1;
    }
    if (Nr == 1) {  // a single row is a 1D signal
        ndim = 1;
        winfos.ndims = 1;
    }
    if (ndim == 1 && do_separable == 0) {
        puts("Warning: 1D DWT was requested, which is incompatible with non-separable transform.");
        puts("Ignoring the do_separable option.");
        do_separable = 1;
    }
    // do_separable == 0 (2-D): the reference convolves with the four hlen x hlen outer products of the bank
    // (src/nonseparable.cu:32-83).  For such tensor-product kernels that is the separable transform with the H and V
    // bands exchanged (LH = low-pass along y x high-pass along x is what the separable path calls V, :72-78), so this
    // build runs the separable kernels -- O(hlen) instead of O(hlen^2) per sample -- on a band table with H and V
    // swapped (nonsep_table below).  Genuinely non-separable custom kernels: set_filters_forward with four filters.
    if (ndim != 1 && ndim != 2) {
        printf("ERROR: ndim=%d is not implemented\n", ndim);
        state = W_CREATION_ERROR;
        return;
    }

    // filters: per-instance copy of the bank
    filters_ = calloc(1, sizeof(wstate_t));
    if (filters_) WS(filters_)->dev = pdwt_get_device();
    filters_t* fb = filters_ ? F(filters_) : NULL;
    int hlen = fb ? SFX(pdwt_compute_filters_separable)(this->wname, do_swt, fb) : 0;
    if (hlen <= 0) {
        printf("ERROR: unknown wavelet name %s\n", this->wname);
        state = W_CREATION_ERROR;
        return;
    }
    winfos.hlen = hlen;

    // maximum level the size allows (== pywt.dwt_max_level), reference src/wt.cu:155-165
    const int N = (ndim == 2) ? (Nr < Nc ? Nr : Nc) : Nc;
    const int wmaxlev = w_ilog2(N / (hlen - 1));
    if (winfos.nlevels > wmaxlev) {
        printf("Warning: required level (%d) is greater than the maximum possible level for %s (%d) on a %dx%d image.\n", winfos.nlevels,
               this->wname, wmaxlev, winfos.Nc, winfos.Nr);
        printf("Forcing nlevels = %d\n", wmaxlev);
        winfos.nlevels = wmaxlev;
    }
    if (winfos.nlevels < 1) {
        printf("ERROR: a %dx%d image is too small for one level of %s\n", Nc, Nr, this->wname);
        state = W_CREATION_ERROR;
        return;
    }
    if (do_cycle_spinning && do_swt) puts("Warning: makes little sense to use Cycle spinning with stationary Wavelet transform");
    if (do_cycle_spinning && ndim == 1) {
        puts("ERROR: cycle spinning is not implemented for 1D. Use SWT instead.");
        state = W_CREATION_ERROR;
        return;
    }

    // device buffers: image, scratch, bands
    const size_t nimg = (size_t)Nr * Nc;
    d_image = (DTYPE*)pdwt_malloc(nimg * sizeof(DTYPE));
    d_tmp = (DTYPE*)pdwt_malloc(pdwt_tmp_elems(to_pdwt(winfos)) * sizeof(DTYPE));
    d_coeffs = SFX(pdwt_create_coeffs_buffer)(to_pdwt(winfos));
    if (!d_image || !d_tmp || !d_coeffs) {
        printf("ERROR: Wavelets(): device allocation failed: %s\n", pdwt_last_error_string());
        state = W_CREATION_ERROR;
        return;
    }
    int rc;
    if (!img) rc = pdwt_memset(d_image, 0, nimg * sizeof(DTYPE));
    else if (memisonhost) rc = pdwt_memcpy_h2d(d_image, img, nimg * sizeof(DTYPE));
    else rc = pdwt_memcpy_d2d_foreign(d_image, img, nimg * sizeof(DTYPE));
    if (rc != PDWT_OK) {
        report("Wavelets(): image upload", rc);
        state = W_CREATION_ERROR;
    }
}

Wavelets::Wavelets(const Wavelets& W)
    : d_image(NULL), d_coeffs(NULL), d_tmp(NULL), current_shift_r(W.current_shift_r), current_shift_c(W.current_shift_c),
      do_separable(W.do_separable), do_cycle_spinning(W.do_cycle_spinning), winfos(W.winfos), state(W.state), filters_(NULL)
{
    memcpy(wname, W.wname, sizeof(wname));
    DevScope dev_scope_(W.filters_);  // the copy lives on the source's device
    if (W.filters_) {
        filters_ = calloc(1, sizeof(wstate_t));
        if (filters_) {
            *F(filters_) = *F(W.filters_);
            WS(filters_)->dev = WS(W.filters_)->dev;
            WS(filters_)->raw_ptr_taken = WS(W.filters_)->raw_ptr_taken;  // (d_sum / sum_valid stay 0: the copy starts without a cached norm)
            WS(filters_)->norm_cache = WS(W.filters_)->norm_cache;
            const size_t nb = 4 * (size_t)winfos.hlen * winfos.hlen * sizeof(DTYPE);
            for (int d = 0; d < 2; d++) {  // deep copy of the custom 2-D kernels
                DTYPE* src = d ? WS(W.filters_)->d_k2i : WS(W.filters_)->d_k2f;
                if (!src) continue;
                DTYPE* dst = (DTYPE*)pdwt_malloc(nb);
                if (dst && pdwt_memcpy_d2d(dst, src, nb) != PDWT_OK) {
                    pdwt_free(dst);
                    dst = NULL;
                }
                (d ? WS(filters_)->d_k2i : WS(filters_)->d_k2f) = dst;
            }
        }
    }
    if (!W.d_image || !W.d_coeffs || (winfos.ndims != 1 && winfos.ndims != 2)) {
        if (winfos.ndims != 1 && winfos.ndims != 2) puts("ERROR: 3D wavelets not implemented yet");
        state = W_CREATION_ERROR;
        return;
    }
    const size_t nimg = (size_t)winfos.Nr * winfos.Nc;
    d_image = (DTYPE*)pdwt_malloc(nimg * sizeof(DTYPE));
    d_tmp = (DTYPE*)pdwt_malloc(pdwt_tmp_elems(to_pdwt(winfos)) * sizeof(DTYPE));
    d_coeffs = SFX(pdwt_create_coeffs_buffer)(to_pdwt(winfos));
    if (!d_image || !d_tmp || !d_coeffs || pdwt_memcpy_d2d(d_image, W.d_image, nimg * sizeof(DTYPE)) != PDWT_OK ||
        SFX(pdwt_copy_coeffs_buffer)(d_coeffs, W.d_coeffs, to_pdwt(winfos)) != PDWT_OK) {
        printf("ERROR: Wavelets(copy): %s\n", pdwt_last_error_string());
        state = W_CREATION_ERROR;
    }
}

static void drop_graphs(void* st);
Wavelets::~Wavelets()
{
    ON_MY_DEVICE();
    if (d_image) pdwt_free(d_image);
    if (d_coeffs) SFX(pdwt_free_coeffs_buffer)(d_coeffs, to_pdwt(winfos));
    if (d_tmp) pdwt_free(d_tmp);
    drop_graphs(filters_);
    if (filters_) {
        if (WS(filters_)->d_k2f) pdwt_free(WS(filters_)->d_k2f);
        if (WS(filters_)->d_k2i) pdwt_free(WS(filters_)->d_k2i);
        if (WS(filters_)->d_sum) pdwt_free(WS(filters_)->d_sum);
    }
    free(filters_);
}

// ---- transforms ------------------------------------------------------------------------------------
// Band table handed to the level drivers.  Separable request: d_coeffs itself.  Non-separable request (2-D): the same
// bands with H and V exchanged at every level, see the constructor.
static DTYPE** nonsep_table(const Wavelets& W, DTYPE** scratch)
{
    if (W.do_separable || W.winfos.ndims != 2) return W.d_coeffs;
    scratch[0] = W.d_coeffs[0];
    for (int l = 0; l < W.winfos.nlevels; l++) {
        scratch[3 * l + 1] = W.d_coeffs[3 * l + 2];
        scratch[3 * l + 2] = W.d_coeffs[3 * l + 1];
        scratch[3 * l + 3] = W.d_coeffs[3 * l + 3];
    }
    return scratch;
}

// PDWT_GRAPH=1: record the launches of one forward()/inverse() of this instance once and replay them as ONE graph
// launch.  Worth it where the transform is launch-bound (small images: 6 launches of ~4 us of CPU enqueue each for
// 512^2 db4 L3); pointless for large ones.  Everything a recorded launch depends on is fixed for the instance's life
// (d_image, bands, d_tmp, sizes) except the filter taps: set_filters_* drops the graphs.
static bool graph_mode()
{
    static const int on = getenv("PDWT_GRAPH") ? atoi(getenv("PDWT_GRAPH")) : 0;
    return on == 1;
}
static void drop_graphs(void* st)
{
    if (!st) return;
    for (int d = 0; d < 2; d++)
        if (WS(st)->graph[d]) {
            pdwt_graph_destroy(WS(st)->graph[d]);
            WS(st)->graph[d] = NULL;
        }
}
// run `enqueue` (which only launches kernels on the library stream) through the instance's graph `dir` when possible
template <typename F>
static int run_graphed(void* st, int dir, bool eligible, F enqueue)
{
    if (!eligible || !graph_mode() || !st || WS(st)->graph_off || !pdwt_graph_allowed()) return enqueue();
    if (!WS(st)->graph[dir]) {
        if (pdwt_graph_capture_begin() != PDWT_OK) {
            WS(st)->graph_off = 1;
            return enqueue();
        }
        const int rc = enqueue();  // recorded, not executed
        void* exec = NULL;
        const int rc2 = pdwt_graph_capture_end(&exec);
        if (rc != PDWT_OK || rc2 != PDWT_OK) {
            // Nothing ran while recording.  A launcher's first use may do what a capture refuses (an LDS opt-in of a kernel, an
            // allocation of the diagnostic probes): the recording failed, not the transform -- plain launches from now on, and the
            // enqueue itself decides whether there is an error to report.
            if (exec) pdwt_graph_destroy(exec);
            WS(st)->graph_off = 1;
            return enqueue();
        }
        WS(st)->graph[dir] = exec;
    }
    return pdwt_graph_launch(WS(st)->graph[dir]);
}

void Wavelets::forward()
{
    ON_MY_DEVICE();
    if (state == W_CREATION_ERROR) {
        puts("Warning: forward transform not computed, as there was an error when creating the wavelets");
        return;
    }
    if (do_cycle_spinning) {
        current_shift_r = rand() % winfos.Nr;
        current_shift_c = rand() % winfos.Nc;
        circshift(current_shift_r, current_shift_c, 1);
    }
    coeffs_changed(filters_);
    const pdwt_info w = to_pdwt(winfos);
    const bool haar = (winfos.hlen == 2) && !winfos.do_swt;  // dedicated 2-tap kernels
    DTYPE* swapped[3 * 32 + 1];
    DTYPE** bands = nonsep_table(*this, swapped);
    const int rc = run_graphed(filters_, 0, true, [&]() -> int {
        int rc;
        if (winfos.ndims == 1) {
            if (haar) rc = SFX(pdwt_haar_forward1d)(d_image, d_coeffs, d_tmp, w);
            else if (!winfos.do_swt) rc = SFX(pdwt_forward_separable_1d)(d_image, d_coeffs, d_tmp, w, F(filters_));
            else rc = SFX(pdwt_forward_swt_separable_1d)(d_image, d_coeffs, d_tmp, w, F(filters_));
        } else {
            DTYPE* k2 = (!do_separable && filters_) ? WS(filters_)->d_k2f : NULL;  // custom non-separable kernels (nonsep.hip)
            if (k2) rc = winfos.do_swt ? SFX(pdwt_forward_swt_nonseparable)(d_image, d_coeffs, d_tmp, w, k2)
                                       : SFX(pdwt_forward_nonseparable)(d_image, d_coeffs, d_tmp, w, k2);
            else if (haar) rc = SFX(pdwt_haar_forward2d)(d_image, d_coeffs, d_tmp, w);
            else if (!winfos.do_swt) rc = SFX(pdwt_forward_separable)(d_image, bands, d_tmp, w, F(filters_));
            else rc = SFX(pdwt_forward_swt_separable)(d_image, bands, d_tmp, w, F(filters_));
        }
        return rc;
    });
    if (rc != PDWT_OK) {
        report("Wavelets::forward()", rc);
        state = W_FORWARD_ERROR;
        return;
    }
    state = W_FORWARD;
}

void Wavelets::inverse()
{
    ON_MY_DEVICE();
    if (state == W_INVERSE) {
        puts("Warning: W.inverse() has already been run. Inverse is available in W.get_image()");
        return;
    }
    if (state == W_CREATION_ERROR || state == W_FORWARD_ERROR || state == W_THRESHOLD_ERROR) {
        puts("Warning: inverse transform not computed, as there was an error in a previous stage");
        return;
    }
    coeffs_changed(filters_);  // inverse() consumes band 0
    const pdwt_info w = to_pdwt(winfos);
    const bool haar = (winfos.hlen == 2) && !winfos.do_swt;
    DTYPE* swapped[3 * 32 + 1];
    DTYPE** bands = nonsep_table(*this, swapped);
    if (winfos.ndims == 2 && !do_separable && filters_ && WS(filters_)->d_k2f && !WS(filters_)->d_k2i) {
        puts("ERROR: Wavelets::inverse(): custom non-separable forward filters were set without their inverse (set_filters_inverse)");
        state = W_INVERSE_ERROR;
        return;
    }
    const int rc = run_graphed(filters_, 1, true, [&]() -> int {
        int rc;
        if (winfos.ndims == 1) {
            if (haar) rc = SFX(pdwt_haar_inverse1d)(d_image, d_coeffs, d_tmp, w);
            else if (!winfos.do_swt) rc = SFX(pdwt_inverse_separable_1d)(d_image, d_coeffs, d_tmp, w, F(filters_));
            else rc = SFX(pdwt_inverse_swt_separable_1d)(d_image, d_coeffs, d_tmp, w, F(filters_));
        } else {
            DTYPE* k2 = (!do_separable && filters_) ? WS(filters_)->d_k2i : NULL;
            if (k2) rc = winfos.do_swt ? SFX(pdwt_inverse_swt_nonseparable)(d_image, d_coeffs, d_tmp, w, k2)
                                       : SFX(pdwt_inverse_nonseparable)(d_image, d_coeffs, d_tmp, w, k2);
            else if (haar) rc = SFX(pdwt_haar_inverse2d)(d_image, d_coeffs, d_tmp, w);
            else if (!winfos.do_swt) rc = SFX(pdwt_inverse_separable)(d_image, bands, d_tmp, w, F(filters_));
            else rc = SFX(pdwt_inverse_swt_separable)(d_image, bands, d_tmp, w, F(filters_));
        }
        return rc;
    });
    if (rc != PDWT_OK) {
        report("Wavelets::inverse()", rc);
        state = W_INVERSE_ERROR;
        return;
    }
    if (do_cycle_spinning) circshift(-current_shift_r, -current_shift_c, 1);
    state = W_INVERSE;
}

// ---- coefficient utilities -----------------------------------------------------------------------
static double* sum_scratch(wstate_t* st)
{
    if (!st->d_sum) {
        const size_t nb = pdwt_sum_scratch_doubles() * sizeof(double);
        st->d_sum = (double*)pdwt_malloc(nb);
        if (st->d_sum && pdwt_memset(st->d_sum, 0, nb) != PDWT_OK) {
            pdwt_free(st->d_sum);
            st->d_sum = NULL;
        }
    }
    return st->d_sum;
}

// Knob "norm_in_threshold" (PDWT_NORM_IN_THRESHOLD; pdwt_debug_set): -1 = per instance (Wavelets::set_norm_cache, default
// OFF: norm1() always reduces the bands), 0 = never, 1 = every instance (process-wide opt-in).  INTEGRATION.md section B.
static bool norm_in_threshold(const wstate_t* st)
{
    int v = -1;
    if (pdwt_debug_get("norm_in_threshold", &v) != PDWT_OK) v = -1;
    return v < 0 ? st->norm_cache == 1 : v == 1;
}

int Wavelets::custom_filters() const { return filters_ ? WS(filters_)->custom : 0; }

intptr_t Wavelets::norm1_scratch_int_ptr(void)
{
    ON_MY_DEVICE();
    if (state == W_CREATION_ERROR) return 0;
    wstate_t* st = WS(filters_);
    return (st && sum_scratch(st)) ? (intptr_t)st->d_sum : 0;
}

void Wavelets::set_norm_cache(int on)
{
    if (!filters_) return;
    WS(filters_)->norm_cache = on ? 1 : 0;
    if (!on) WS(filters_)->sum_valid = 0;
}

void Wavelets::soft_threshold(DTYPE beta, int do_thresh_appcoeffs, int normalize)
{
    ON_MY_DEVICE();
    if (state == W_INVERSE) {
        puts("Warning: Wavelets(): cannot threshold coefficients, as they were modified by W.inverse()");
        return;
    }
    if (state == W_CREATION_ERROR) return;
    coeffs_changed(filters_);
    wstate_t* st = WS(filters_);
    int rc;
    if (st && !st->raw_ptr_taken && norm_in_threshold(st) && !(beta < (DTYPE)0)) {
        (void)sum_scratch(st);
        if (st->d_sum) {
            rc = SFX(pdwt_soft_thresh_sum)(d_coeffs, beta, to_pdwt(winfos), do_thresh_appcoeffs, normalize, st->d_sum);
            if (rc == PDWT_OK) st->sum_valid = 1;
        } else {
            rc = SFX(pdwt_soft_thresh)(d_coeffs, beta, to_pdwt(winfos), do_thresh_appcoeffs, normalize);
        }
    } else {
        rc = SFX(pdwt_soft_thresh)(d_coeffs, beta, to_pdwt(winfos), do_thresh_appcoeffs, normalize);
    }
    if (rc != PDWT_OK) {
        report("Wavelets::soft_threshold()", rc);
        state = W_THRESHOLD_ERROR;
    }
}

// the double-precision value behind the last norm1() of this thread (shards of a batch are combined in double: wt_capi.cpp)
static thread_local double g_last_norm1 = 0.0;
double w_last_norm1_double(void) { return g_last_norm1; }

DTYPE Wavelets::norm1()
{
    ON_MY_DEVICE();
    if (state == W_CREATION_ERROR) return 0;
    wstate_t* st = WS(filters_);
    double d = 0;
    if (st && st->sum_valid && st->d_sum && !st->raw_ptr_taken && norm_in_threshold(st)) {
        // the last soft_threshold() left sum|c| behind and no method has touched the bands since
        const int rc = pdwt_sum_scratch_read(st->d_sum, &d);
        if (rc == PDWT_OK) {
            g_last_norm1 = d;
            return (DTYPE)d;
        }
        report("Wavelets::norm1()", rc);
    }
    int rc = SFX(pdwt_norm1_as_double)(d_coeffs, to_pdwt(winfos), &d);
    if (rc != PDWT_OK) report("Wavelets::norm1()", rc);
    g_last_norm1 = d;
    return (DTYPE)d;
}

// norm1() in two halves (additions, include/wt.h): begin enqueues, end reads.  The scratch of the one-pass threshold is
// reused: while its value is current (opted-in instances) nothing is launched at all.
void Wavelets::norm1_begin()
{
    ON_MY_DEVICE();
    if (state == W_CREATION_ERROR) return;
    wstate_t* st = WS(filters_);
    if (!st || !sum_scratch(st)) return;
    if (st->sum_valid && !st->raw_ptr_taken && norm_in_threshold(st)) return;
    const int rc = SFX(pdwt_norm1_enqueue)(d_coeffs, to_pdwt(winfos), st->d_sum);
    if (rc != PDWT_OK) report("Wavelets::norm1_begin()", rc);
    st->norm_enqueued = (rc == PDWT_OK);
}

int Wavelets::norm1_pending() const
{
    if (state == W_CREATION_ERROR) return 0;
    const wstate_t* st = WS(filters_);
    if (!st || !st->d_sum) return 0;
    const bool cached = st->sum_valid && !st->raw_ptr_taken && norm_in_threshold(st);
    return (cached || st->norm_enqueued) ? 1 : 0;
}

double Wavelets::norm1_end()
{
    ON_MY_DEVICE();
    if (state == W_CREATION_ERROR) return 0;
    wstate_t* st = WS(filters_);
    double d = 0;
    const bool cached = st && st->sum_valid && st->d_sum && !st->raw_ptr_taken && norm_in_threshold(st);
    if (st && st->d_sum && (cached || st->norm_enqueued)) {
        st->norm_enqueued = 0;
        const int rc = pdwt_sum_scratch_read(st->d_sum, &d);
        if (rc == PDWT_OK) {
            g_last_norm1 = d;
            return d;
        }
        report("Wavelets::norm1_end()", rc);
    }
    (void)norm1();  // no scratch / nothing enqueued: the one-call path
    return g_last_norm1;
}

// The remaining coefficient utilities (src/wt.cu:320-358): same state rule as soft_threshold.
#define PDWT_THRESH_METHOD(NAME, CALL)                                                                         \
    if (state == W_INVERSE) {                                                                                  \
        puts("Warning: Wavelets(): cannot threshold coefficients, as they were modified by W.inverse()");      \
        return;                                                                                                \
    }                                                                                                          \
    if (state == W_CREATION_ERROR) return;                                                                     \
    coeffs_changed(filters_);                                                                                  \
    {                                                                                                          \
        const int rc = CALL;                                                                                   \
        if (rc != PDWT_OK) {                                                                                   \
            report("Wavelets::" NAME "()", rc);                                                                \
            state = W_THRESHOLD_ERROR;                                                                         \
        }                                                                                                      \
    }
void Wavelets::hard_threshold(DTYPE beta, int do_thresh_appcoeffs, int normalize)
{
    ON_MY_DEVICE();
    PDWT_THRESH_METHOD("hard_threshold", SFX(pdwt_hard_thresh)(d_coeffs, beta, to_pdwt(winfos), do_thresh_appcoeffs, normalize))
}
void Wavelets::group_soft_threshold(DTYPE beta, int do_thresh_appcoeffs, int normalize)
{
    ON_MY_DEVICE();
    PDWT_THRESH_METHOD("group_soft_threshold", SFX(pdwt_group_soft_thresh)(d_coeffs, beta, to_pdwt(winfos), do_thresh_appcoeffs, normalize))
}
void Wavelets::shrink(DTYPE beta, int do_thresh_appcoeffs)
{
    ON_MY_DEVICE();
    PDWT_THRESH_METHOD("shrink", SFX(pdwt_shrink)(d_coeffs, beta, to_pdwt(winfos), do_thresh_appcoeffs))
}
void Wavelets::proj_linf(DTYPE beta, int do_thresh_appcoeffs)
{
    ON_MY_DEVICE();
    PDWT_THRESH_METHOD("proj_linf", SFX(pdwt_proj_linf)(d_coeffs, beta, to_pdwt(winfos), do_thresh_appcoeffs))
}

// src/wt.cu:364-366: if inplace = 1 the result is in d_image, otherwise in d_tmp
void Wavelets::circshift(int sr, int sc, int inplace)
{
    ON_MY_DEVICE();
    if (state == W_CREATION_ERROR || !d_image || !d_tmp) return;
    const int rc = SFX(pdwt_circshift)(d_image, d_tmp, to_pdwt(winfos), sr, sc, inplace);
    if (rc != PDWT_OK) report("Wavelets::circshift()", rc);
}

DTYPE Wavelets::norm2sq()
{
    ON_MY_DEVICE();
    if (state == W_CREATION_ERROR) return 0;
    DTYPE res = 0;
    const int rc = SFX(pdwt_norm2sq)(d_coeffs, to_pdwt(winfos), &res);
    if (rc != PDWT_OK) report("Wavelets::norm2sq()", rc);
    return res;
}

// Custom filter banks (src/wt.cu:560-602).  The taps become per-instance state (the reference uploads them to the
// process-global constant memory, SURVEY B-1).  Only the separable path exists in this build: filter3/filter4 of
// the non-separable form are rejected like the reference rejects their absence (-2).
// upload four len x len host kernels (LL, LH, HL, HH) into one device buffer; NULL on failure
static DTYPE* upload_k2(DTYPE* f1, DTYPE* f2, DTYPE* f3, DTYPE* f4, unsigned len)
{
    const size_t nb1 = (size_t)len * len * sizeof(DTYPE);
    DTYPE* d = (DTYPE*)pdwt_malloc(4 * nb1);
    if (!d) return NULL;
    DTYPE* src[4] = {f1, f2, f3, f4};
    for (int b = 0; b < 4; b++)
        if (pdwt_memcpy_h2d((char*)d + b * nb1, src[b], nb1) != PDWT_OK) {
            pdwt_free(d);
            return NULL;
        }
    return d;
}

int Wavelets::set_filters_forward(char* filtername, uint len, DTYPE* filter1, DTYPE* filter2, DTYPE* filter3, DTYPE* filter4)
{
    ON_MY_DEVICE();
    if (len > PDWT_MAX_FILTER_WIDTH) {
        printf("ERROR: Wavelets.set_filters_forward(): filter length (%d) exceeds the maximum size (%d)\n", (int)len, PDWT_MAX_FILTER_WIDTH);
        return -1;
    }
    if (!filters_) {
        filters_ = calloc(1, sizeof(wstate_t));
        if (!filters_) return -3;
    }
    if (!filter1 || !filter2 || len < 1) return -2;
    drop_graphs(filters_);  // recorded launches carry the old taps
    WS(filters_)->custom = 1;
    if (!do_separable) {  // four len x len kernels (w_set_filters_forward_nonseparable, src/nonseparable.cu:86-95)
        if (filter3 == NULL || filter4 == NULL) {
            puts("ERROR: Wavelets.set_filters_forward(): expected argument 4 and 5 for non-separable filtering");
            return -2;
        }
        DTYPE* d = upload_k2(filter1, filter2, filter3, filter4, len);
        if (!d) return -3;
        if (WS(filters_)->d_k2f) pdwt_free(WS(filters_)->d_k2f);
        WS(filters_)->d_k2f = d;
    } else {  // two 1-D filters (w_set_filters_forward, src/separable.cu:56-63)
        filters_t* f = F(filters_);
        for (unsigned i = 0; i < PDWT_MAX_FILTER_WIDTH; i++) {
            f->L[i] = (i < len) ? filter1[i] : (DTYPE)0;
            f->H[i] = (i < len) ? filter2[i] : (DTYPE)0;
        }
        f->hlen = (int)len;
    }
    winfos.hlen = (int)len;
    if (filtername) {
        strncpy(wname, filtername, sizeof(wname) - 1);
        wname[sizeof(wname) - 1] = 0;
    }
    return 0;
}

// the inverse filters are assumed to have the length given to set_filters_forward() (src/wt.cu:584-602)
int Wavelets::set_filters_inverse(DTYPE* filter1, DTYPE* filter2, DTYPE* filter3, DTYPE* filter4)
{
    ON_MY_DEVICE();
    if (!filter1 || !filter2 || !filters_) return -2;
    drop_graphs(filters_);
    WS(filters_)->custom = 1;
    const int len = winfos.hlen;
    if (!do_separable) {
        if (filter3 == NULL || filter4 == NULL) {
            puts("ERROR: Wavelets.set_filters_inverse(): expected argument 4 and 5 for non-separable filtering");
            return -2;
        }
        DTYPE* d = upload_k2(filter1, filter2, filter3, filter4, (unsigned)len);
        if (!d) return -3;
        if (WS(filters_)->d_k2i) pdwt_free(WS(filters_)->d_k2i);
        WS(filters_)->d_k2i = d;
        return 0;
    }
    filters_t* f = F(filters_);
    for (int i = 0; i < PDWT_MAX_FILTER_WIDTH; i++) {
        f->IL[i] = (i < len) ? filter1[i] : (DTYPE)0;
        f->IH[i] = (i < len) ? filter2[i] : (DTYPE)0;
    }
    return 0;
}

// In-place addition of wavelet coefficients: this += alpha * W (src/wt.cu:624-657); the operand comes by value
// (deep copy) as in the reference header.
int Wavelets::add_wavelet(Wavelets W, DTYPE alpha)
{
    ON_MY_DEVICE();
    if ((winfos.nlevels != W.winfos.nlevels) || (strcasecmp(wname, W.wname))) {
        puts("ERROR: add_wavelet(): right operand is not the same transform (wname, level)");
        return -1;
    }
    if (state == W_INVERSE || W.state == W_INVERSE) {
        puts("WARNING: add_wavelet(): this operation makes no sense when wavelet has just been inverted");
        return 1;
    }
    if (winfos.Nr != W.winfos.Nr || winfos.Nc != W.winfos.Nc || winfos.ndims != W.winfos.ndims) {
        puts("ERROR: add_wavelet(): operands do not have the same geometry");
        return -2;
    }
    if ((winfos.do_swt) ^ (W.winfos.do_swt)) {
        puts("ERROR: add_wavelet(): operands should both use SWT or DWT");
        return -3;
    }
    if ((do_cycle_spinning && W.do_cycle_spinning) && ((current_shift_r != W.current_shift_r) || (current_shift_c != W.current_shift_c))) {
        puts("ERROR: add_wavelet(): operands do not have the same current shift");
        return -4;
    }
    if (state == W_CREATION_ERROR || W.state == W_CREATION_ERROR || !d_coeffs || !W.d_coeffs) return -5;
    coeffs_changed(filters_);
    const int rc = SFX(pdwt_add_coeffs)(d_coeffs, W.d_coeffs, to_pdwt(winfos), alpha);
    if (rc != PDWT_OK) {
        report("Wavelets::add_wavelet()", rc);
        return -5;
    }
    pdwt_sync();  // W (a by-value copy) is destroyed on return: its bands must outlive the launch
    return 0;
}

// ---- data movement ---------------------------------------------------------------------------------
int Wavelets::get_image(DTYPE* res)
{
    ON_MY_DEVICE();
    if (!d_image || !res) return 0;
    const size_t n = (size_t)winfos.Nr * winfos.Nc;
    if (pdwt_memcpy_d2h(res, d_image, n * sizeof(DTYPE)) != PDWT_OK) return 0;
    return n > (size_t)INT_MAX ? INT_MAX : (int)n;  // element count as in the reference; saturates for >= 2^31 elements
}

void Wavelets::set_image(DTYPE* img, int mem_is_on_device)
{
    ON_MY_DEVICE();
    if (!d_image || !img) return;
    const size_t nb = (size_t)winfos.Nr * winfos.Nc * sizeof(DTYPE);
    int rc = mem_is_on_device ? pdwt_memcpy_d2d_foreign(d_image, img, nb) : pdwt_memcpy_h2d(d_image, img, nb);
    if (rc != PDWT_OK) report("Wavelets::set_image()", rc);
    if (state != W_CREATION_ERROR) state = W_INIT;
}

// band index -> element count.  2D: 0=A, then (H,V,D) per level; 1D: 0=A, then D per level.
static long long band_elems(const w_info& w, int num)
{
    return pdwt_band_size(to_pdwt(w), num, NULL, NULL);
}

void Wavelets::set_coeff(DTYPE* coeff, int num, int mem_is_on_device)
{
    ON_MY_DEVICE();
    if (!d_coeffs || !coeff) return;
    const long long n = band_elems(winfos, num);
    if (n <= 0) {
        printf("ERROR: set_coeff(): invalid coefficient index %d\n", num);
        return;
    }
    coeffs_changed(filters_);
    const size_t nb = (size_t)n * sizeof(DTYPE);
    int rc = mem_is_on_device ? pdwt_memcpy_d2d_foreign(d_coeffs[num], coeff, nb) : pdwt_memcpy_h2d(d_coeffs[num], coeff, nb);
    if (rc != PDWT_OK) report("Wavelets::set_coeff()", rc);
}

int Wavelets::get_coeff(DTYPE* coeff, int num)
{
    ON_MY_DEVICE();
    if (state == W_INVERSE) {
        puts("Warning: get_coeff(): inverse() has been performed, the coefficients has been modified and do not make sense anymore.");
        return 0;
    }
    if (!d_coeffs || !coeff) return 0;
    const long long n = band_elems(winfos, num);
    if (n <= 0) {
        printf("ERROR: get_coeff(): invalid coefficient index %d\n", num);
        return 0;
    }
    if (pdwt_memcpy_d2h(coeff, d_coeffs[num], (size_t)n * sizeof(DTYPE)) != PDWT_OK) return 0;
    return n > (long long)INT_MAX ? INT_MAX : (int)n;  // saturates for >= 2^31 elements (SWT bands of huge images)
}

void Wavelets::print_informations()
{
    ON_MY_DEVICE();
    const char* yn[2] = {"no", "yes"};
    puts("------------- Wavelet transform infos ------------");
    printf("Data dimensions : ");
    if (winfos.ndims == 2) printf("(%d, %d)\n", winfos.Nr, winfos.Nc);
    else if (winfos.Nr == 1) printf("%d\n", winfos.Nc);
    else printf("(%d, %d) [batched 1D transform]\n", winfos.Nr, winfos.Nc);
    printf("Wavelet name : %s\n", wname);
    printf("Number of levels : %d\n", winfos.nlevels);
    printf("Stationary WT : %s\n", yn[winfos.do_swt ? 1 : 0]);
    printf("Cycle spinning : %s\n", yn[do_cycle_spinning ? 1 : 0]);
    printf("Separable transform : %s\n", yn[do_separable ? 1 : 0]);
    // image (1) + bands + scratch (2); SWT keeps 3L+1 (2D) / L+1 (1D) full-size bands
    const double n = (double)winfos.Nr * winfos.Nc * sizeof(DTYPE);
    double mem;
    if (!winfos.do_swt) mem = 5 * n;
    else if (winfos.ndims == 2) mem = (3 * winfos.nlevels + 4) * n;
    else mem = (winfos.nlevels + 4) * n;
    printf("Estimated memory footprint : %.2f MB\n", mem / 1e6);
    char name[256] = "unknown";
    pdwt_device_name(name, (int)sizeof(name));
    printf("Running on device : %s\n", name);
    puts("--------------------------------------------------");
}

intptr_t Wavelets::image_int_ptr(void) { return (intptr_t)d_image; }
intptr_t Wavelets::coeff_int_ptr(int num)
{
    if (filters_) {  // the caller may write the band through this pointer: no more norm bookkeeping for this instance
        WS(filters_)->raw_ptr_taken = 1;
        WS(filters_)->sum_valid = 0;
    }
    return (intptr_t)d_coeffs[num];
}
