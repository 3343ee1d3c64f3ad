/*
 * pdwt_hip.h -- C-ABI of the MI355X-native PDWT hot path (libpdwt_hip.so).
 *
 * This is the drop-in boundary: every entry point below replaces one host-callable function of
 * the reference's level-driver layer (L2 in SURVEY.md section 1), which is the only thing the
 * reference's `Wavelets` class (src/wt.cu) calls to do device work.  Signatures are plain C:
 * device pointers, a host array of device pointers for the coefficient bands, a POD geometry
 * struct passed by value (== reference `w_info`, src/utils.h:9-19) and int error codes.  No HIP,
 * torch or C++ types appear, so the host side (include/wt.h + pdwt_amd/csrc/wt.cpp) builds with
 * a plain host compiler (g++) and any FFI (ctypes / Cython / cgo) can bind it.
 *
 * Differences from the reference seam, on purpose:
 *   - the filter bank is an ARGUMENT (per-instance state) instead of process-global
 *     __constant__ memory uploaded at construction (src/separable.cu:48-51, quirk B-1);
 *   - every function returns 0 on success or a negative PDWT_E* code (the reference's drivers
 *     always return 0 and never check a CUDA call, src/wt.cu:14-21 / B-10);
 *   - precision is a suffix (_f32/_f64) rather than a compile-time DTYPE macro, so one kernel
 *     library serves both libpdwt.so and libpdwtd.so (Makefile:29-39 of the reference).
 *
 * All device work is enqueued on ONE HIP stream per device (pdwt_get_stream(); owned by the library and
 * ordered against the NULL stream unless pdwt_set_stream / PDWT_STREAM_NONBLOCKING say otherwise); nothing
 * synchronises with the host except the functions documented to.
 */
#ifndef PDWT_HIP_H
#define PDWT_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PDWT_MAX_FILTER_WIDTH 40 /* reference: MAX_FILTER_WIDTH, src/common.h:15 */

/* error codes (negative), 0 = success */
#define PDWT_OK 0
#define PDWT_EINVAL (-1)   /* bad argument (NULL pointer, bad geometry, hlen out of range) */
#define PDWT_EUNKNOWN (-2) /* unknown wavelet name (same value as src/separable.cu:42-45) */
#define PDWT_EHIP (-3)     /* a HIP runtime call failed (see pdwt_last_error_string) */
#define PDWT_ENOMEM (-4)
#define PDWT_ENOTSUP (-5)  /* the optional facility is not available here (RCCL cannot be loaded, a device list it cannot take): use the fallback */

/* == reference `struct w_info`, src/utils.h:9-19 (same field order, 6 x int32) */
typedef struct pdwt_info {
    int ndims;   /* 1 = (batched) 1D along rows, 2 = 2D */
    int Nr;      /* rows (1D: number of signals in the batch) */
    int Nc;      /* columns (1D: samples per signal) */
    int nlevels; /* decomposition levels */
    int do_swt;  /* 1 = stationary (undecimated, a-trous) transform */
    int hlen;    /* filter length */
} pdwt_info;

/* The four 1-D filters of a bank, indexed exactly like pywt.Wavelet.filter_bank /
 * the reference's c_kern_L/H/IL/IH (src/common.h:28-31): L=dec_lo, H=dec_hi, IL=rec_lo, IH=rec_hi. */
typedef struct pdwt_filters_f32 {
    int hlen;
    float L[PDWT_MAX_FILTER_WIDTH], H[PDWT_MAX_FILTER_WIDTH], IL[PDWT_MAX_FILTER_WIDTH], IH[PDWT_MAX_FILTER_WIDTH];
} pdwt_filters_f32;
typedef struct pdwt_filters_f64 {
    int hlen;
    double L[PDWT_MAX_FILTER_WIDTH], H[PDWT_MAX_FILTER_WIDTH], IL[PDWT_MAX_FILTER_WIDTH], IH[PDWT_MAX_FILTER_WIDTH];
} pdwt_filters_f64;

/* ---------------------------------------------------------------------------------------------
 * Device / memory plumbing.  Replaces the bare cudaMalloc/cudaMemcpy/cudaMemset/cudaFree calls
 * the reference's class makes (src/wt.cu:117-130,421-468; src/common.cu:400-488) and the
 * cudaGetDeviceProperties call in print_informations (src/wt.cu:543-549).
 * ------------------------------------------------------------------------------------------- */
int pdwt_device_count(void);                 /* >=0, or PDWT_EHIP */
int pdwt_set_device(int dev);                /* reference has none (TODO.txt:15) */
int pdwt_get_device(void);
int pdwt_device_name(char* buf, int buflen); /* src/wt.cu:543-549 */
void* pdwt_malloc(size_t nbytes);            /* NULL on failure */
int pdwt_free(void* dptr);
int pdwt_memset(void* dptr, int byte, size_t nbytes);              /* stream-ordered */
int pdwt_memcpy_h2d(void* dst, const void* src, size_t nbytes);    /* blocking */
int pdwt_memcpy_d2h(void* dst, const void* src, size_t nbytes);    /* blocking (syncs the stream) */
int pdwt_memcpy_d2d(void* dst, const void* src, size_t nbytes);    /* stream-ordered (both buffers owned by the library) */
/* device-to-device copy FROM OR TO A BUFFER OF THE CALLER (Wavelets(d_ptr, memisonhost=0), set_image(d_ptr, 1),
 * set_coeff(d_ptr, num, 1)): waits for the NULL stream first, copies, waits for the copy -- the blocking semantics
 * of the reference's cudaMemcpy (src/wt.cu:121-124,433-436), whatever stream the caller's producer ran on. */
int pdwt_memcpy_d2d_foreign(void* dst, const void* src, size_t nbytes);
int pdwt_sync(void);                         /* wait for the library stream of the current device */
void* pdwt_get_stream(void);                 /* the hipStream_t all launches go to (opaque) */
/* Streams.  By default all work of a device goes to ONE stream the library creates WITHOUT hipStreamNonBlocking: it
 * is ordered against the NULL stream in both directions (legacy default-stream semantics), so a reference program,
 * whose own kernels run on the NULL stream, needs no extra synchronisation.  PDWT_STREAM_NONBLOCKING=1 (environment,
 * read once) creates it non-blocking instead.  pdwt_set_stream(s, 1) makes the library enqueue on the caller's
 * stream `s` (NULL = the NULL stream) for the current device; pdwt_set_stream(NULL, 0) returns to the library stream. */
int pdwt_set_stream(void* user_stream, int use_it);
const char* pdwt_last_error_string(void);    /* text of the last failing HIP call (thread-local) */

/* timing helpers for bench.py: HIP events recorded on the library stream */
void* pdwt_event_create(void);
int pdwt_event_record(void* ev);
int pdwt_event_sync(void* ev);
float pdwt_event_elapsed_ms(void* ev_start, void* ev_stop); /* <0 on error */
int pdwt_event_destroy(void* ev);
/* Per-kernel timing: when enabled, every kernel launch of the library on this thread is bracketed
 * by HIP events; pdwt_ktime_read() synchronises and returns {launch count, total ms} for the
 * kernel `kernel_id` (PDWT_K_* below) since the last reset.  Off by default (no events recorded). */
/* stream capture of the launches of one transform into a graph (launch-bound small transforms; no reference
 * counterpart).  begin -> enqueue drivers as usual (nothing executes) -> end returns an executable graph. */
int pdwt_graph_allowed(void);
int pdwt_graph_capture_begin(void);
int pdwt_graph_capture_end(void** exec_out);
int pdwt_graph_launch(void* exec);
int pdwt_graph_destroy(void* exec);
/* ---------------------------------------------------------------------------------------------
 * Batched 2-D DWT (no reference counterpart: the reference has no batching, TODO.txt:15; BASELINE.json's north star asks for
 * batched images).  nimg equally sized float32 images, each with its own band table and scratch exactly as for
 * pdwt_forward_separable_f32 (pdwt_create_coeffs_buffer_f32, pdwt_tmp_elems); every level of ALL images runs in ONE launch.
 * create returns NULL when the geometry is outside the streaming level kernels (then run the images one by one); the object
 * only keeps device-side pointer tables: images, bands and scratch stay the caller's and must outlive it.
 * Round 5: `info` may also describe a Haar transform (hlen 2: one launch per level of the Haar kernels, any size, either precision) or,
 * in float32, an undecimated one (do_swt = 1: one launch per level of the fused SWT level kernels, banks of up to 40 taps; bands and
 * scratch as for pdwt_forward_swt_separable_f32) -- the same create / forward / inverse / destroy entry points.
 * ------------------------------------------------------------------------------------------- */
void* pdwt_batch2d_create_f32(int nimg, float* const* d_images, float** const* d_coeffs, float* const* d_tmps, pdwt_info info);
int pdwt_batch2d_forward_f32(void* batch, const pdwt_filters_f32* f);
int pdwt_batch2d_inverse_f32(void* batch, const pdwt_filters_f32* f);
void pdwt_batch2d_destroy(void* batch);
/* the same in double precision (libpdwtd): every level of all images in one launch of the fused double-precision level kernels; any even
 * bank of up to 40 taps, odd sizes included, every level at least 16 rows and the (padded) bank length in either direction; NULL otherwise.
 * The object of the _f64 create goes to the _f64 forward / inverse; a handle of the other precision is refused with PDWT_EINVAL (every
 * handle carries its kind).  Either destroy releases any handle. */
void* pdwt_batch2d_create_f64(int nimg, double* const* d_images, double** const* d_coeffs, double* const* d_tmps, pdwt_info info);
int pdwt_batch2d_forward_f64(void* batch, const pdwt_filters_f64* f);
int pdwt_batch2d_inverse_f64(void* batch, const pdwt_filters_f64* f);
void pdwt_batch2d_destroy_f64(void* batch);

/* In-kernel clock probe of the fused level kernels of dwt_lds.hip (the C5 kernels): while enabled, workgroup 0 of every such
 * launch records the shader-clock counter and the 100 MHz real-time counter at its start and end.  slot = direction * 8 + size
 * class (forward 0, inverse 8; class 0 = 16384 rows, 1 = 8192, 2 = 4096, ...): the last launch of that kind.  shader_mhz = the
 * clock the workgroup actually ran at, span_us its lifetime; 0 when nothing was recorded.  Synchronises the stream. */
/* The one exchange step of the batch split driven from ONE host process (include/wt_batch.h): all-reduce(SUM) of one double per device
 * over RCCL (xGMI).  in[i] / out[i] are device pointers on devices[i] (distinct devices; may alias); the reduction is enqueued on every
 * device's library stream and the sum is returned in *result.  RCCL is loaded at run time: pdwt_rccl_available() says whether it could
 * be; PDWT_ENOTSUP = not here / device list not usable -> add the per-device doubles on the host.  (One process per GPU: pdwt_amd/batch.py
 * does the same all-reduce through torch.distributed, backend "nccl" = RCCL.)  Reference: none (single-GPU, TODO.txt:15). */
int pdwt_rccl_available(void);
int pdwt_rccl_allreduce_sum_f64(int n, const int* devices, const double* const* in, double* const* out, double* result);
/* the double the reductions above work on: element pdwt_sum_result_index() of a scratch buffer holds the result of
 * pdwt_norm1_enqueue_* / the one-pass threshold; pdwt_sum_spare_index() is a free double behind it (all-reduce output) */
size_t pdwt_sum_result_index(void);
size_t pdwt_sum_spare_index(void);
/* One-time hardware self-check for a new stepping / firmware: the hand-counted `s_waitcnt vmcnt(N)` pipelines of the streaming kernels
 * rely on a wave's loads and stores retiring in order, stores issued with EXEC = 0 included (undocumented; established on gfx950 with
 * tools/probes/vmcnt_order.hip).  Runs a compact form of that probe on the current device (~10 ms, 0.8 GB of scratch it frees again) and
 * returns the number of registers that were read before their load had landed: 0 = the assumption holds; > 0 = run with PDWT_CASC=0
 * PDWT_STREAM=0 (compiler-counted kernels); < 0 = PDWT_E*. */
long long pdwt_selfcheck_vmcnt_order(void);
/* Bandwidth probe (measurement only, bench.py roofline.copy_ceiling): one launch on the library stream that copies `bytes` from src to dst
 * (mode 0), only reads src (1; dst needs 16 valid bytes) or only writes dst (2) with 16-byte accesses, eight in flight per lane, one
 * contiguous chunk per workgroup.  Time it with pdwt_event_*. */
int pdwt_probe_bandwidth(const void* src, void* dst, size_t bytes, int mode);
int pdwt_clock_probe_enable(int on);
int pdwt_clock_probe_read(int slot, double* shader_mhz, double* span_us);
/* diagnostic: enable(2) / enable(3) make EVERY workgroup of the forward / inverse launches record (the last launch wins);
 * dump copies nblocks x {clk0, t0, clk1, t1} (t in 100 MHz ticks) out.  tools/lds_trace.py */
int pdwt_clock_probe_dump(unsigned long long* out, int nblocks);
int pdwt_ktime_enable(int on);
int pdwt_ktime_reset(void);
int pdwt_ktime_read(int kernel_id, int* n_launches, double* total_ms);
const char* pdwt_kernel_name(int kernel_id); /* NULL if out of range */
int pdwt_kernel_count(void);

/* ---------------------------------------------------------------------------------------------
 * Filters.  Replaces w_compute_filters_separable (src/separable.cu:19-54, src/separable.h:5):
 * same name lookup (case-insensitive, 72 names of src/filters.cpp:5919-6002, haar aliases
 * short-circuit when !do_swt) and same return value (hlen, or -2 when unknown), but the taps are
 * written to *out (may be NULL to only query hlen) instead of device constant memory.
 * ------------------------------------------------------------------------------------------- */
int pdwt_compute_filters_separable_f32(const char* wname, int do_swt, pdwt_filters_f32* out);
int pdwt_compute_filters_separable_f64(const char* wname, int do_swt, pdwt_filters_f64* out);
int pdwt_num_wavelets(void);                  /* 72 */
const char* pdwt_wavelet_name(int idx);       /* table order of src/filters.cpp:5919-6002 */

/* ---------------------------------------------------------------------------------------------
 * Coefficient buffers.  Replace w_create/free/copy_coeffs_buffer[_1d]
 * (src/common.h:62-68, src/common.cu:400-488).  Layout (host array of device pointers):
 *   2D: [A_L, H1,V1,D1, ..., H_L,V_L,D_L] (3L+1 bands), 1D: [A_L, D1..D_L] (L+1 bands);
 *   level i band size = ceil-halved i times (src/utils.cu:24-27), SWT: all bands Nr x Nc;
 *   band 0 is allocated at level-1 size (it doubles as scratch, src/common.cu:421-423).
 * All bands live in ONE device allocation (band pointers are 256-byte aligned offsets into it),
 * zero-filled; free with pdwt_free_coeffs_buffer_*.
 * ------------------------------------------------------------------------------------------- */
float** pdwt_create_coeffs_buffer_f32(pdwt_info info);   /* dispatches on info.ndims */
double** pdwt_create_coeffs_buffer_f64(pdwt_info info);
int pdwt_free_coeffs_buffer_f32(float** coeffs, pdwt_info info);
int pdwt_free_coeffs_buffer_f64(double** coeffs, pdwt_info info);
int pdwt_copy_coeffs_buffer_f32(float** dst, float** src, pdwt_info info);
int pdwt_copy_coeffs_buffer_f64(double** dst, double** src, pdwt_info info);
/* number of bands and element count of band `num` (the arithmetic of src/wt.cu:441-465,480-504) */
int pdwt_num_bands(pdwt_info info);
long long pdwt_band_size(pdwt_info info, int num, int* band_Nr, int* band_Nc);

/* ---------------------------------------------------------------------------------------------
 * Transform drivers.  One per reference driver, same argument meaning:
 *   (d_image, d_coeffs /+host array of device ptrs+/, d_tmp /+2*Nr*Nc elements+/, info by value)
 * + the filter bank.  Observable effects are the reference's: forward fills every band and
 * leaves d_image intact; inverse overwrites d_image and clobbers band 0 (src/wt.cu:273-307,
 * SURVEY B-5/B-6).  Scratch usage inside d_tmp is an implementation detail.
 *   forward_separable      <- w_forward_separable        src/separable.cu:179-209
 *   forward_separable_1d   <- w_forward_separable_1d     src/separable.cu:214-236
 *   inverse_separable      <- w_inverse_separable        src/separable.cu:332-364
 *   inverse_separable_1d   <- w_inverse_separable_1d     src/separable.cu:368-395
 *   forward_swt_separable[_1d] <- src/separable.cu:496-537
 *   inverse_swt_separable[_1d] <- src/separable.cu:629-672
 *   haar_forward2d/inverse2d/forward1d/inverse1d <- src/haar.cu:61-119,163-221
 * ------------------------------------------------------------------------------------------- */
/* test / tuning knobs (not part of the reference seam; names and meaning in INTEGRATION.md).  Each knob is
 * initialised ONCE from its PDWT_<NAME> environment variable and changed at run time only through
 * pdwt_debug_set; e.g. "force_twopass" = 1 makes the 2D DWT drivers use the two-pass (row kernel + column
 * kernel) form instead of the fused level kernels.  Unknown key: PDWT_EINVAL. */
int pdwt_debug_set(const char* key, int value);
int pdwt_debug_get(const char* key, int* value);

/* minimum element count of the d_tmp scratch the drivers need (2*Nr*Nc as in src/wt.cu:128-130,
 * plus alignment slack for the sub-buffers carved out of it) */
size_t pdwt_tmp_elems(pdwt_info info);

#define PDWT_DECL_DRIVERS(T, S)                                                                               \
    int pdwt_forward_separable_##S(T* d_image, T** d_coeffs, T* d_tmp, pdwt_info info, const pdwt_filters_##S* f);       \
    int pdwt_forward_separable_1d_##S(T* d_image, T** d_coeffs, T* d_tmp, pdwt_info info, const pdwt_filters_##S* f);    \
    int pdwt_inverse_separable_##S(T* d_image, T** d_coeffs, T* d_tmp, pdwt_info info, const pdwt_filters_##S* f);       \
    int pdwt_inverse_separable_1d_##S(T* d_image, T** d_coeffs, T* d_tmp, pdwt_info info, const pdwt_filters_##S* f);    \
    int pdwt_forward_swt_separable_##S(T* d_image, T** d_coeffs, T* d_tmp, pdwt_info info, const pdwt_filters_##S* f);   \
    int pdwt_forward_swt_separable_1d_##S(T* d_image, T** d_coeffs, T* d_tmp, pdwt_info info, const pdwt_filters_##S* f);\
    int pdwt_inverse_swt_separable_##S(T* d_image, T** d_coeffs, T* d_tmp, pdwt_info info, const pdwt_filters_##S* f);   \
    int pdwt_inverse_swt_separable_1d_##S(T* d_image, T** d_coeffs, T* d_tmp, pdwt_info info, const pdwt_filters_##S* f);\
    int pdwt_haar_forward2d_##S(T* d_image, T** d_coeffs, T* d_tmp, pdwt_info info);                                     \
    int pdwt_haar_inverse2d_##S(T* d_image, T** d_coeffs, T* d_tmp, pdwt_info info);                                     \
    int pdwt_haar_forward1d_##S(T* d_image, T** d_coeffs, T* d_tmp, pdwt_info info);                                     \
    int pdwt_haar_inverse1d_##S(T* d_image, T** d_coeffs, T* d_tmp, pdwt_info info);
PDWT_DECL_DRIVERS(float, f32)
PDWT_DECL_DRIVERS(double, f64)

/* ---------------------------------------------------------------------------------------------
 * Coefficient utilities on the hot path (BASELINE.json north_star: soft_threshold, norm1).
 *   soft_thresh <- w_call_soft_thresh  src/common.cu:219-249 (+ kernels src/common.cu:13-52):
 *                  in-place copysign(max(|v|-beta,0),v) on every detail band, band 0 only when
 *                  do_thresh_appcoeffs; normalize>0 divides beta by sqrt(2) per level.
 *                  ONE launch for all bands (device-side band table) instead of L launches.
 *   norm1       <- Wavelets::norm1  src/wt.cu:398-418 (3L+1 cublas asum calls): sum of |c| over
 *                  ALL bands incl. band 0.  One reduction launch (wave64 shuffles -> LDS ->
 *                  per-block double partial) + one finalize launch; the result is accumulated in
 *                  double and rounded once; *out is written after a stream sync.
 * ------------------------------------------------------------------------------------------- */
int pdwt_soft_thresh_f32(float** d_coeffs, float beta, pdwt_info info, int do_thresh_appcoeffs, int normalize);
int pdwt_soft_thresh_f64(double** d_coeffs, double beta, pdwt_info info, int do_thresh_appcoeffs, int normalize);
int pdwt_norm1_f32(float** d_coeffs, pdwt_info info, float* out);
int pdwt_norm1_f64(double** d_coeffs, pdwt_info info, double* out);
/* Threshold and norm in ONE pass over the bands: soft_thresh as above and, as a by-product, sum |c| over ALL bands (the
 * approximation band included, thresholded or not) of the thresholded coefficients -- what pdwt_norm1 would return right
 * after.  `d_scratch`: device buffer of pdwt_sum_scratch_doubles() doubles, ANY contents (the entry point resets the
 * kernels' arrival counter on the library stream in front of every launch; no zero-fill is required of the caller),
 * reusable for any number of calls on the library stream -- one reduction at a time per buffer; no host synchronisation.
 * pdwt_sum_scratch_read copies the result out (synchronises the stream).  The block partials are added in a fixed
 * order by the last block to finish: run-to-run deterministic. */
size_t pdwt_sum_scratch_doubles(void);
int pdwt_soft_thresh_sum_f32(float** d_coeffs, float beta, pdwt_info info, int do_thresh_appcoeffs, int normalize, double* d_scratch);
int pdwt_soft_thresh_sum_f64(double** d_coeffs, double beta, pdwt_info info, int do_thresh_appcoeffs, int normalize, double* d_scratch);
int pdwt_sum_scratch_read(const double* d_scratch, double* out);
/* same reduction, result in double regardless of T (used to combine shards across GPUs) */
int pdwt_norm1_as_double_f32(float** d_coeffs, pdwt_info info, double* out);
int pdwt_norm1_as_double_f64(double** d_coeffs, pdwt_info info, double* out);
/* the same reduction, ENQUEUED only: partial sums and result go to `d_scratch` (pdwt_sum_scratch_doubles() doubles, owned by
 * the caller, any contents -- see above); pdwt_sum_scratch_read fetches the value later.  Lets a host thread start the reductions of several devices
 * before it waits for any of them (include/wt_batch.h). */
int pdwt_norm1_enqueue_f32(float** d_coeffs, pdwt_info info, double* d_scratch);
int pdwt_norm1_enqueue_f64(double** d_coeffs, pdwt_info info, double* d_scratch);

/* ---------------------------------------------------------------------------------------------
 * Remaining coefficient utilities of the class and the circular shift of
 * cycle spinning (row 2).  All in place on the band table, ONE launch each.
 *   hard_thresh       <- w_call_hard_thresh  src/common.cu:252-283 (+ kernels :57-94): v if |v| > beta else 0*v.
 *                        As in the reference the approximation band is thresholded with the UN-normalised beta
 *                        (it computes beta/sqrt(2)^L but passes beta, :262-270).
 *   proj_linf         <- w_call_proj_linf  src/common.cu:286-315 (+ :96-131): copysign(min(|v|,beta),v).
 *   shrink            <- w_shrink  src/common.cu:346-371 (3L+1 cublas scal): v / (1+beta).
 *   group_soft_thresh <- w_call_group_soft_thresh  src/common.cu:318-343 (+ :134-198): per position
 *                        r = max(1 - beta/||(h,v,d[,a])||_2, 0) (0 when the norm is 0), applied to h,v,d[,a];
 *                        the approximation joins the group at the last scale only (do_thresh_appcoeffs).
 *   norm2sq           <- Wavelets::norm2sq  src/wt.cu:370-395 (3L+1 cublas nrm2): sum of c^2 over all bands.
 *                        The reference's 1-D branch adds cublas_asum (sum |c|) of the detail bands (:389):
 *                        FIXED, the squared l2 norm is returned (knob "norm2sq_ref1d" = 1 reproduces the reference
 *                        value).  Accumulated in double, rounded once.
 *   add_coeffs        <- w_add_coeffs / w_add_coeffs_1d  src/common.cu:499-526 (3L+1 cublas axpy):
 *                        dst[k] += alpha*src[k] for every band (whole bands, also for odd sizes in 1-D where
 *                        the reference's Nc/2 sizing leaves the last column of each band out).
 *   circshift         <- w_call_circshift + w_kern_circshift  src/common.cu:202-211,378-396:
 *                        out[y][x] = in[(y-sr) mod Nr][(x-sc) mod Nc] (sr forced to 0 for ndims 1); inplace != 0
 *                        leaves the result in d_image (d_image2 is the copy), else in d_image2.
 * ------------------------------------------------------------------------------------------- */
#define PDWT_DECL_UTILS(T, S)                                                                                   \
    int pdwt_hard_thresh_##S(T** d_coeffs, T beta, pdwt_info info, int do_thresh_appcoeffs, int normalize);     \
    int pdwt_proj_linf_##S(T** d_coeffs, T beta, pdwt_info info, int do_thresh_appcoeffs);                      \
    int pdwt_shrink_##S(T** d_coeffs, T beta, pdwt_info info, int do_thresh_appcoeffs);                         \
    int pdwt_group_soft_thresh_##S(T** d_coeffs, T beta, pdwt_info info, int do_thresh_appcoeffs, int normalize); \
    int pdwt_norm2sq_##S(T** d_coeffs, pdwt_info info, T* out);                                                 \
    int pdwt_norm2sq_as_double_##S(T** d_coeffs, pdwt_info info, double* out);                                  \
    int pdwt_add_coeffs_##S(T** d_dst, T** d_src, pdwt_info info, T alpha);                                     \
    int pdwt_circshift_##S(T* d_image, T* d_image2, pdwt_info info, int sr, int sc, int inplace);
PDWT_DECL_UTILS(float, f32)
PDWT_DECL_UTILS(double, f64)

/* ---------------------------------------------------------------------------------------------
 * Non-separable 2-D transform with four arbitrary hlen x hlen kernels (custom banks only).
 *   <- w_forward / w_inverse / w_forward_swt / w_inverse_swt  src/nonseparable.cu:233-292, 408-452
 *      (+ kernels :114-226, 301-400).
 * d_kernels: DEVICE pointer to 4*hlen*hlen taps, row-major [y][x], in the order LL, LH, HL, HH -- the
 * contents of the reference's c_kern_LL/LH/HL/HH constant arrays (src/nonseparable.cu:7-11) -- the forward
 * set for the forward calls, the inverse set for the inverse calls.  Same d_image / d_coeffs / d_tmp contract
 * as the separable drivers.  For the named wavelets (outer-product kernels) use the separable drivers on a
 * band table with H and V exchanged instead: same result, O(hlen) per sample.
 * ------------------------------------------------------------------------------------------- */
#define PDWT_DECL_NONSEP(T, S)                                                                                     \
    int pdwt_forward_nonseparable_##S(T* d_image, T** d_coeffs, T* d_tmp, pdwt_info info, const T* d_kernels);     \
    int pdwt_inverse_nonseparable_##S(T* d_image, T** d_coeffs, T* d_tmp, pdwt_info info, const T* d_kernels);     \
    int pdwt_forward_swt_nonseparable_##S(T* d_image, T** d_coeffs, T* d_tmp, pdwt_info info, const T* d_kernels); \
    int pdwt_inverse_swt_nonseparable_##S(T* d_image, T** d_coeffs, T* d_tmp, pdwt_info info, const T* d_kernels);
PDWT_DECL_NONSEP(float, f32)
PDWT_DECL_NONSEP(double, f64)

#ifdef __cplusplus
}
#endif
#endif /* PDWT_HIP_H */
