// common.hpp -- internals shared by the HIP translation units of libpdwt_hip.so (gfx950 only).
#pragma once
#include <atomic>
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/pdwt_hip.h"

namespace pdwt {

// ---- error plumbing ---------------------------------------------------------------------------
void set_last_error(hipError_t e, const char* what, const char* file, int line);

#define PDWT_HIP_TRY(expr)                                              \
    do {                                                                \
        hipError_t _e = (expr);                                         \
        if (_e != hipSuccess) {                                         \
            ::pdwt::set_last_error(_e, #expr, __FILE__, __LINE__);      \
            return PDWT_EHIP;                                           \
        }                                                               \
    } while (0)

// ---- stream + per-kernel timing ---------------------------------------------------------------
hipStream_t stream();  // the library stream of the current device (created lazily, non-blocking)

// kernel ids for pdwt_ktime_* (names in runtime.hip must stay in this order)
enum KernelId {
    K_FWD2D_FUSED = 0,
    K_INV2D_FUSED,
    K_ANA_ROWS,
    K_ANA_COLS,
    K_SYN_COLS,
    K_SYN_ROWS,
    K_SWT_ANA_ROWS,
    K_SWT_ANA_COLS,
    K_SWT_SYN_COLS,
    K_SWT_SYN_ROWS,
    K_HAAR2D_FWD,
    K_HAAR2D_INV,
    K_HAAR1D_FWD,
    K_HAAR1D_INV,
    K_SOFT_THRESH,
    K_ABS_SUM,
    K_ABS_SUM_FINAL,
    K_FWD2D_CASC,  // two levels per launch (dwt_casc.hip)
    K_INV2D_CASC,
    K_FWD2D_STREAM,  // one streaming launch per level (dwt_stream.hip)
    K_INV2D_STREAM,
    K_FWD2D_SMALL,   // register-tile kernels for the small, latency-bound levels (dwt_small.hip)
    K_INV2D_SMALL,
    K_FWD2D_F64,     // fused row+column level kernels for double-precision / long float32 banks (dwt_lds.hip)
    K_INV2D_F64,
    K_THRESH_SUM,    // soft threshold that also leaves sum|c| of the result behind (utils.hip)
    K_COUNT
};

// Bracket a launch with events when pdwt_ktime_enable(1) is active; otherwise free.
struct KTimer {
    int id;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    bool ext = false;  // events attached to the launch itself (PDWT_LAUNCH_KT) instead of recorded around it
    explicit KTimer(int kernel_id, bool attach_to_launch = false);
    ~KTimer();
};
// Launch through a KTimer constructed with attach_to_launch = true: when timing is on, the two events ride on the
// dispatch itself (hipExtLaunchKernelGGL) and read the kernel's own start / end timestamps -- what rocprofv3 reports --
// instead of the stream-ordered interval around it, which also contains the dispatch latency of ~2 us.
#define PDWT_LAUNCH_KT(kt, kernel, grid, block, lds, ...)                                                                  \
    do {                                                                                                                   \
        if ((kt).e0 && (kt).e1) hipExtLaunchKernelGGL(kernel, grid, block, lds, pdwt::stream(), (kt).e0, (kt).e1, 0, __VA_ARGS__); \
        else hipLaunchKernelGGL(kernel, grid, block, lds, pdwt::stream(), __VA_ARGS__);                                    \
    } while (0)

// More than 64 KB of dynamic LDS per workgroup is opt-in per (kernel, device).  The driver call is made ONCE for each pair
// (one function-local static per kernel: the kernel is the template argument) and asks for the most a kernel can ever request
// (160 KB), so afterwards nothing but the launch itself is on the enqueue path.
template <auto Kernel>
inline int lds_opt_in()
{
    static std::atomic<unsigned long long> done{0};
    int dev = 0;
    PDWT_HIP_TRY(hipGetDevice(&dev));
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(done.load(std::memory_order_relaxed) & bit)) {
        PDWT_HIP_TRY(hipFuncSetAttribute((const void*)Kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        done.fetch_or(bit, std::memory_order_relaxed);
    }
    return PDWT_OK;
}

// The same for kernels that are only known as a pointer at the call site (templates instantiated over many lengths): one driver
// call per (kernel, device), remembered in a small table (runtime.hip); every later launch is a lookup.
int lds_opt_in_ptr(const void* kernel);

// ---- in-kernel clock probe (pdwt_clock_probe_*, runtime.hip) ----------------------------------------------------
// Workgroup 0 of a probed launch stores (s_memtime, s_memrealtime) when it starts and when it ends: the shader-clock count over
// the constant 100 MHz count = the clock the kernel actually ran at, to be compared with what amdsmi reports for the same
// window (bench.py --config c5).  slot = 16 records of 4 x u64; nullptr while the probe is off (the default).
unsigned long long* clock_probe_slot(int slot);
inline int clock_probe_size_class(int rows) { int c = 0; while ((rows << c) < 16384 && c < 7) c++; return c; }  // 16384 rows -> 0, 8192 -> 1, ...
// all != 0 (pdwt_clock_probe_enable(2), diagnostic): EVERY workgroup records, 4 x u64 each, at probe + 4 * blockIdx.x
constexpr int kClockProbeAllBlocks = 16384;
unsigned long long* clock_probe_all(int* all);
__device__ __forceinline__ void clock_probe_stamp(unsigned long long* probe, int second, int all = 0)
{
    if (probe && (all || (blockIdx.x == 0 && blockIdx.y == 0))) {  // (uniform)
        unsigned long long* const p = probe + (all ? 4 * (size_t)blockIdx.x : (size_t)0);
        const unsigned long long c = clock64(), t = wall_clock64();
        if (threadIdx.x == 0) {
            p[2 * second] = c;
            p[2 * second + 1] = t;
        }
    }
}

// ---- tuning / test knobs ------------------------------------------------------------------------
// Every PDWT_* environment knob is read ONCE (first use of the table), never on the enqueue path; tests and tuning
// scripts change a value at run time through pdwt_debug_set("<name>", value).  Names and meaning: INTEGRATION.md.
enum KnobId {
    KN_FORCE_TWOPASS = 0,  // 1: 2D DWT levels as row kernel + column kernel (no fused / streaming / cascade kernels)
    KN_TILED_COLS,         // 1: LDS-tiled column kernels even where a register-ring instantiation exists
    KN_CASC,               // 0: one launch per level instead of the two-levels-per-launch kernels
    KN_CASC_WAVES,         // forward cascade: waves per launch (0 = auto)
    KN_CASC_NV,            // forward cascade: input rows in flight per wave (0 = auto)
    KN_CASC_MIN,           // smallest input (pixels) the cascade kernels take
    KN_CASC_IWAVES,        // inverse cascade: waves per launch (0 = auto)
    KN_CASC_IPFD,          // inverse cascade: prefetch distance in steps
    KN_CASC_WG,            // forward cascade: waves stacked per workgroup with LDS ring hand-off (0 = auto, 1 = independent waves)
    KN_CASC_IWG,           // inverse cascade: the same (0 = auto, 1 = independent waves)
    KN_CASC_L3,            // third level folded into the inverse cascade launch: 1 = streamed (dwt_casc_inv3.hip) where it applies, 2 = prologue form only, 0 = never
    KN_CASC_SPEC,          // cascade kernels: straight-line wave programs for the common per-wave row counts (bit 0 forward, bit 1 inverse)
    KN_CASC_XCDW,          // cascade wave programs: units a workgroup on an even XCD takes more (odd: fewer) than the even split of its strip (0 = even split)
    KN_DWT1D_LDS_KB,       // batched 1-D, all levels in one launch: largest LDS footprint (KB) of a row's buffers the fused kernels take
    KN_STREAM,             // 0: LDS-tiled fused level kernels instead of the streaming ones
    KN_STREAM_R,           // streaming level kernels: rows per wave (0 = auto)
    KN_STREAM_WAVES,       // streaming level kernels: target waves per launch
    KN_STREAM_NARROW,      // streaming forward: 8-byte lanes at or below this many pixels
    KN_SMALL,              // 0: no register-tile kernels for the small (latency-bound) levels
    KN_ROWS_TR,            // 0: generic row kernels for long double-precision banks
    KN_RING_R,             // ring column kernels: forced chunk height (0 = auto)
    KN_RING_WAVES,         // ring column kernels: waves per launch
    KN_SWTF,               // 0: two-pass SWT instead of the fused per-level kernels
    KN_SWTF_M,             // fused SWT forward: rows per chunk (0 = auto)
    KN_SWTF_MI,            // fused SWT inverse: rows per chunk (0 = auto)
    KN_SWTF_XCD,           // fused SWT levels: XCD-aware tile order
    KN_SWTF_ALT,           // fused SWT inverse: neighbouring chunks of a residue class walk in opposite directions (shared rows out of L2)
    KN_SWTF_PERM,          // fused SWT inverse, tap spacing 4/8/16: residue-major LDS rows
    KN_SWTF_F64,           // 0: two-pass SWT in double precision instead of the fused per-level kernels
    KN_F64_LDS,            // LDS-ring form of the fused long double-precision level kernels (dwt_lds.hip)
    KN_F64_LDS_MIN,        // ... smallest level side (pixels)
    KN_F64_LDS_WGS,        // ... workgroups to aim for
    KN_F64_LDS_MINGROUPS,  // ... shortest chunk, in groups of 4 output rows
    KN_F64_LDS_SKEW,       // dwt_lds.hip kernels, two workgroups per CU: % of a chunk pair's rows that go to the workgroup dispatched first (0 = even)
    KN_NORM2SQ_REF1D,      // 1: norm2sq of a 1-D transform adds sum|d| of the detail bands like the reference (src/wt.cu:389) instead of sum d^2
    KN_NORM_IN_THRESHOLD,  // sum|c| computed inside soft_threshold() and returned by the next norm1(): -1 = per instance (set_norm_cache), 0 = never, 1 = always
    KN_SELFCHECK,          // 1: the first use of a hand-counted-wait kernel on a device runs pdwt_selfcheck_vmcnt_order() (cached; failure -> compiler-counted kernels); 0: trust the build guard; 2: behave as if it had failed (test hook)
    KN_DWT1D_F64,          // 0: per-level row kernels for batched 1-D in double precision instead of the fused all-levels kernels
    KN_SWTF_LONG,          // 0: two-pass SWT for banks of more than 20 taps instead of the run-time-tap-count fused level kernels
    KN_F64_TAIL,           // double-precision 2-D: levels of at most this many pixels per side run in ONE launch per direction (0 = off)
    KN_EXP0,               // experimental knobs of the round (meaning: see the code that reads them)
    KN_EXP1,
    KN_EXP2,
    KN_EXP3,
    KN_NONSEP_TILED,       // custom non-separable banks (nonsep.hip): 1 = LDS-tiled kernels for levels that fill the chip, 0 = one-thread-per-output kernels, 2 = tiled at every size
    KN_F64_LAT,            // 0: direct-form level kernels (dwt_lds.hip) also for the orthogonal double-precision banks that have a lattice table (dwt_lat.hip)
    KN_F64_LAT_MIN,        // ... smallest level side (pixels) the lattice level kernels take
    KN_DWT1D_NT_MB,        // batched 1-D, float32: smallest image (MB) whose fused kernels read their rows / bands with non-temporal loads (0 = never)
    KN_COUNT
};
int knob(KnobId id);
// The hand-counted `s_waitcnt vmcnt(N)` pipelines (stream_dev.hpp) rely on undocumented ordering of a wave's loads and stores, EXEC = 0
// stores included: verified ONCE per device, on the first use of such a kernel (selfcheck.hip; ~10 ms), cached; false = the check found
// stale registers there -> callers take the compiler-counted kernels.
bool counted_waits_ok();
void stat_casc_spec(int inverse);  // (test statistics: a wave-program kernel was launched)
int knob_set(const char* name, int value);  // PDWT_OK / PDWT_EINVAL (unknown name)
int knob_get(const char* name, int* value);

// ---- size rule --------------------------------------------------------------------------------
// ceil-half: reference w_div2, src/utils.cu:24-27
__host__ __device__ inline int div2(int n) { return (n + 1) >> 1; }
inline int idiv_up(int a, int b) { return (a + b - 1) / b; }

// ---- periodic index helpers (device) ------------------------------------------------------------
// Index into a length-n line after the virtual "repeat last sample when n is odd" extension,
// then periodisation (reference src/separable.cu:116-121, SURVEY A-1).  Full modulo: equals the
// reference's single wrap whenever that is valid and stays in-bounds for any tile overhang.
__device__ __forceinline__ int wrap_ext(int s, int n)
{
    const int np = n + (n & 1);
    if ((unsigned)s >= (unsigned)np) {
        s %= np;
        if (s < 0) s += np;
    }
    return (s == n) ? n - 1 : s;
}
// plain periodic index (synthesis and a-trous passes, src/separable.cu:270-273,430-433)
__device__ __forceinline__ int wrap_per(int s, int n)
{
    if ((unsigned)s >= (unsigned)n) {
        s %= n;
        if (s < 0) s += n;
    }
    return s;
}

// Two filters of a bank passed BY VALUE in the kernarg segment: the taps arrive through scalar
// loads (SGPRs / scalar cache), no __constant__ symbol, no per-process global state (SURVEY B-1).
template <typename T>
struct Taps2 {
    T a[PDWT_MAX_FILTER_WIDTH];
    T b[PDWT_MAX_FILTER_WIDTH];
};

template <typename T> struct FiltersOf;
template <> struct FiltersOf<float> { using type = pdwt_filters_f32; };
template <> struct FiltersOf<double> { using type = pdwt_filters_f64; };

template <typename T>
inline Taps2<T> taps_fwd(const typename FiltersOf<T>::type* f)
{
    Taps2<T> t;
    for (int i = 0; i < PDWT_MAX_FILTER_WIDTH; i++) { t.a[i] = f->L[i]; t.b[i] = f->H[i]; }
    return t;
}
template <typename T>
inline Taps2<T> taps_inv(const typename FiltersOf<T>::type* f, T scale = T(1))
{
    Taps2<T> t;
    for (int i = 0; i < PDWT_MAX_FILTER_WIDTH; i++) { t.a[i] = f->IL[i] * scale; t.b[i] = f->IH[i] * scale; }
    return t;
}

template <typename T> __device__ __forceinline__ T fma_t(T a, T b, T c);
template <> __device__ __forceinline__ float fma_t<float>(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
template <> __device__ __forceinline__ double fma_t<double>(double a, double b, double c) { return __builtin_fma(a, b, c); }

// band-size bookkeeping shared by coeffs.hip / utils
struct BandGeom {
    int nbands;
    int Nr[3 * 32 + 1], Nc[3 * 32 + 1];
    size_t alloc_elems[3 * 32 + 1];  // allocation size (band 0 is level-1 sized)
};
int band_geometry(const pdwt_info& w, BandGeom* g);  // PDWT_OK / PDWT_EINVAL

}  // namespace pdwt
