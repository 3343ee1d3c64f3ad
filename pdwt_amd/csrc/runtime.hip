// runtime.hip -- device / memory / stream / event plumbing of the C-ABI (include/pdwt_hip.h).
// Replaces the bare CUDA runtime calls of the reference's class (src/wt.cu:117-130,421-468,543-549).
#include <set>
#include <utility>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "common.hpp"

namespace pdwt {

static thread_local char g_err[512] = "";

void set_last_error(hipError_t e, const char* what, const char* file, int line)
{
    snprintf(g_err, sizeof(g_err), "%s: %s (%s:%d)", hipGetErrorString(e), what, file, line);
    (void)hipGetLastError();  // clear the sticky error
}

// One stream per device, created on first use.  It is a BLOCKING stream (hipStreamDefault): it keeps the legacy
// ordering against the NULL stream in both directions, which is what a program written for the reference relies on
// (the reference runs everything on the NULL stream, src/wt.cu: bare kernel launches and cudaMemcpy) -- its own
// NULL-stream kernels that fill an image before set_image(..., 1) or read d_image / d_coeffs after forward() are
// ordered without any call.  PDWT_STREAM_NONBLOCKING=1 opts out (the caller then orders with pdwt_sync / events);
// pdwt_set_stream() makes the library enqueue on a stream of the caller instead (e.g. torch's current stream).
static std::mutex g_mu;
static hipStream_t g_streams[64] = {};
static hipStream_t g_user_streams[64] = {};
static bool g_user_set[64] = {};

hipStream_t stream()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_user_set[dev]) return g_user_streams[dev];
    if (!g_streams[dev]) {
        static const bool nonblocking = getenv("PDWT_STREAM_NONBLOCKING") && atoi(getenv("PDWT_STREAM_NONBLOCKING")) == 1;
        if (hipStreamCreateWithFlags(&g_streams[dev], nonblocking ? hipStreamNonBlocking : hipStreamDefault) != hipSuccess) {
            g_streams[dev] = nullptr;
        }
    }
    return g_streams[dev];
}

// ---- > 64 KB of dynamic LDS: opt-in once per (kernel, device) --------------------------------------
int lds_opt_in_ptr(const void* kernel)
{
    static std::mutex mu;
    static std::set<std::pair<const void*, int>> done;
    int dev = 0;
    PDWT_HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    if (done.count({kernel, dev})) return PDWT_OK;
    PDWT_HIP_TRY(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    done.insert({kernel, dev});
    return PDWT_OK;
}

// ---- knobs: environment read once, pdwt_debug_set at run time ---------------------------------------
struct KnobDef { const char* name; const char* env; int dflt; };
static const KnobDef g_knob_defs[KN_COUNT] = {
    {"force_twopass", "PDWT_FORCE_TWOPASS", 0}, {"tiled_cols", "PDWT_TILED_COLS", 0},
    {"casc", "PDWT_CASC", 1}, {"casc_waves", "PDWT_CASC_WAVES", 0}, {"casc_nv", "PDWT_CASC_NV", 0},
    {"casc_min", "PDWT_CASC_MIN", 2048 * 2048}, {"casc_iwaves", "PDWT_CASC_IWAVES", 0}, {"casc_ipfd", "PDWT_CASC_IPFD", 1},
    {"casc_wg", "PDWT_CASC_WG", 0}, {"casc_iwg", "PDWT_CASC_IWG", 0}, {"casc_l3", "PDWT_CASC_L3", 1}, {"casc_spec", "PDWT_CASC_SPEC", 7}, {"casc_xcdw", "PDWT_CASC_XCDW", 0}, {"dwt1d_lds_kb", "PDWT_DWT1D_LDS_KB", 80},
    {"stream", "PDWT_STREAM", 1}, {"stream_r", "PDWT_STREAM_R", 0}, {"stream_waves", "PDWT_STREAM_WAVES", 8192},
    {"stream_narrow", "PDWT_STREAM_NARROW", 2048 * 2048}, {"small", "PDWT_SMALL", 1},
    {"rows_tr", "PDWT_ROWS_TR", 1}, {"ring_r", "PDWT_RING_R", 0}, {"ring_waves", "PDWT_RING_WAVES", 4096},
    {"swtf", "PDWT_SWTF", 1}, {"swtf_m", "PDWT_SWTF_M", 0}, {"swtf_mi", "PDWT_SWTF_MI", 0},
    {"swtf_xcd", "PDWT_SWTF_XCD", 1}, {"swtf_alt", "PDWT_SWTF_ALT", 1}, {"swtf_perm", "PDWT_SWTF_PERM", 1}, {"swtf_f64", "PDWT_SWTF_F64", 1},
    {"f64_lds", "PDWT_F64_LDS", 1}, {"f64_lds_min", "PDWT_F64_LDS_MIN", 256}, {"f64_lds_wgs", "PDWT_F64_LDS_WGS", 512}, {"f64_lds_mingroups", "PDWT_F64_LDS_MINGROUPS", 1}, {"f64_lds_skew", "PDWT_F64_LDS_SKEW", 64}, {"norm2sq_ref1d", "PDWT_NORM2SQ_REF1D", 0},
    {"norm_in_threshold", "PDWT_NORM_IN_THRESHOLD", -1},
    {"selfcheck", "PDWT_SELFCHECK", 1}, {"dwt1d_f64", "PDWT_DWT1D_F64", 1}, {"swtf_long", "PDWT_SWTF_LONG", 1}, {"f64_tail", "PDWT_F64_TAIL", 0},
    {"exp0", "PDWT_EXP0", 0}, {"exp1", "PDWT_EXP1", 0}, {"exp2", "PDWT_EXP2", 0}, {"exp3", "PDWT_EXP3", 0}, {"nonsep_tiled", "PDWT_NONSEP_TILED", 1},
    {"f64_lat", "PDWT_F64_LAT", 1}, {"f64_lat_min", "PDWT_F64_LAT_MIN", 4096}, {"dwt1d_nt_mb", "PDWT_DWT1D_NT_MB", 192},
};
static int g_knob_vals[KN_COUNT];
static std::once_flag g_knob_once;
static void knob_init()
{
    for (int i = 0; i < KN_COUNT; i++) {
        const char* e = getenv(g_knob_defs[i].env);
        g_knob_vals[i] = (e && *e) ? atoi(e) : g_knob_defs[i].dflt;
    }
}
int knob(KnobId id)
{
    std::call_once(g_knob_once, knob_init);
    return g_knob_vals[id];
}
int knob_set(const char* name, int value)
{
    std::call_once(g_knob_once, knob_init);
    if (!name) return PDWT_EINVAL;
    for (int i = 0; i < KN_COUNT; i++) {
        if (!strcmp(name, g_knob_defs[i].name)) {
            g_knob_vals[i] = value;
            return PDWT_OK;
        }
    }
    return PDWT_EINVAL;
}

// launch statistics for tests (pdwt_debug_get("stat_casc_spec_fwd" / "stat_casc_spec_inv"): launches of the straight-line wave-program
// kernels since the process started)
std::atomic<int> g_stat_spec_fwd{0}, g_stat_spec_inv{0};
void stat_casc_spec(int inverse) { (inverse ? g_stat_spec_inv : g_stat_spec_fwd).fetch_add(1, std::memory_order_relaxed); }
// ("stat_lat_fwd" / "stat_lat_inv": launches of the lattice level kernels of dwt_lat.hip)
std::atomic<int> g_stat_lat_fwd{0}, g_stat_lat_inv{0};
void stat_lat(int inverse) { (inverse ? g_stat_lat_inv : g_stat_lat_fwd).fetch_add(1, std::memory_order_relaxed); }

int knob_get(const char* name, int* value)
{
    std::call_once(g_knob_once, knob_init);
    if (!name || !value) return PDWT_EINVAL;
    if (!strcmp(name, "stat_casc_spec_fwd") || !strcmp(name, "stat_casc_spec_inv")) {
        *value = (name[15] == 'f' ? g_stat_spec_fwd : g_stat_spec_inv).load(std::memory_order_relaxed);
        return PDWT_OK;
    }
    if (!strcmp(name, "stat_lat_fwd") || !strcmp(name, "stat_lat_inv")) {
        *value = (name[9] == 'f' ? g_stat_lat_fwd : g_stat_lat_inv).load(std::memory_order_relaxed);
        return PDWT_OK;
    }
    for (int i = 0; i < KN_COUNT; i++) {
        if (!strcmp(name, g_knob_defs[i].name)) {
            *value = g_knob_vals[i];
            return PDWT_OK;
        }
    }
    return PDWT_EINVAL;
}

// ---- in-kernel clock probe ------------------------------------------------------------------------
static unsigned long long* g_probe[64] = {};
static unsigned long long* g_probe_allbuf[64] = {};
static bool g_probe_on = false;
static int g_probe_all = 0;  // 0 = off, 1 = forward launches, 2 = inverse launches record every workgroup
unsigned long long* clock_probe_all(int* all)
{
    *all = 0;
    if (!g_probe_all) return nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    if (!g_probe_allbuf[dev] && hipMalloc(&g_probe_allbuf[dev], (size_t)kClockProbeAllBlocks * 4 * sizeof(unsigned long long)) != hipSuccess) return nullptr;
    *all = g_probe_all;
    return g_probe_allbuf[dev];
}
unsigned long long* clock_probe_slot(int slot)
{
    if (!g_probe_on || slot < 0 || slot >= 16) return nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    if (!g_probe[dev]) {
        if (hipMalloc(&g_probe[dev], 16 * 4 * sizeof(unsigned long long)) != hipSuccess) return nullptr;
        (void)hipMemsetAsync(g_probe[dev], 0, 16 * 4 * sizeof(unsigned long long), stream());
    }
    return g_probe[dev] + 4 * slot;
}

// ---- per-kernel timing ---------------------------------------------------------------------
static const char* const g_knames[K_COUNT] = {
    "fwd2d_fused", "inv2d_fused", "ana_rows", "ana_cols", "syn_cols", "syn_rows",
    "swt_ana_rows", "swt_ana_cols", "swt_syn_cols", "swt_syn_rows",
    "haar2d_fwd", "haar2d_inv", "haar1d_fwd", "haar1d_inv", "soft_thresh", "abs_sum", "abs_sum_final",
    "fwd2d_casc", "inv2d_casc", "fwd2d_stream", "inv2d_stream", "fwd2d_small", "inv2d_small", "fwd2d_f64", "inv2d_f64", "thresh_sum",
};
struct KRec { int id; hipEvent_t e0, e1; };
static thread_local bool g_kt_on = false;
static thread_local std::vector<KRec>* g_kt = nullptr;
static thread_local std::vector<hipEvent_t>* g_evpool = nullptr;

static hipEvent_t ev_get()
{
    if (g_evpool && !g_evpool->empty()) {
        hipEvent_t e = g_evpool->back();
        g_evpool->pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

KTimer::KTimer(int kernel_id, bool attach_to_launch) : id(kernel_id), ext(attach_to_launch)
{
    if (!g_kt_on) return;
    e0 = ev_get();
    e1 = ev_get();
    if (e0 && !ext) (void)hipEventRecord(e0, stream());
}
KTimer::~KTimer()
{
    if (!e0 || !e1) return;
    if (!ext) (void)hipEventRecord(e1, stream());
    if (!g_kt) g_kt = new std::vector<KRec>();
    g_kt->push_back(KRec{id, e0, e1});
}

}  // namespace pdwt

using namespace pdwt;

extern "C" {

int pdwt_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}
int pdwt_set_device(int dev)
{
    PDWT_HIP_TRY(hipSetDevice(dev));
    return PDWT_OK;
}
int pdwt_get_device(void)
{
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return PDWT_EHIP;
    return dev;
}
int pdwt_device_name(char* buf, int buflen)
{
    if (!buf || buflen <= 0) return PDWT_EINVAL;
    int dev = 0;
    PDWT_HIP_TRY(hipGetDevice(&dev));
    hipDeviceProp_t p;
    PDWT_HIP_TRY(hipGetDeviceProperties(&p, dev));
    snprintf(buf, (size_t)buflen, "%s (%s, %d CUs)", p.name, p.gcnArchName, p.multiProcessorCount);
    return PDWT_OK;
}
void* pdwt_malloc(size_t nbytes)
{
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, nbytes ? nbytes : 1);
    if (e != hipSuccess) {
        set_last_error(e, "hipMalloc", __FILE__, __LINE__);
        return nullptr;
    }
    return p;
}
int pdwt_free(void* dptr)
{
    if (!dptr) return PDWT_OK;
    PDWT_HIP_TRY(hipStreamSynchronize(stream()));
    PDWT_HIP_TRY(hipFree(dptr));
    return PDWT_OK;
}
int pdwt_memset(void* dptr, int byte, size_t nbytes)
{
    if (!nbytes) return PDWT_OK;
    PDWT_HIP_TRY(hipMemsetAsync(dptr, byte, nbytes, stream()));
    return PDWT_OK;
}
int pdwt_memcpy_h2d(void* dst, const void* src, size_t nbytes)
{
    if (!nbytes) return PDWT_OK;
    PDWT_HIP_TRY(hipMemcpyAsync(dst, src, nbytes, hipMemcpyHostToDevice, stream()));
    PDWT_HIP_TRY(hipStreamSynchronize(stream()));
    return PDWT_OK;
}
int pdwt_memcpy_d2h(void* dst, const void* src, size_t nbytes)
{
    if (!nbytes) return PDWT_OK;
    PDWT_HIP_TRY(hipMemcpyAsync(dst, src, nbytes, hipMemcpyDeviceToHost, stream()));
    PDWT_HIP_TRY(hipStreamSynchronize(stream()));
    return PDWT_OK;
}
int pdwt_memcpy_d2d(void* dst, const void* src, size_t nbytes)
{
    if (!nbytes) return PDWT_OK;
    PDWT_HIP_TRY(hipMemcpyAsync(dst, src, nbytes, hipMemcpyDeviceToDevice, stream()));
    return PDWT_OK;
}
int pdwt_memcpy_d2d_foreign(void* dst, const void* src, size_t nbytes)
{
    if (!nbytes) return PDWT_OK;
    // the producer of `src` is unknown: wait for the NULL stream (and every blocking stream with it) first, copy on the
    // library stream, and wait for the copy so that the caller may reuse or free `src` at once
    PDWT_HIP_TRY(hipStreamSynchronize(nullptr));
    PDWT_HIP_TRY(hipMemcpyAsync(dst, src, nbytes, hipMemcpyDeviceToDevice, stream()));
    PDWT_HIP_TRY(hipStreamSynchronize(stream()));
    return PDWT_OK;
}
int pdwt_sync(void)
{
    PDWT_HIP_TRY(hipStreamSynchronize(stream()));
    return PDWT_OK;
}
void* pdwt_get_stream(void) { return (void*)stream(); }
int pdwt_set_stream(void* user_stream, int use_it)
{
    int dev = 0;
    PDWT_HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return PDWT_EINVAL;
    // whatever is still queued on the stream used so far must not be overtaken by work on the new one
    PDWT_HIP_TRY(hipStreamSynchronize(stream()));
    std::lock_guard<std::mutex> lk(g_mu);
    g_user_set[dev] = use_it != 0;
    g_user_streams[dev] = use_it ? (hipStream_t)user_stream : nullptr;
    return PDWT_OK;
}
const char* pdwt_last_error_string(void) { return g_err; }

void* pdwt_event_create(void)
{
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return (void*)e;
}
int pdwt_event_record(void* ev)
{
    PDWT_HIP_TRY(hipEventRecord((hipEvent_t)ev, stream()));
    return PDWT_OK;
}
int pdwt_event_sync(void* ev)
{
    PDWT_HIP_TRY(hipEventSynchronize((hipEvent_t)ev));
    return PDWT_OK;
}
float pdwt_event_elapsed_ms(void* a, void* b)
{
    float ms = -1.f;
    if (hipEventElapsedTime(&ms, (hipEvent_t)a, (hipEvent_t)b) != hipSuccess) return -1.f;
    return ms;
}
int pdwt_event_destroy(void* ev)
{
    PDWT_HIP_TRY(hipEventDestroy((hipEvent_t)ev));
    return PDWT_OK;
}

// ---- launch-bound transforms: stream capture into a hipGraph ------------------------------------------------
// A small multi-level transform is a handful of 4-5 us launches whose cost is the CPU enqueue, not the GPU (512^2 db4
// L3: 6 launches, 24 us per pair of which < 8 us is kernel time).  The class can record the launches of one forward()
// or inverse() once and replay them as ONE graph launch (wt.cpp, opt-in: PDWT_GRAPH=1).
int pdwt_graph_allowed(void) { return (g_kt_on || g_probe_on) ? 0 : 1; }  // per-kernel event timing / the clock probe (allocates on first use) and capture do not mix
int pdwt_graph_capture_begin(void)
{
    // the self-check of the hand-counted waits cannot run inside a capture (allocations, a host read-back): take the device's verdict NOW,
    // so that the kernels recorded into the graph are the ones the check allows (a graph user's first transform is the captured one)
    (void)pdwt::counted_waits_ok();
    PDWT_HIP_TRY(hipStreamBeginCapture(pdwt::stream(), hipStreamCaptureModeThreadLocal));
    return PDWT_OK;
}
int pdwt_graph_capture_end(void** exec_out)
{
    if (!exec_out) return PDWT_EINVAL;
    *exec_out = nullptr;
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture(pdwt::stream(), &g);
    if (e != hipSuccess || !g) {
        (void)hipGetLastError();
        return PDWT_EHIP;
    }
    hipGraphExec_t x = nullptr;
    e = hipGraphInstantiate(&x, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess || !x) {
        (void)hipGetLastError();
        return PDWT_EHIP;
    }
    *exec_out = (void*)x;
    return PDWT_OK;
}
int pdwt_graph_launch(void* exec)
{
    if (!exec) return PDWT_EINVAL;
    PDWT_HIP_TRY(hipGraphLaunch((hipGraphExec_t)exec, pdwt::stream()));
    return PDWT_OK;
}
int pdwt_graph_destroy(void* exec)
{
    if (exec) PDWT_HIP_TRY(hipGraphExecDestroy((hipGraphExec_t)exec));
    return PDWT_OK;
}

// ---- first-party bandwidth probe (bench.py: roofline.copy_ceiling; tools/probes/hbm_ceiling.hip is the full sweep) ----------------
// mode 0: copy src -> dst, 1: read src only, 2: write dst only.  16 bytes per lane and instruction, eight in flight per lane, every
// workgroup owns one contiguous chunk (the walk that streams best on this chip: profiles/r04_hbm_ceiling.md).  Runs on the library
// stream; the caller times it with pdwt_event_*.
namespace {
typedef float probe_v4f __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_bw_probe(const probe_v4f* __restrict__ in, probe_v4f* __restrict__ out, size_t n4, int mode)
{
    // 4 KiB blocks per workgroup (rounded up: the last workgroups of an uneven split find idx >= n4 and do nothing, and the tail of a
    // byte count that is not a multiple of 4 KiB is the last, partial block -- every 16-byte element is moved exactly once)
    const size_t G = gridDim.x, nblk = (n4 + 255) / 256, per = (nblk + G - 1) / G;
    probe_v4f acc = {0.f, 0.f, 0.f, 0.f};
    for (size_t k = 0; k < per; k += 8) {
        probe_v4f r[8];
        size_t idx[8];
        bool on[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            idx[u] = (blockIdx.x * per + k + u) * 256 + threadIdx.x;
            on[u] = (k + u < per) && idx[u] < n4;  // (k + u >= per would be the NEXT workgroup's chunk: moved twice)
            if (mode != 2) r[u] = in[on[u] ? idx[u] : threadIdx.x];
            else r[u] = probe_v4f{1.f, 2.f, 3.f, (float)u};
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (mode == 1) acc += r[u];
            else if (on[u]) out[idx[u]] = r[u];
        }
    }
    if (mode == 1 && acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc;  // (never true: keeps the loads alive)
}
}  // namespace
int pdwt_probe_bandwidth(const void* src, void* dst, size_t bytes, int mode)
{
    if (mode < 0 || mode > 2 || bytes < 4096 || (mode != 2 && !src) || !dst) return PDWT_EINVAL;
    int dev = 0, ncu = 256;
    PDWT_HIP_TRY(hipGetDevice(&dev));
    PDWT_HIP_TRY(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
    hipLaunchKernelGGL(k_bw_probe, dim3((unsigned)(8 * ncu)), dim3(256), 0, pdwt::stream(), (const probe_v4f*)src, (probe_v4f*)dst, bytes / 16, mode);
    PDWT_HIP_TRY(hipGetLastError());
    return PDWT_OK;
}

int pdwt_clock_probe_enable(int on)
{
    g_probe_on = on == 1;
    g_probe_all = on >= 2 ? on - 1 : 0;  // 2: every workgroup of the forward launches, 3: of the inverse launches (diagnostic)
    return PDWT_OK;
}
int pdwt_clock_probe_dump(unsigned long long* out, int nblocks)
{
    int dev = 0;
    PDWT_HIP_TRY(hipGetDevice(&dev));
    if (!out || nblocks < 1 || nblocks > kClockProbeAllBlocks || dev < 0 || dev >= 64 || !g_probe_allbuf[dev]) return PDWT_EINVAL;
    return pdwt_memcpy_d2h(out, g_probe_allbuf[dev], (size_t)nblocks * 4 * sizeof(unsigned long long));
}
int pdwt_clock_probe_read(int slot, double* shader_mhz, double* span_us)
{
    if (slot < 0 || slot >= 16 || !shader_mhz || !span_us) return PDWT_EINVAL;
    int dev = 0;
    PDWT_HIP_TRY(hipGetDevice(&dev));
    *shader_mhz = *span_us = 0.0;
    if (dev < 0 || dev >= 64 || !g_probe[dev]) return PDWT_OK;
    unsigned long long h[4];
    const int rc = pdwt_memcpy_d2h(h, g_probe[dev] + 4 * slot, sizeof(h));
    if (rc != PDWT_OK) return rc;
    if (h[3] > h[1] && h[2] > h[0]) {
        const double ticks = (double)(h[3] - h[1]);  // 100 MHz
        *span_us = ticks / 100.0;
        *shader_mhz = (double)(h[2] - h[0]) / ticks * 100.0;
    }
    return PDWT_OK;
}
int pdwt_ktime_enable(int on)
{
    g_kt_on = on != 0;
    return PDWT_OK;
}
int pdwt_ktime_reset(void)
{
    if (g_kt) {
        if (!g_evpool) g_evpool = new std::vector<hipEvent_t>();
        (void)hipStreamSynchronize(stream());
        for (auto& r : *g_kt) {
            g_evpool->push_back(r.e0);
            g_evpool->push_back(r.e1);
        }
        g_kt->clear();
    }
    return PDWT_OK;
}
int pdwt_ktime_read(int kernel_id, int* n_launches, double* total_ms)
{
    if (kernel_id < 0 || kernel_id >= K_COUNT) return PDWT_EINVAL;
    PDWT_HIP_TRY(hipStreamSynchronize(stream()));
    int n = 0;
    double tot = 0;
    if (g_kt) {
        for (auto& r : *g_kt) {
            if (r.id != kernel_id) continue;
            float ms = 0;
            if (hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) {
                tot += ms;
                n++;
            }
        }
    }
    if (n_launches) *n_launches = n;
    if (total_ms) *total_ms = tot;
    return PDWT_OK;
}
const char* pdwt_kernel_name(int kernel_id) { return (kernel_id >= 0 && kernel_id < K_COUNT) ? g_knames[kernel_id] : nullptr; }
int pdwt_kernel_count(void) { return K_COUNT; }

}  // extern "C"
