// collective.hip -- the one exchange step of the batch split from a SINGLE host process: all-reduce(SUM) of one double per device over
// RCCL (xGMI between the GPUs of a node).  SURVEY.md 8(e): the batch shards with no data-path collective; norm1() of the whole batch is
// the sum of the per-shard partial sums.  Processes-per-GPU deployments do this through torch.distributed (pdwt_amd/batch.py: backend
// "nccl" = RCCL); this is the counterpart for include/wt_batch.h, where one process drives all devices: ncclCommInitAll over the
// shards' devices (once per device set, cached) and one grouped ncclAllReduce on the library streams.
// RCCL is loaded at run time (dlopen): the kernel library has no link-time dependency on it, and a box without it -- or a device list
// RCCL cannot take (repeated devices) -- reports PDWT_ENOTSUP so that the caller adds the doubles on the host instead.
// Reference: none (the reference is single-GPU, TODO.txt:15).
#include <dlfcn.h>

#include <map>
#include <mutex>
#include <vector>

#include "common.hpp"

// The few RCCL declarations this file needs, stated here instead of #include <rccl/rccl.h>: the library is a run-time option (dlopen),
// so its development headers must not be a BUILD requirement of libpdwt_hip.so.  Values are those of the stable NCCL 2.x ABI
// (rccl.h: ncclSuccess = 0, ncclInvalidArgument = 4, ncclSum = 0, ncclFloat64 = 8); tests/test_cabi_symbols.py checks them, and the
// function-pointer signatures below, against the header when the image has one.
typedef struct ncclComm* ncclComm_t;
typedef int ncclResult_t;
typedef int ncclDataType_t;
typedef int ncclRedOp_t;
static constexpr ncclResult_t ncclSuccess = 0, ncclInvalidArgument = 4;
static constexpr ncclRedOp_t ncclSum = 0;
static constexpr ncclDataType_t ncclDouble = 8;

namespace pdwt {
namespace {
struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};
RcclApi& rccl()
{
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            api.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (api.lib) break;
        }
        if (!api.lib) return;
        api.CommInitAll = (decltype(api.CommInitAll))dlsym(api.lib, "ncclCommInitAll");
        api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.lib, "ncclCommDestroy");
        api.AllReduce = (decltype(api.AllReduce))dlsym(api.lib, "ncclAllReduce");
        api.GroupStart = (decltype(api.GroupStart))dlsym(api.lib, "ncclGroupStart");
        api.GroupEnd = (decltype(api.GroupEnd))dlsym(api.lib, "ncclGroupEnd");
        api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.lib, "ncclGetErrorString");
        api.ok = api.CommInitAll && api.CommDestroy && api.AllReduce && api.GroupStart && api.GroupEnd;
    });
    return api;
}
std::mutex g_comm_mu;
// one communicator set per device list (kept for the life of the process); an EMPTY vector records a device list ncclCommInitAll refused,
// so that every later norm1() of that batch takes the host sum at once instead of paying for the init again
std::map<std::vector<int>, std::vector<ncclComm_t>> g_comms;
}  // namespace
}  // namespace pdwt

using namespace pdwt;

extern "C" {
int pdwt_rccl_available(void) { return rccl().ok ? 1 : 0; }

// in[i], out[i]: device pointers on devices[i] (one double each; may alias).  Enqueued on each device's library stream, then the
// value of device 0 is copied to *result (synchronises that stream only: every device holds the same sum once its stream gets there).
int pdwt_rccl_allreduce_sum_f64(int n, const int* devices, const double* const* in, double* const* out, double* result)
{
    if (n < 1 || n > 64 || !devices || !in || !out || !result) return PDWT_EINVAL;
    RcclApi& api = rccl();
    if (!api.ok) return PDWT_ENOTSUP;
    std::vector<int> devs(devices, devices + n);
    for (int i = 0; i < n; i++) {
        if (!in[i] || !out[i]) return PDWT_EINVAL;
        for (int j = 0; j < i; j++)
            if (devs[i] == devs[j]) return PDWT_ENOTSUP;  // a communicator has one rank per device
    }
    int prev = 0;
    PDWT_HIP_TRY(hipGetDevice(&prev));
    std::lock_guard<std::mutex> lk(g_comm_mu);
    auto it = g_comms.find(devs);
    if (it == g_comms.end()) {
        std::vector<ncclComm_t> comms((size_t)n);
        const ncclResult_t r = api.CommInitAll(comms.data(), n, devs.data());
        (void)hipSetDevice(prev);
        if (r != ncclSuccess) {
            set_last_error(hipErrorUnknown, api.GetErrorString ? api.GetErrorString(r) : "ncclCommInitAll", __FILE__, __LINE__);
            g_comms.emplace(devs, std::vector<ncclComm_t>());
            return PDWT_ENOTSUP;
        }
        it = g_comms.emplace(devs, comms).first;
    }
    if (it->second.empty()) return PDWT_ENOTSUP;  // (refused before)
    ncclResult_t r = api.GroupStart();
    for (int i = 0; i < n && r == ncclSuccess; i++) {
        if (hipSetDevice(devs[i]) != hipSuccess) {
            r = ncclInvalidArgument;
            break;
        }
        r = api.AllReduce(in[i], out[i], 1, ncclDouble, ncclSum, it->second[(size_t)i], stream());
    }
    const ncclResult_t re = api.GroupEnd();
    if (r == ncclSuccess) r = re;
    int rc = PDWT_OK;
    if (r != ncclSuccess) {
        set_last_error(hipErrorUnknown, api.GetErrorString ? api.GetErrorString(r) : "ncclAllReduce", __FILE__, __LINE__);
        rc = PDWT_EHIP;
    } else if (hipSetDevice(devs[0]) != hipSuccess || hipMemcpyAsync(result, out[0], sizeof(double), hipMemcpyDeviceToHost, stream()) != hipSuccess ||
               hipStreamSynchronize(stream()) != hipSuccess) {
        rc = PDWT_EHIP;
    }
    (void)hipSetDevice(prev);
    return rc;
}
}
