// phase_sep.hip -- does separating the READ phase of a streaming kernel from its WRITE phase beat a mixed copy on this box?
// (not product code)  profiles/r04_hbm_ceiling.md: out of the Infinity Cache a pure read runs at 6.1-7.0 TB/s, a
// pure write at 5.6-5.9, a mixed copy at 5.1-5.4 -- a time-separated read phase + write phase would average ~6.4 if the memory
// system rewarded it.  This probe copies a buffer with a persistent grid (one workgroup of 1024 threads per CU, one contiguous chunk
// per workgroup) that stages K bytes per workgroup in LDS per round:
//   mode 0  "local":  every workgroup alternates read K / write K on its own (phases of different CUs drift apart);
//   mode 1  "chip":   a grid-wide barrier (one atomic counter, agent scope) after every read phase and after every write phase, the
//                     stores drained (vmcnt 0) before the barrier: the whole chip reads, then the whole chip writes;
//   mode 2  "chip, no drain": the same barriers without waiting for the stores to be acknowledged.
// K = 4 ... 128 KiB per workgroup = 1 ... 32 MiB per phase chip-wide.  Reference lines from the same run: read-only, write-only and the
// plain chunk-walk copy of tools/probes/hbm_ceiling.hip.
//
//   hipcc --offload-arch=gfx950 -O3 -o phase_sep phase_sep.hip && ./phase_sep [GiB]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>
#define CK(x)                                                           \
    do {                                                                \
        hipError_t e = (x);                                             \
        if (e != hipSuccess) {                                          \
            printf("err %s line %d\n", hipGetErrorString(e), __LINE__); \
            exit(1);                                                    \
        }                                                               \
    } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));

// (compiler-visible loads: an inline-asm load "defines" its register at once as far as hipcc knows -- it then reuses the destination, e.g. as
// the address register of the next load, while the load is still in flight: a first version of this file faulted that way)
template <bool NT>
__device__ __forceinline__ v4f ld(const v4f* p)
{
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}
__device__ __forceinline__ void st(v4f* p, v4f d) { asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(d) : "memory"); }
__device__ __forceinline__ void waitall() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }  // (stores only: loads are the compiler's)

// grid barrier: the counter only grows; barrier number b (1-based) is passed when it has reached b * G
__device__ __forceinline__ void grid_barrier(unsigned* cnt, unsigned target)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        // (bounded: a workgroup that is not resident would otherwise hang the box; cnt[1] counts the give-ups)
        int spins = 0;
        while (__hip_atomic_load(cnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1 << 22)) {
                __hip_atomic_fetch_add(cnt + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    __syncthreads();
}

// U = 16-byte accesses in flight per lane inside a phase
template <bool NT, int U>
__global__ __launch_bounds__(1024) void k_phase(const v4f* __restrict__ in, v4f* __restrict__ out, size_t n4, int kel /* 16-B elements per round */,
                                                 unsigned* cnt, int mode)
{
    extern __shared__ v4f lds[];
    const size_t G = gridDim.x, per = n4 / G;  // (the host makes n4 a multiple of G * kel)
    const size_t base = blockIdx.x * per;
    unsigned bar = 0;
    for (size_t off = 0; off < per; off += kel) {
        for (int j = threadIdx.x; j < kel; j += 1024 * U) {
            v4f r[U];
#pragma unroll
            for (int u = 0; u < U; u++)
                if (j + u * 1024 < kel) r[u] = ld<NT>(in + base + off + j + u * 1024);
            waitall();
#pragma unroll
            for (int u = 0; u < U; u++)
                if (j + u * 1024 < kel) lds[j + u * 1024] = r[u];
        }
        if (mode) grid_barrier(cnt, (++bar) * (unsigned)G);
        else __syncthreads();
        for (int j = threadIdx.x; j < kel; j += 1024) st(out + base + off + j, lds[j]);
        if (mode == 1) waitall();
        if (mode) grid_barrier(cnt, (++bar) * (unsigned)G);
        else __syncthreads();
    }
}
// the reference lines (chunk walk, 8 in flight; cf. hbm_ceiling.hip)
template <bool NT>
__global__ __launch_bounds__(256) void k_read(const v4f* __restrict__ in, float* __restrict__ sink, size_t n4)
{
    const size_t G = gridDim.x, per = (n4 / 256 + G - 1) / G;
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    for (size_t k = 0; k < per; k += 8) {
        v4f r[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const size_t i = (blockIdx.x * per + k + u) * 256 + threadIdx.x;
            r[u] = ld<NT>(in + ((k + u < per && i < n4) ? i : threadIdx.x));
        }
        waitall();
#pragma unroll
        for (int u = 0; u < 8; u++) acc += r[u];
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;
}
__global__ __launch_bounds__(256) void k_write(v4f* __restrict__ out, size_t n4)
{
    const size_t G = gridDim.x, per = (n4 / 256 + G - 1) / G;
    const v4f d = {1.f, 2.f, 3.f, (float)threadIdx.x};
    for (size_t k = 0; k < per; k++) {
        const size_t i = (blockIdx.x * per + k) * 256 + threadIdx.x;
        if (i < n4) st(out + i, d);
    }
}
template <bool NT>
__global__ __launch_bounds__(256) void k_copy(const v4f* __restrict__ in, v4f* __restrict__ out, size_t n4)
{
    const size_t G = gridDim.x, per = (n4 / 256 + G - 1) / G;
    for (size_t k = 0; k < per; k += 8) {
        v4f r[8];
        size_t idx[8];
        bool on[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            idx[u] = (blockIdx.x * per + k + u) * 256 + threadIdx.x;
            on[u] = (k + u < per) && idx[u] < n4;
            r[u] = ld<NT>(in + (on[u] ? idx[u] : threadIdx.x));
        }
        waitall();
#pragma unroll
        for (int u = 0; u < 8; u++)
            if (on[u]) st(out + idx[u], r[u]);
    }
}

template <typename F>
static float timeit(F f, int reps)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    f();
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    std::vector<float> t;
    for (int i = 0; i < reps; i++) {
        CK(hipEventRecord(e0));
        f();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
    return t[t.size() / 2] * 1e3f;  // median, us
}

int main(int argc, char** argv)
{
    setvbuf(stdout, NULL, _IONBF, 0);
    const double gib = argc > 1 ? atof(argv[1]) : 2.0;
    const int only = argc > 2 ? atoi(argv[2]) : -1;  // (debug: 0 = reference lines only, 1 = phase kernels only)
    hipDeviceProp_t pr;
    CK(hipGetDeviceProperties(&pr, 0));
    const int ncu = pr.multiProcessorCount;
    printf("# %s, %d CUs; TB/s = (bytes read + bytes written) / median launch time\n", pr.name, ncu);
    CK(hipFuncSetAttribute((const void*)k_phase<false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)k_phase<true, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)k_phase<false, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)k_phase<true, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const size_t maxb = (size_t)(gib * (1ull << 30));
    v4f *a, *b;
    float* sink;
    unsigned* cnt;
    CK(hipMalloc(&a, maxb));
    CK(hipMalloc(&b, maxb));
    CK(hipMalloc(&sink, 256));
    CK(hipMalloc(&cnt, 256));
    CK(hipMemset(cnt, 0, 256));
    CK(hipMemset(a, 1, maxb));
    CK(hipMemset(b, 2, maxb));
    for (size_t bytes : {(size_t)256 << 20, (size_t)1 << 30, (size_t)2 << 30, (size_t)4 << 30}) {
        if (bytes > maxb) continue;
        const size_t n4 = bytes / 16;
        const int reps = 7;
        printf("\n## buffer %zu MiB (x2: source + destination)\n", bytes >> 20);
        if (only != 1) {
            const int grid = ncu * 8;
            const float r0 = timeit([&] { k_read<false><<<grid, 256>>>(a, sink, n4); }, reps);
            const float r1 = timeit([&] { k_read<true><<<grid, 256>>>(a, sink, n4); }, reps);
            const float w0 = timeit([&] { k_write<<<grid, 256>>>(b, n4); }, reps);
            const float c0 = timeit([&] { k_copy<false><<<grid, 256>>>(a, b, n4); }, reps);
            const float c1 = timeit([&] { k_copy<true><<<grid, 256>>>(a, b, n4); }, reps);
            printf("reference: read %.2f (nt %.2f)  write %.2f  copy %.2f (nt loads %.2f) TB/s;  read then write, back to back launches: %.2f TB/s\n", bytes / r0 / 1e6,
                   bytes / r1 / 1e6, bytes / w0 / 1e6, 2.0 * bytes / c0 / 1e6, 2.0 * bytes / c1 / 1e6, 2.0 * bytes / (std::min(r0, r1) + w0) / 1e6);
        }
        printf("| K per workgroup | per phase, chip | local | local nt | chip (drain) | chip nt (drain) | chip (no drain) | chip nt (no drain) |\n|---|---|---|---|---|---|---|---|\n");
        if (only == 0) continue;
        for (int kkb : {4, 8, 16, 32, 64, 128}) {
            const int kel = kkb * 1024 / 16;
            const int grid = ncu;  // one workgroup of 1024 threads per CU: every workgroup is resident (the grid barrier needs that)
            if (n4 % ((size_t)grid * kel)) continue;
            const size_t lds = (size_t)kkb * 1024;
            float t[6];
            int q = 0;
            for (int mode : {0, 1, 2})
                for (int nt = 0; nt < 2; nt++) {
                    t[q++] = timeit(
                        [&] {
                            CK(hipMemsetAsync(cnt, 0, 4, 0));  // (cnt[1], the give-up count, is kept)
                            if (nt) k_phase<true, 8><<<grid, 1024, lds>>>(a, b, n4, kel, cnt, mode);
                            else k_phase<false, 8><<<grid, 1024, lds>>>(a, b, n4, kel, cnt, mode);
                        },
                        reps);
                }
            printf("| %d KiB | %.0f MiB | %.2f | %.2f | %.2f | %.2f | %.2f | %.2f |\n", kkb, (double)kkb * grid / 1024.0, 2.0 * bytes / t[0] / 1e6, 2.0 * bytes / t[1] / 1e6,
                   2.0 * bytes / t[2] / 1e6, 2.0 * bytes / t[3] / 1e6, 2.0 * bytes / t[4] / 1e6, 2.0 * bytes / t[5] / 1e6);
            fflush(stdout);
        }
    }
    unsigned h[2] = {0, 0};
    CK(hipMemcpy(h, cnt, 8, hipMemcpyDeviceToHost));
    printf("\ngrid-barrier give-ups (must be 0): %u\n", h[1]);
    return 0;
}
