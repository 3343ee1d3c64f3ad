// probe: are a wave's global loads and stores retired IN ORDER with respect to each other on gfx950, i.e. does
// `s_waitcnt vmcnt(N)` with N younger STORES outstanding guarantee that the older LOADS have landed?
// Each wave streams through a table whose element i holds f(i); per step: 2 loads (far apart: L2 misses), then NST stores,
// then vmcnt(NST), then the loaded registers are checked against f.  MODE 0: stores to a small (L2-resident) buffer -> fast acks;
// MODE 1: stores to a large buffer (memory).  Prints the number of stale registers seen.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float fval(size_t i) { return (float)(i % 1000003u); }

template <int NST, bool COUNTED, bool EXEC0 = false>
__global__ __launch_bounds__(256) void k(const float* __restrict__ tab, size_t n4, float* sink, size_t sink4, int steps, unsigned long long* bad)
{
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nthr = (size_t)gridDim.x * blockDim.x;
    v4f a = {0, 0, 0, 0}, b = {0, 0, 0, 0};
    unsigned long long nbad = 0;
    size_t i0 = gid % n4, i1 = (gid + n4 / 2) % n4;
    asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(a) : "v"(tab + 4 * i0) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(b) : "v"(tab + 4 * i1) : "memory");
    for (int s = 0; s < steps; s++) {
        if (s > 0) {
            if (COUNTED) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(NST) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b) : : "memory");
        } else asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b) : : "memory");
        v4f ca, cb;  // opaque copies (a plain copy may be coalesced with the load register: the next tied load then gets a fresh register)
        asm volatile("v_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7" : "=&v"(ca.x), "=&v"(ca.y), "=&v"(ca.z), "=&v"(ca.w) : "v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w));
        asm volatile("v_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7" : "=&v"(cb.x), "=&v"(cb.y), "=&v"(cb.z), "=&v"(cb.w) : "v"(b.x), "v"(b.y), "v"(b.z), "v"(b.w));
        const size_t c0 = i0, c1 = i1;
        i0 = (i0 + nthr) % n4;
        i1 = (i1 + nthr) % n4;
        asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(a) : "v"(tab + 4 * i0) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(b) : "v"(tab + 4 * i1) : "memory");
        // check the registers of the previous step
        nbad += (ca.x != fval(4 * c0)) + (ca.w != fval(4 * c0 + 3)) + (cb.x != fval(4 * c1)) + (cb.w != fval(4 * c1 + 3));
        const v4f o = ca + cb;
#pragma unroll
        for (int q = 0; q < NST; q++) {
            float* p = sink + 4 * ((gid + (size_t)(s * NST + q) * nthr) % sink4);
            if (EXEC0) {
                unsigned long long saved, zero = (s & 1) ? 0ull : ~0ull;  // every second step: all lanes off
                asm volatile("s_and_saveexec_b64 %0, %3\n\tglobal_store_dwordx4 %1, %2, off\n\ts_nop 1\n\ts_mov_b64 exec, %0" : "=&s"(saved) : "v"(p), "v"(o), "s"(zero) : "memory", "scc");
            } else
            asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(o) : "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b) : : "memory");
    if (nbad) atomicAdd(bad, nbad);
}

int main()
{
    const size_t n4 = (size_t)256 << 20 >> 2;  // 1 GiB of floats / 4
    float *tab, *sink_small, *sink_big;
    unsigned long long* bad;
    CK(hipMalloc(&tab, n4 * 16));
    CK(hipMalloc(&sink_small, 1 << 20));
    CK(hipMalloc(&sink_big, (size_t)2 << 30));
    CK(hipMalloc(&bad, 8));
    {
        float* h = (float*)malloc(n4 * 16);
        for (size_t i = 0; i < 4 * n4; i++) h[i] = (float)(i % 1000003u);
        CK(hipMemcpy(tab, h, n4 * 16, hipMemcpyHostToDevice));
        free(h);
    }
    auto run = [&](const char* name, auto kern, float* sink, size_t sink4) {
        unsigned long long z = 0, r = 0;
        hipMemcpy(bad, &z, 8, hipMemcpyHostToDevice);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(2048), dim3(256), 0, 0, tab, n4, sink, sink4, 400, bad);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(&r, bad, 8, hipMemcpyDeviceToHost);
        printf("%-44s stale registers: %llu   (%.2f ms)\n", name, r, ms);
        return 0;
    };
    run("vmcnt(0), 4 stores -> 1 MB", k<4, false>, sink_small, (1 << 20) / 16);
    run("vmcnt(4), 4 stores -> 1 MB (L2 hits)", k<4, true>, sink_small, (1 << 20) / 16);
    run("vmcnt(4), 4 stores -> 2 GB (memory)", k<4, true>, sink_big, ((size_t)2 << 30) / 16);
    run("vmcnt(1), 1 store  -> 1 MB", k<1, true>, sink_small, (1 << 20) / 16);
    run("vmcnt(1), 1 store  -> 2 GB", k<1, true>, sink_big, ((size_t)2 << 30) / 16);
    run("vmcnt(0), 4 stores -> 2 GB", k<4, false>, sink_big, ((size_t)2 << 30) / 16);
    run("vmcnt(4), 4 stores, EXEC = 0 every 2nd step", k<4, true, true>, sink_big, ((size_t)2 << 30) / 16);
    run("vmcnt(1), 1 store,  EXEC = 0 every 2nd step", k<1, true, true>, sink_small, (1 << 20) / 16);
    return 0;
}
