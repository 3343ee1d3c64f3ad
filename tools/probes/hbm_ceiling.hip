// hbm_ceiling.hip -- first-party bandwidth ceilings of this box (not product code): what a hand-written kernel streams once the
// working set has left the 256 MiB Infinity Cache.  bench.py prices every kernel against the 8 TB/s of the specification; this probe
// says what read-only / write-only / copy kernels reach at 64 MB ... 8 GB, with the cache policies a kernel can ask for, so that a
// fraction "of what the memory system gives" can stand next to it (profiles/r04_hbm_ceiling.md).
//
//   hipcc --offload-arch=gfx950 -O3 -o hbm_ceiling hbm_ceiling.hip && ./hbm_ceiling [max_GB]
//
// Every kernel moves 16 bytes per lane and instruction; U independent accesses are in flight per lane; the grid is persistent
// (workgroups-per-CU x 256 CUs) and walks the buffer with a grid stride, or -- "chunk" mode -- every workgroup owns one contiguous chunk.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>
#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e = (x);                                                        \
        if (e != hipSuccess) {                                                     \
            printf("err %s line %d\n", hipGetErrorString(e), __LINE__);            \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));
enum Pol { DEF = 0, NT = 1, SC1 = 2, SC01 = 3 };

template <int P>
__device__ __forceinline__ v4f ld(const v4f* p)
{
    v4f d;
    if constexpr (P == DEF) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d) : "v"(p) : "memory");
    if constexpr (P == NT) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(d) : "v"(p) : "memory");
    if constexpr (P == SC1) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(d) : "v"(p) : "memory");
    if constexpr (P == SC01) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(d) : "v"(p) : "memory");
    return d;
}
template <int P>
__device__ __forceinline__ void st(v4f* p, v4f d)
{
    if constexpr (P == DEF) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(d) : "memory");
    if constexpr (P == NT) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(d) : "memory");
    if constexpr (P == SC1) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(d) : "memory");
    if constexpr (P == SC01) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(d) : "memory");
}
__device__ __forceinline__ void waitall() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// walk: 0 = grid stride (workgroup b takes 4 KiB blocks b, b + G, ...), 1 = one contiguous chunk per workgroup
template <int P, int U>
__global__ __launch_bounds__(256) void k_read(const v4f* __restrict__ in, float* __restrict__ sink, size_t n4, int walk)
{
    const size_t G = gridDim.x, per = (n4 / 256 + G - 1) / G;  // 4 KiB blocks per workgroup
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    for (size_t k = 0; k < per; k += U) {
        v4f r[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t blk = walk ? blockIdx.x * per + k + u : (k + u) * G + blockIdx.x;
            const size_t i = blk * 256 + threadIdx.x;
            r[u] = ld<P>(in + (i < n4 ? i : threadIdx.x));
        }
        waitall();
#pragma unroll
        for (int u = 0; u < U; u++) acc += r[u];
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;  // (never true: keeps the loads alive)
}
template <int P, int U>
__global__ __launch_bounds__(256) void k_write(v4f* __restrict__ out, size_t n4, int walk)
{
    const size_t G = gridDim.x, per = (n4 / 256 + G - 1) / G;
    const v4f d = {1.f, 2.f, 3.f, (float)threadIdx.x};
    for (size_t k = 0; k < per; k += U) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t blk = walk ? blockIdx.x * per + k + u : (k + u) * G + blockIdx.x;
            const size_t i = blk * 256 + threadIdx.x;
            if (i < n4) st<P>(out + i, d);
        }
    }
}
template <int PL, int PS, int U>
__global__ __launch_bounds__(256) void k_copy(const v4f* __restrict__ in, v4f* __restrict__ out, size_t n4, int walk)
{
    const size_t G = gridDim.x, per = (n4 / 256 + G - 1) / G;
    for (size_t k = 0; k < per; k += U) {
        v4f r[U];
        size_t idx[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t blk = walk ? blockIdx.x * per + k + u : (k + u) * G + blockIdx.x;
            idx[u] = blk * 256 + threadIdx.x;
            r[u] = ld<PL>(in + (idx[u] < n4 ? idx[u] : threadIdx.x));
        }
        waitall();
#pragma unroll
        for (int u = 0; u < U; u++)
            if (idx[u] < n4) st<PS>(out + idx[u], r[u]);
    }
}
// the row pattern of the 2-D kernels: a wave reads 1 KiB of a row, rows are `pitch` bytes apart, R rows per wave, 4 strips per workgroup
template <int P, int U>
__global__ __launch_bounds__(256) void k_read_rows(const v4f* __restrict__ in, float* __restrict__ sink, int nr, int nc4, int R)
{
    const int lane = threadIdx.x & 63, strip = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int x = strip * 64 + lane;
    if (x >= nc4) return;
    const int y0 = blockIdx.y * R;
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    for (int q = 0; q < R; q += U) {
        v4f r[U];
#pragma unroll
        for (int u = 0; u < U; u++) r[u] = ld<P>(in + (size_t)min(y0 + q + u, nr - 1) * nc4 + x);
        waitall();
#pragma unroll
        for (int u = 0; u < U; u++) acc += r[u];
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;
}

template <typename F>
static float timeit(F f, int reps)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    f();
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
    const double maxgb = argc > 1 ? atof(argv[1]) : 8.0;
    int ncu = 256;
    hipDeviceProp_t pr;
    CK(hipGetDeviceProperties(&pr, 0));
    ncu = pr.multiProcessorCount;
    printf("# %s, %d CUs; TB/s = (bytes read + bytes written) / median time of %s launches\n", pr.name, ncu, "5-9");
    const size_t maxb = (size_t)(maxgb * (1ull << 30));
    v4f *a, *b;
    float* sink;
    CK(hipMalloc(&a, maxb));
    CK(hipMalloc(&b, maxb));
    CK(hipMalloc(&sink, 256));
    CK(hipMemset(a, 1, maxb));
    CK(hipMemset(b, 2, maxb));
    const char* pn[4] = {"default", "nt", "sc1", "sc0sc1"};
    for (size_t bytes : {(size_t)64 << 20, (size_t)256 << 20, (size_t)1 << 30, (size_t)2 << 30, (size_t)4 << 30, (size_t)8 << 30}) {
        if (bytes > maxb) continue;
        const size_t n4 = bytes / 16;
        const int reps = bytes >= ((size_t)2 << 30) ? 5 : 9;
        printf("\n## buffer %zu MiB\n", bytes >> 20);
        for (int walk = 0; walk < 2; walk++)
            for (int wpc : {4, 8, 16}) {  // workgroups per CU in the grid (<= 8 resident at 256 threads)
                const int grid = ncu * wpc;
                float r0 = timeit([&] { k_read<DEF, 8><<<grid, 256>>>(a, sink, n4, walk); }, reps);
                float r1 = timeit([&] { k_read<NT, 8><<<grid, 256>>>(a, sink, n4, walk); }, reps);
                float w0 = timeit([&] { k_write<DEF, 8><<<grid, 256>>>(b, n4, walk); }, reps);
                float w1 = timeit([&] { k_write<NT, 8><<<grid, 256>>>(b, n4, walk); }, reps);
                float c0 = timeit([&] { k_copy<DEF, DEF, 8><<<grid, 256>>>(a, b, n4, walk); }, reps);
                float c1 = timeit([&] { k_copy<NT, NT, 8><<<grid, 256>>>(a, b, n4, walk); }, reps);
                float c2 = timeit([&] { k_copy<DEF, NT, 8><<<grid, 256>>>(a, b, n4, walk); }, reps);
                printf("%s grid %5d: read %.2f (nt %.2f)  write %.2f (nt %.2f)  copy %.2f (nt/nt %.2f, ld default st nt %.2f) TB/s\n",
                       walk ? "chunk " : "stride", grid, bytes / r0 / 1e6, bytes / r1 / 1e6, bytes / w0 / 1e6, bytes / w1 / 1e6, 2.0 * bytes / c0 / 1e6,
                       2.0 * bytes / c1 / 1e6, 2.0 * bytes / c2 / 1e6);
            }
        // accesses in flight per lane and the remaining policies, grid stride, 8 workgroups per CU
        {
            const int grid = ncu * 8;
            float u2 = timeit([&] { k_copy<DEF, DEF, 2><<<grid, 256>>>(a, b, n4, 0); }, reps);
            float u4 = timeit([&] { k_copy<DEF, DEF, 4><<<grid, 256>>>(a, b, n4, 0); }, reps);
            float u16 = timeit([&] { k_copy<DEF, DEF, 16><<<grid, 256>>>(a, b, n4, 0); }, reps);
            printf("copy, accesses in flight per lane 2 / 4 / 16: %.2f / %.2f / %.2f TB/s\n", 2.0 * bytes / u2 / 1e6, 2.0 * bytes / u4 / 1e6, 2.0 * bytes / u16 / 1e6);
            float s1 = timeit([&] { k_copy<SC1, SC1, 8><<<grid, 256>>>(a, b, n4, 0); }, reps);
            float s2 = timeit([&] { k_copy<SC01, SC01, 8><<<grid, 256>>>(a, b, n4, 0); }, reps);
            float rs1 = timeit([&] { k_read<SC1, 8><<<grid, 256>>>(a, sink, n4, 0); }, reps);
            float ws1 = timeit([&] { k_write<SC1, 8><<<grid, 256>>>(b, n4, 0); }, reps);
            printf("policy %s: copy %.2f  read %.2f  write %.2f;  %s: copy %.2f TB/s\n", pn[SC1], 2.0 * bytes / s1 / 1e6, bytes / rs1 / 1e6, bytes / ws1 / 1e6, pn[SC01],
                   2.0 * bytes / s2 / 1e6);
        }
        // the row pattern of the 2-D kernels on an image of 4096 floats per row (16 KiB pitch)
        {
            const int nc4 = 1024, nr = (int)(n4 / nc4);
            for (int R : {16, 64, 256}) {
                dim3 g(nc4 / 256, (nr + R - 1) / R);
                float rr = timeit([&] { k_read_rows<DEF, 8><<<g, 256>>>(a, sink, nr, nc4, R); }, reps);
                printf("row pattern (1 KiB per wave and row, 16 KiB pitch) R = %3d rows per wave, %6d waves: read %.2f TB/s\n", R, g.x * g.y * 4, bytes / rr / 1e6);
            }
        }
    }
    return 0;
}
