// write-bandwidth probe (not product code): pure float4 stores, and the 1-read : 4-write mix of a forward SWT level
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(256) void k_write(float4* __restrict__ out, size_t n4, int per)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    const float4 v = make_float4(1.f, 2.f, 3.f, (float)threadIdx.x);
    for (int k = 0; k < per; k++, i += stride) if (i < n4) out[i] = v;
}
__global__ __launch_bounds__(256) void k_r1w4(const float4* __restrict__ in, float4* __restrict__ a, float4* __restrict__ b, float4* __restrict__ c, float4* __restrict__ d, size_t n4, int per)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    for (int k = 0; k < per; k++, i += stride) if (i < n4) { float4 v = in[i]; a[i] = v; v.x += 1.f; b[i] = v; v.y += 1.f; c[i] = v; v.z += 1.f; d[i] = v; }
}
template <typename F> float timeit(F f, int reps = 20) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) f();
    hipDeviceSynchronize(); hipEventRecord(e0);
    for (int i = 0; i < reps; i++) f();
    hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1e3f / reps;
}
int main() {
    const size_t N = (size_t)4096 * 4096;  // one band
    float4 *in, *o[16];
    hipMalloc(&in, N * 4);
    for (int i = 0; i < 16; i++) hipMalloc(&o[i], N * 4);
    hipMemset(in, 1, N * 4);
    for (int grid : {2048, 4096, 8192, 16384}) {
        int per = (int)((N / 4 + (size_t)grid * 256 - 1) / ((size_t)grid * 256));
        // rotate over 16 bands so that the working set (1 GB) exceeds the Infinity Cache
        int it = 0;
        float us = timeit([&] { k_write<<<grid, 256>>>(o[(it++) & 15], N / 4, per); }, 64);
        printf("write 64 MB (rotating over 1 GB) grid %5d: %.1f us  %.2f TB/s\n", grid, us, N * 4 / us / 1e6);
        it = 0;
        us = timeit([&] { int k = (it++) & 3; k_r1w4<<<grid, 256>>>(in, o[4 * k], o[4 * k + 1], o[4 * k + 2], o[4 * k + 3], N / 4, per); }, 32);
        printf("read 64 MB + write 4 x 64 MB grid %5d: %.1f us  %.2f TB/s\n", grid, us, 5.0 * N * 4 / us / 1e6);
    }
    return 0;
}
