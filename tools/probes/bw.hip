// bandwidth probes shaped like the level-1 DWT kernels (not product code)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

__global__ __launch_bounds__(256) void k_copy4(const float4* __restrict__ in, float4* __restrict__ out, size_t n4, int per)
{
    size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x);
    size_t stride = (size_t)gridDim.x * 256;
    for (int k = 0; k < per; k++, i += stride) if (i < n4) out[i] = in[i];
}
// fwd-shaped: read float4 rows (two rows -> one output row), write 4 bands of float2
__global__ __launch_bounds__(256) void k_fwdshape(const float4* __restrict__ in, float2* __restrict__ a, float2* __restrict__ h, float2* __restrict__ v, float2* __restrict__ d, int Nr, int Nc4, int R)
{
    int lane = threadIdx.x & 63, strip = blockIdx.x * 4 + (threadIdx.x >> 6);
    int x = strip * 64 + lane; if (x >= Nc4) return;
    int y0 = blockIdx.y * R;
    for (int q = 0; q < R; q += 4) {
        float4 r[8];
#pragma unroll
        for (int u = 0; u < 8; u++) r[u] = in[(size_t)(2 * (y0 + q) + u) * Nc4 + x];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            float4 s0 = r[2*u], s1 = r[2*u+1];
            size_t o = (size_t)(y0 + q + u) * Nc4 + x;
            a[o] = make_float2(s0.x + s1.x, s0.y + s1.y); h[o] = make_float2(s0.z + s1.z, s0.w + s1.w);
            v[o] = make_float2(s0.x - s1.x, s0.y - s1.y); d[o] = make_float2(s0.z - s1.z, s0.w - s1.w);
        }
    }
}
// inv-shaped: read 4 bands float2, write float4 rows
__global__ __launch_bounds__(256) void k_invshape(float4* __restrict__ out, const float2* __restrict__ a, const float2* __restrict__ h, const float2* __restrict__ v, const float2* __restrict__ d, int Nr, int Nc4, int R)
{
    int lane = threadIdx.x & 63, strip = blockIdx.x * 4 + (threadIdx.x >> 6);
    int x = strip * 64 + lane; if (x >= Nc4) return;
    int y0 = blockIdx.y * R;
    for (int q = 0; q < R; q += 4) {
        float2 A[4], H[4], V[4], D[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { size_t o = (size_t)(y0 + q + u) * Nc4 + x; A[u] = a[o]; H[u] = h[o]; V[u] = v[o]; D[u] = d[o]; }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            out[(size_t)(2 * (y0 + q + u)) * Nc4 + x] = make_float4(A[u].x + H[u].x, A[u].y + H[u].y, V[u].x + D[u].x, V[u].y + D[u].y);
            out[(size_t)(2 * (y0 + q + u) + 1) * Nc4 + x] = make_float4(A[u].x - H[u].x, A[u].y - H[u].y, V[u].x - D[u].x, V[u].y - D[u].y);
        }
    }
}
template <typename F> float timeit(F f, int reps = 30) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) f();
    std::vector<float> t;
    for (int i = 0; i < reps; i++) { hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); t.push_back(ms); }
    std::sort(t.begin(), t.end()); return t[t.size() / 2] * 1e3f;
}
int main() {
    const int N = 4096; size_t n = (size_t)N * N;
    float *in, *out, *b[4];
    CK(hipMalloc(&in, n * 4)); CK(hipMalloc(&out, n * 4));
    for (int i = 0; i < 4; i++) CK(hipMalloc(&b[i], n));
    hipMemset(in, 1, n * 4);
    for (int grid : {1024, 2048, 4096, 8192, 16384}) {
        int per = (int)((n / 4 + (size_t)grid * 256 - 1) / ((size_t)grid * 256));
        float us = timeit([&] { k_copy4<<<grid, 256>>>((float4*)in, (float4*)out, n / 4, per); });
        printf("copy4 64MB->64MB grid %5d per %3d: %.1f us  %.2f TB/s (r+w)\n", grid, per, us, 2.0 * n * 4 / us / 1e6);
    }
    for (int R : {4, 8, 16, 32, 64}) {
        dim3 g(N / 4 / 256, N / 2 / R);
        float us = timeit([&] { k_fwdshape<<<g, 256>>>((float4*)in, (float2*)b[0], (float2*)b[1], (float2*)b[2], (float2*)b[3], N, N / 4, R); });
        printf("fwdshape R=%2d waves %5d: %.1f us  %.2f TB/s\n", R, g.x * g.y * 4, us, 2.0 * n * 4 / us / 1e6);
        float us2 = timeit([&] { k_invshape<<<g, 256>>>((float4*)out, (float2*)b[0], (float2*)b[1], (float2*)b[2], (float2*)b[3], N, N / 4, R); });
        printf("invshape R=%2d waves %5d: %.1f us  %.2f TB/s\n", R, g.x * g.y * 4, us2, 2.0 * n * 4 / us2 / 1e6);
    }
    // small levels: same shapes at 2048 and 1024
    for (int M : {2048, 1024}) for (int R : {4, 8, 16}) {
        dim3 g((M / 4 + 255) / 256, M / 2 / R);
        float us = timeit([&] { k_fwdshape<<<g, 256>>>((float4*)in, (float2*)b[0], (float2*)b[1], (float2*)b[2], (float2*)b[3], M, M / 4, R); });
        printf("fwdshape M=%d R=%2d waves %5d: %.1f us  %.2f TB/s\n", M, R, g.x * g.y * 4, us, 2.0 * M * M * 4 / us / 1e6);
    }
    // empty kernel launch + back-to-back dependent tiny kernels
    float us = timeit([&] { k_copy4<<<1, 256>>>((float4*)in, (float4*)out, 256, 1); });
    printf("tiny kernel: %.2f us\n", us);
    us = timeit([&] { for (int i = 0; i < 6; i++) k_copy4<<<1, 256>>>((float4*)in, (float4*)out, 256, 1); });
    printf("6 tiny kernels: %.2f us\n", us);
    return 0;
}
