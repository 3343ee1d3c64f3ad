// Sustained FP64 FMA rate and shader clock against the LDS read traffic that feeds the FMAs (is a db20-sized kernel power-bound?).
// Each iteration: R ds_read_b128 (2 doubles each) + 16 v_fmac_f64 with SGPR taps; 512 workgroups of 256 threads = 2 waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/f64pow.bin tools/probes/f64pow.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double dbl2 __attribute__((ext_vector_type(2)));
template <int R>
__global__ __launch_bounds__(256, 2) void spin(double* out, long long* clk, int iters, const double* __restrict__ taps)
{
    __shared__ __attribute__((aligned(16))) double lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = 1.0 + 1e-9 * i;
    __syncthreads();
    double a[16];
#pragma unroll
    for (int k = 0; k < 16; k++) a[k] = threadIdx.x + k;
    dbl2 x[8];
#pragma unroll
    for (int k = 0; k < 8; k++) x[k] = dbl2{1.0 + 1e-9 * (threadIdx.x + k), 1.0 - 1e-9 * k};
    double s0 = taps[0], s1 = taps[1];
    const char* base = (const char*)lds + (threadIdx.x & 63) * 16;
    long long t0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; i++) {
        const char* p = base + ((i & 7) << 10);
#pragma unroll
        for (int r = 0; r < R; r++) x[r] = *reinterpret_cast<const dbl2*>(p + r * 1024 * 4);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            a[2 * k] = __builtin_fma(x[k].x, s0, a[2 * k]);
            a[2 * k + 1] = __builtin_fma(x[k].y, s1, a[2 * k + 1]);
        }
        asm volatile("" : "+s"(s0), "+s"(s1));
    }
    long long t1 = clock64(), w1 = wall_clock64();
    double s = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) s += a[k];
    if (s == 12345.0) out[blockIdx.x] = s;
    if (threadIdx.x == 0) {
        clk[2 * blockIdx.x] = t1 - t0;
        clk[2 * blockIdx.x + 1] = w1 - w0;
    }
}
template <int R>
void run(double* d, long long* c, double* taps)
{
    const int wgs = 512, iters = 20000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int blk = 0; blk < 4; blk++) {
        for (int i = 0; i < 300; i++) hipLaunchKernelGGL((spin<R>), dim3(wgs), dim3(256), 0, 0, d, c, iters, taps);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((spin<R>), dim3(wgs), dim3(256), 0, 0, d, c, iters, taps);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> h(2 * wgs);
        (void)hipMemcpy(h.data(), c, sizeof(long long) * 2 * wgs, hipMemcpyDeviceToHost);
        std::vector<double> mhz;
        for (int i = 0; i < wgs; i++) mhz.push_back(h[2 * i] / (h[2 * i + 1] / 100.0));
        std::sort(mhz.begin(), mhz.end());
        printf("b128 reads per 16 FMAs %d: %7.1f us  %5.1f TFLOP/s  clock med %.0f MHz  cycles per FMA per wave %.2f\n", R, ms * 1e3,
               2.0 * 16 * iters * 256.0 * wgs / (ms * 1e-3) / 1e12, mhz[wgs / 2], (double)h[0] / (16.0 * iters));
    }
}
int main()
{
    double *d, *taps;
    long long* c;
    (void)hipMalloc(&d, 1 << 20);
    (void)hipMalloc(&c, 1 << 20);
    (void)hipMalloc(&taps, 64);
    double ht[2] = {1.0000001, 0.9999999};
    (void)hipMemcpy(taps, ht, 16, hipMemcpyHostToDevice);
    run<0>(d, c, taps);
    run<1>(d, c, taps);
    run<2>(d, c, taps);
    run<4>(d, c, taps);
    run<8>(d, c, taps);
    return 0;
}
