// FP64 FMA issue rate per wave and per SIMD: v_fma_f64 (three VGPR/constant sources) and v_fmac_f64 with an SGPR tap, against
// the number of waves per SIMD (256-thread workgroups, launched as 256 * n workgroups = n waves per SIMD).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/f64clk.bin tools/probes/f64clk.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
template <int NACC, int MODE>
__global__ __launch_bounds__(256) void spin(double* out, long long* clk, int iters, const double* __restrict__ taps)
{
    double a[NACC], x[NACC];
#pragma unroll
    for (int k = 0; k < NACC; k++) {
        a[k] = threadIdx.x + k;
        x[k] = 1.0 + 1e-9 * (threadIdx.x + k);
    }
    double s0 = taps[0], s1 = taps[1];
    const double b = 1.0000001, c = 1e-9;
    long long t0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {
#pragma unroll
            for (int k = 0; k < NACC; k++) a[k] = __builtin_fma(a[k], b, c);
        } else {
#pragma unroll
            for (int k = 0; k < NACC; k++) a[k] = __builtin_fma(x[k], (k & 1) ? s1 : s0, a[k]);
            asm volatile("" : "+s"(s0), "+s"(s1));
        }
    }
    long long t1 = clock64(), w1 = wall_clock64();
    double s = 0;
#pragma unroll
    for (int k = 0; k < NACC; k++) s += a[k];
    if (s == 12345.0) out[blockIdx.x] = s;
    if (threadIdx.x == 0) {
        clk[2 * blockIdx.x] = t1 - t0;
        clk[2 * blockIdx.x + 1] = w1 - w0;
    }
}
template <int NACC, int MODE>
void run(double* d, long long* c, double* taps, int wgs)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int iters = 20000;
    float best = 1e9;
    std::vector<long long> h(2 * wgs);
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((spin<NACC, MODE>), dim3(wgs), dim3(256), 0, 0, d, c, iters, taps);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        best = std::min(best, ms);
    }
    (void)hipMemcpy(h.data(), c, sizeof(long long) * 2 * wgs, hipMemcpyDeviceToHost);
    std::vector<double> mhz, cyc;
    for (int i = 0; i < wgs; i++) {
        mhz.push_back(h[2 * i] / (h[2 * i + 1] / 100.0));
        cyc.push_back((double)h[2 * i] / ((double)iters * NACC));
    }
    std::sort(mhz.begin(), mhz.end());
    std::sort(cyc.begin(), cyc.end());
    const double flops = 2.0 * NACC * iters * 256.0 * wgs;
    printf("mode %d acc %2d wgs %4d: %8.1f us %5.1f TFLOP/s  clock med %.0f MHz  cycles per FMA per wave: min %.2f med %.2f max %.2f\n", MODE, NACC, wgs,
           best * 1e3, flops / (best * 1e-3) / 1e12, mhz[wgs / 2], cyc.front(), cyc[wgs / 2], cyc.back());
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
    for (int wgs : {256, 512, 1024}) {
        run<8, 0>(d, c, taps, wgs);
        run<8, 1>(d, c, taps, wgs);
    }
    // sustained: does the clock hold?  (~3 s of back-to-back launches, one line per ~0.3 s)
    for (int blk = 0; blk < 10; blk++) {
        for (int i = 0; i < 400; i++) hipLaunchKernelGGL((spin<8, 1>), dim3(512), dim3(256), 0, 0, d, c, 20000, taps);
        run<8, 1>(d, c, taps, 512);
    }
    return 0;
}
