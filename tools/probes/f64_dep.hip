// f64_dep.hip -- how many independent v_fmac_f64 chains does one wave need on gfx950?  N accumulators, each FMA depends on the one N instructions
// earlier (v_fmac_f64_e32 acc, s, v: the form of the double-precision kernels).  1 and 2 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/f64_dep tools/probes/f64_dep.hip && tools/probes/f64_dep
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

template <int N>
__global__ __launch_bounds__(256) void k_f(double* out, int iters, double seed, const double* tabp)
{
    const double a = seed + threadIdx.x * 1e-9;
    double f[N], x[16];
    const double s0 = tabp[0];
#pragma unroll
    for (int i = 0; i < N; i++) f[i] = a + i;
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = a * (i + 1);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 64 / N; r++)
#pragma unroll
            for (int i = 0; i < N; i++) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(f[i]) : "s"(s0), "v"(x[(i + r) & 15]));
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < N; i++) s += f[i];
    if (s == 12345.678) out[0] = s;
}

template <int N>
static double run(int wgs, int iters, double* out, const double* tab)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    std::vector<float> ms;
    for (int rep = 0; rep < 7; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k_f<N>, dim3(wgs), dim3(256), 0, 0, out, iters, 1.0 + rep, tab);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float t;
        (void)hipEventElapsedTime(&t, e0, e1);
        ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    return ms[ms.size() / 2];
}

int main()
{
    double *out, *tab;
    (void)hipMalloc(&out, 64);
    (void)hipMalloc(&tab, 16 * 8);
    double h[16];
    for (int i = 0; i < 16; i++) h[i] = 1.0 - 1e-9 * (i + 1);
    (void)hipMemcpy(tab, h, sizeof(h), hipMemcpyHostToDevice);
    const int iters = 20000;
    printf("| waves per SIMD | 1 chain | 2 chains | 4 chains | 8 chains | 16 chains |  (ms, TFLOP/s)\n|---|---|---|---|---|---|\n");
    for (int wpc : {1, 2, 3, 4}) {
        const int wgs = 256 * wpc;
        const double flop = (double)wgs * 4 * iters * 64.0 * 128.0;
        double t[5] = {run<1>(wgs, iters, out, tab), run<2>(wgs, iters, out, tab), run<4>(wgs, iters, out, tab), run<8>(wgs, iters, out, tab), run<16>(wgs, iters, out, tab)};
        printf("| %d |", wpc);
        for (int m = 0; m < 5; m++) printf(" %.2f (%.1f) |", t[m], flop / t[m] / 1e9);
        printf("\n");
    }
    return 0;
}
