// f64_sgpr.hip -- what does a v_fma_f64 / v_fmac_f64 cost on gfx950 when one multiplicand is a SCALAR register pair (the taps of every
// double-precision kernel of this library arrive that way) against all-vector operands?  16 accumulators x 4 per iteration, 2 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/f64_sgpr tools/probes/f64_sgpr.hip && tools/probes/f64_sgpr
// mode 0: v_fma_f64 v, v, v, v   mode 1: v_fma_f64 v, s, v, v (VOP3)   mode 2: v_fmac_f64_e32 v, s, v (VOP2)   mode 3: v_fmac_f64_e32 v, v, v
// mode 4: v_fma_f64 v, -s, v, v with a DIFFERENT scalar pair per instruction (16 pairs)   mode 5: as 2 with 16 different scalar pairs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

template <int MODE>
__global__ __launch_bounds__(256) void k_f(double* out, int iters, double seed, const double* tabp)
{
    const double a = seed + threadIdx.x * 1e-9;
    double f[16], x[16];
    double sreg[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        f[i] = a + i;
        x[i] = a * (i + 1);
        sreg[i] = tabp[i];  // uniform -> scalar registers
    }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int i = 0; i < 16; i++) {
                if (MODE == 0) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(f[i]) : "v"(x[i]), "v"(x[(i + 1) & 15]));
                if (MODE == 1) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(f[i]) : "s"(sreg[0]), "v"(x[i]));
                if (MODE == 2) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(f[i]) : "s"(sreg[0]), "v"(x[i]));
                if (MODE == 3) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(f[i]) : "v"(x[(i + 1) & 15]), "v"(x[i]));
                if (MODE == 4) asm volatile("v_fma_f64 %0, -%1, %2, %0" : "+v"(f[i]) : "s"(sreg[i]), "v"(x[i]));
                if (MODE == 5) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(f[i]) : "s"(sreg[i]), "v"(x[i]));
            }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s += f[i];
    if (s == 12345.678) out[0] = s;
}

template <int MODE>
static double run(int wgs, int iters, double* out, const double* tab)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    std::vector<float> ms;
    for (int rep = 0; rep < 7; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_f<MODE>, dim3(wgs), dim3(256), 0, 0, out, iters, 1.0 + rep, tab);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float t;
        hipEventElapsedTime(&t, e0, e1);
        ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    return ms[ms.size() / 2];
}

int main()
{
    double *out, *tab;
    hipMalloc(&out, 64);
    hipMalloc(&tab, 16 * 8);
    double h[16];
    for (int i = 0; i < 16; i++) h[i] = 1.0 - 1e-9 * (i + 1);
    hipMemcpy(tab, h, sizeof(h), hipMemcpyHostToDevice);
    const int iters = 20000;
    printf("| waves per SIMD | vfma vvv | vfma svv | vfmac_e32 sv | vfmac_e32 vv | vfma -s(16 pairs) | vfmac s(16 pairs) |  (ms, TFLOP/s)\n|---|---|---|---|---|---|---|\n");
    for (int wpc : {1, 2}) {
        const int wgs = 256 * wpc;
        const double flop = (double)wgs * 4 * iters * 64.0 * 128.0;
        double t[6] = {run<0>(wgs, iters, out, tab), run<1>(wgs, iters, out, tab), run<2>(wgs, iters, out, tab), run<3>(wgs, iters, out, tab), run<4>(wgs, iters, out, tab), run<5>(wgs, iters, out, tab)};
        printf("| %d |", wpc);
        for (int m = 0; m < 6; m++) printf(" %.2f (%.1f) |", t[m], flop / t[m] / 1e9);
        printf("\n");
    }
    return 0;
}
