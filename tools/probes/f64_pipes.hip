// f64_pipes.hip -- does v_mfma_f64_16x16x4_f64 run beside v_fma_f64 on gfx950, and at what rate?  (VERDICT r4 item 2c asked for the
// column pass of long double-precision banks on the matrix pipe; this probe prices that pipe first.)
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/f64_pipes tools/probes/f64_pipes.hip && tools/probes/f64_pipes
// mode 1: MFMA only (4 independent accumulators per wave), mode 2: VALU only (64 v_fma_f64 per iteration on 16 accumulators: the same
// 1024 x 4 FMAs per iteration as the four MFMAs), mode 3: both in every wave, mode 4: even waves MFMA, odd waves VALU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k_pipes(double* out, int iters, double seed)
{
    const double a = seed + threadIdx.x * 1e-9, b = 1.0 - 1e-9;
    d4 c[4];
    double f[16];
#pragma unroll
    for (int i = 0; i < 4; i++) c[i] = d4{a, a, a, a};
#pragma unroll
    for (int i = 0; i < 16; i++) f[i] = a + i;
    const bool wave_mfma = ((threadIdx.x >> 6) & 1) == 0;
    for (int it = 0; it < iters; it++) {
        if (MODE == 1 || MODE == 3 || (MODE == 4 && wave_mfma)) {
#pragma unroll
            for (int i = 0; i < 4; i++) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[i], 0, 0, 0);
        }
        if (MODE == 2 || MODE == 3 || (MODE == 4 && !wave_mfma)) {
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int i = 0; i < 16; i++) f[i] = __builtin_fma(f[i], b, a);
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
#pragma unroll
    for (int i = 0; i < 16; i++) s += f[i];
    if (s == 12345.678) out[0] = s;
}

template <int MODE>
static double run(int wgs, int iters, double* out)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    std::vector<float> ms;
    for (int rep = 0; rep < 7; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_pipes<MODE>, dim3(wgs), dim3(256), 0, 0, out, iters, 1.0 + rep);
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
    double* out;
    hipMalloc(&out, 64);
    const int iters = 20000;
    printf("| workgroups (256 threads) | waves per SIMD | MFMA only ms (TFLOP/s) | VALU only ms (TFLOP/s) | both in every wave ms | even waves MFMA, odd VALU ms |\n|---|---|---|---|---|---|\n");
    for (int wpc : {1, 2, 4}) {
        const int wgs = 256 * wpc;
        const double waves = (double)wgs * 4;
        const double flop_mfma = waves * iters * 4.0 * 2048.0, flop_valu = waves * iters * 64.0 * 128.0;
        const double t1 = run<1>(wgs, iters, out), t2 = run<2>(wgs, iters, out), t3 = run<3>(wgs, iters, out), t4 = run<4>(wgs, iters, out);
        printf("| %d | %d | %.2f (%.1f) | %.2f (%.1f) | %.2f (sum of the two: %.2f) | %.2f |\n", wgs, wpc, t1, flop_mfma / t1 / 1e9, t2, flop_valu / t2 / 1e9, t3, t1 + t2, t4);
    }
    return 0;
}
