// f64_mix.hip -- what does a scalar / LDS / vector-move instruction between FP64 FMAs cost a wave on gfx950?  64 v_fmac_f64 (16 chains) per iteration,
// plus K extra instructions of one kind spread between them; 1 and 2 waves per SIMD.  Time relative to the FMA-only stream.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/f64_mix tools/probes/f64_mix.hip && tools/probes/f64_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

// KIND 0: nothing, 1: s_mov_b64 (SALU), 2: v_mov_b64 (VALU move), 3: s_nop 0, 4: ds_read_b64 (LDS), 5: s_waitcnt lgkmcnt(0) (satisfied)
template <int KIND, int EVERY>
__global__ __launch_bounds__(256) void k_f(double* out, int iters, double seed, const double* tabp)
{
    __shared__ double lds[512];
    lds[threadIdx.x] = seed;
    __syncthreads();
    const double a = seed + threadIdx.x * 1e-9;
    double f[16], x[16];
    const double s0 = tabp[0];
    double mv = a;
    unsigned long long sm = 1;
    const unsigned la = threadIdx.x * 8;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        f[i] = a + i;
        x[i] = a * (i + 1);
    }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int i = 0; i < 16; i++) {
                asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(f[i]) : "s"(s0), "v"(x[(i + r) & 15]));
                if (KIND && ((r * 16 + i) % EVERY) == 0) {
                    if (KIND == 1) asm volatile("s_mov_b64 %0, %0" : "+s"(sm));
                    if (KIND == 2) asm volatile("v_mov_b64 %0, %0" : "+v"(mv));
                    if (KIND == 3) asm volatile("s_nop 0");
                    if (KIND == 4) asm volatile("ds_read_b64 %0, %1" : "=v"(mv) : "v"(la));
                    if (KIND == 5) asm volatile("s_waitcnt lgkmcnt(0)");
                }
            }
        if (KIND == 4) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(mv));
    }
    double s = mv + (double)sm;
#pragma unroll
    for (int i = 0; i < 16; i++) s += f[i];
    if (s == 12345.678) out[0] = s;
}

template <int KIND, int EVERY>
static double run(int wgs, int iters, double* out, const double* tab)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    std::vector<float> ms;
    for (int rep = 0; rep < 5; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k_f<KIND, EVERY>), dim3(wgs), dim3(256), 0, 0, out, iters, 1.0 + rep, tab);
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
    printf("64 v_fmac_f64 per iteration + extra instructions; ms (relative to FMA only)\n| waves/SIMD | FMA only | +32 s_mov_b64 | +16 s_mov_b64 | +32 v_mov_b64 | +32 s_nop | +16 ds_read_b64 | +32 s_waitcnt |\n|---|---|---|---|---|---|---|---|\n");
    for (int wpc : {1, 2}) {
        const int wgs = 256 * wpc;
        double t0 = run<0, 1>(wgs, iters, out, tab);
        double t[6] = {run<1, 2>(wgs, iters, out, tab), run<1, 4>(wgs, iters, out, tab), run<2, 2>(wgs, iters, out, tab), run<3, 2>(wgs, iters, out, tab), run<4, 4>(wgs, iters, out, tab), run<5, 2>(wgs, iters, out, tab)};
        printf("| %d | %.2f |", wpc, t0);
        for (int m = 0; m < 6; m++) printf(" %.2f (%.2f) |", t[m], t[m] / t0);
        printf("\n");
    }
    return 0;
}
