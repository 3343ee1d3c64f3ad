// How much dynamic LDS can a 256-thread workgroup ask for and still share a CU with a second one?  (MI355X: 160 KiB per CU)
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/ldsocc tools/probes/ldsocc.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256, 2) void spin(double* out, int iters)
{
    extern __shared__ double lds[];
    double a = threadIdx.x, b = 1.0000001;
    for (int i = 0; i < iters; i++) a = __builtin_fma(a, b, 1e-9);
    lds[threadIdx.x] = a;
    __syncthreads();
    if (a == 12345.0) out[blockIdx.x] = lds[(threadIdx.x + 1) & 255];
}
int main()
{
    double* d;
    hipMalloc(&d, 1 << 20);
    hipFuncSetAttribute((const void*)spin, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    const int sizes[] = {32768, 65536, 73728, 77824, 79872, 80640, 81408, 81920, 82688, 98304, 163840};
    for (int s : sizes) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL(spin, dim3(512), dim3(256), s, 0, d, 20000);
        hipEventRecord(e0);
        hipLaunchKernelGGL(spin, dim3(512), dim3(256), s, 0, d, 20000);
        hipEventRecord(e1);
        hipError_t err = hipDeviceSynchronize();
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("lds %6d B: %.1f us (%s)\n", s, ms * 1e3, hipGetErrorString(err));
    }
    return 0;
}
