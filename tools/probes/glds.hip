// probe of global_load_lds_dwordx4 addressing (not product code)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned lds_offset(const void* p) { return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) char*)p; }
__device__ __forceinline__ void glds16(const float* gsrc, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// variant tests: mode 1 = two DMAs back to back (pieces w and w+4), vmcnt(0); mode 2 = DMA, then 4 stores, wait vmcnt(4) only
__global__ void k2(const float* in, float* out, float* junk, int mode, int iters)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* smem = (float*)smem_raw;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lb = lds_offset(smem);
    int bad = 0;
    for (int it = 0; it < iters; it++) {
        for (int i = tid; i < 4096; i += 256) smem[i] = -1.f;
        __syncthreads();
        const float* src = in + (size_t)(blockIdx.x * iters + it) * 2048;
        if (mode == 1) {
            glds16(src + 4 * (w * 64 + lane), __builtin_amdgcn_readfirstlane(lb + (unsigned)w * 1024u));
            glds16(src + 4 * ((w + 4) * 64 + lane), __builtin_amdgcn_readfirstlane(lb + (unsigned)(w + 4) * 1024u));
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            glds16(src + 4 * (w * 64 + lane), __builtin_amdgcn_readfirstlane(lb + (unsigned)w * 1024u));
            glds16(src + 4 * ((w + 4) * 64 + lane), __builtin_amdgcn_readfirstlane(lb + (unsigned)(w + 4) * 1024u));
            float* jp = junk + ((size_t)(blockIdx.x * iters + it) * 256 + tid) * 16;
            v4f d = {1.f, 2.f, 3.f, (float)it};
            for (int q = 0; q < 4; q++) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(jp + 4 * q), "v"(d) : "memory");
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        for (int i = tid; i < 2048; i += 256) if (smem[i] != src[i]) bad++;
        __syncthreads();
    }
    if (bad) atomicAdd((int*)out, bad);
}
__global__ void k(const float* in, float* out, int* info)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* smem = (float*)smem_raw;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 4096; i += 256) smem[i] = -1.f;
    __syncthreads();
    const unsigned lb = lds_offset(smem);
    if (tid == 0) info[0] = (int)lb;
    // wave w copies 64 chunks of 16 B: global chunks (w*64 + lane) reversed within the wave, to LDS piece w
    glds16(in + 4 * (w * 64 + (63 - lane)), __builtin_amdgcn_readfirstlane(lb + 4096u + (unsigned)w * 1024u));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = tid; i < 4096; i += 256) out[i] = smem[i];
}
int main()
{
    float *in, *out; int* info;
    hipMalloc(&in, 4096 * 4); hipMalloc(&out, 4096 * 4); hipMalloc(&info, 16);
    float h[4096]; for (int i = 0; i < 4096; i++) h[i] = (float)i;
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    k<<<1, 256, 16384>>>(in, out, info);
    float o[4096]; int hi[4];
    hipMemcpy(o, out, sizeof(o), hipMemcpyDeviceToHost); hipMemcpy(hi, info, 16, hipMemcpyDeviceToHost);
    printf("lds base offset %d\n", hi[0]);
    int firstnz = -1; for (int i = 0; i < 4096; i++) if (o[i] != -1.f) { firstnz = i; break; }
    printf("first written float index %d (expected 1024)\n", firstnz);
    for (int wv = 0; wv < 4; wv++) { printf("wave %d piece: ", wv); for (int i = 0; i < 12; i++) printf("%g ", o[1024 + wv * 256 + i]); printf(" ... last4: %g %g %g %g\n", o[1024 + wv*256 + 252], o[1024+wv*256+253], o[1024+wv*256+254], o[1024+wv*256+255]); }
    int bad = 0; for (int wv = 0; wv < 4; wv++) for (int l = 0; l < 64; l++) for (int q = 0; q < 4; q++) if (o[1024 + wv * 256 + l * 4 + q] != (float)(4 * (wv * 64 + 63 - l) + q)) bad++;
    printf("mismatches vs expected lane-linear mapping: %d\n", bad);
    {
        const int nb = 1024, iters = 64;
        float *big, *junk; int* cnt;
        hipMalloc(&big, (size_t)nb * iters * 2048 * 4); hipMalloc(&junk, (size_t)nb * iters * 256 * 16 * 4); hipMalloc(&cnt, 4);
        float* hb = (float*)malloc((size_t)nb * iters * 2048 * 4);
        for (size_t i = 0; i < (size_t)nb * iters * 2048; i++) hb[i] = (float)(i % 1000003);
        hipMemcpy(big, hb, (size_t)nb * iters * 2048 * 4, hipMemcpyHostToDevice);
        for (int mode = 1; mode <= 2; mode++) {
            hipMemset(cnt, 0, 4);
            k2<<<nb, 256, 16384>>>(big, (float*)cnt, junk, mode, iters);
            int c; hipMemcpy(&c, cnt, 4, hipMemcpyDeviceToHost);
            printf("mode %d: mismatching floats %d of %zu\n", mode, c, (size_t)nb * iters * 2048);
        }
    }
    return 0;
}
