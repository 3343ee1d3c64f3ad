// probe: (1) does v_pk_fma_f32 broadcast either half of an SGPR pair through op_sel / op_sel_hi?  (2) issue cost of the VALU
// instructions the cascade kernels are made of (cycles per instruction per wave, one wave per SIMD and four)
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
typedef float v2f __attribute__((ext_vector_type(2)));

__global__ void k_opsel(const float* sp, float* out)
{
    // the scalar pair lives in SGPRs (uniform load)
    const v2f s = v2f{sp[0], sp[1]};
    v2f s_u;
    s_u.x = __builtin_amdgcn_readfirstlane(s.x);
    s_u.y = __builtin_amdgcn_readfirstlane(s.y);
    v2f v = v2f{(float)threadIdx.x, (float)threadIdx.x + 100.f};
    v2f r0 = {0, 0}, r1 = {0, 0}, r2 = {0, 0};
    asm volatile("v_pk_fma_f32 %0, %1, %2, 0 op_sel_hi:[1,1,0]" : "=v"(r0) : "v"(v), "s"(s_u));                   // (v.x*s.x, v.y*s.y)
    asm volatile("v_pk_fma_f32 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(v), "s"(s_u));    // low half broadcast: (v.x*s.x, v.y*s.x)
    asm volatile("v_pk_fma_f32 %0, %1, %2, 0 op_sel:[0,1,0] op_sel_hi:[1,1,0]" : "=v"(r2) : "v"(v), "s"(s_u));    // high half broadcast: (v.x*s.y, v.y*s.y)
    float* o = out + threadIdx.x * 6;
    o[0] = r0.x; o[1] = r0.y; o[2] = r1.x; o[3] = r1.y; o[4] = r2.x; o[5] = r2.y;
}

template <int MODE>
__global__ __launch_bounds__(256) void k_rate(float* out, int iters, const float* sp, long long* cyc)
{
    v2f a = {1.f + threadIdx.x, 2.f}, b = {0.5f, 0.25f}, c = {1.f, 1.f}, d = {2.f, 2.f}, e = {3.f, 3.f}, g = {4.f, 4.f};
    v2f s; s.x = __builtin_amdgcn_readfirstlane(sp[0]); s.y = __builtin_amdgcn_readfirstlane(sp[1]);
    float x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3;
    long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            if (MODE == 0) {  // packed FMA, VGPR x SGPR pair
                asm volatile("v_pk_fma_f32 %0, %0, %4, %0\n\tv_pk_fma_f32 %1, %1, %4, %1\n\tv_pk_fma_f32 %2, %2, %4, %2\n\tv_pk_fma_f32 %3, %3, %4, %3" : "+v"(c), "+v"(d), "+v"(e), "+v"(g) : "s"(s));
            } else if (MODE == 1) {  // scalar FMA
                asm volatile("v_fmac_f32 %0, %4, %0\n\tv_fmac_f32 %1, %4, %1\n\tv_fmac_f32 %2, %4, %2\n\tv_fmac_f32 %3, %4, %3" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "s"(s.x));
            } else if (MODE == 2) {  // DPP moves
                asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_mov_b32_dpp %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                             "v_mov_b32_dpp %2, %3 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_mov_b32_dpp %3, %0 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
            } else if (MODE == 3) {  // plain moves
                asm volatile("v_mov_b32 %0, %1\n\tv_mov_b32 %1, %2\n\tv_mov_b32 %2, %3\n\tv_mov_b32 %3, %0" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
            } else if (MODE == 4) {  // packed FMA with a broadcast VGPR half (op_sel) x SGPR pair
                asm volatile("v_pk_fma_f32 %0, %1, %4, %0 op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 %1, %2, %4, %1 op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 %2, %3, %4, %2 op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 %3, %0, %4, %3 op_sel_hi:[0,1,1]" : "+v"(c), "+v"(d), "+v"(e), "+v"(g) : "s"(s));
            } else if (MODE == 5) {  // v_readlane
                int r;
                asm volatile("v_readlane_b32 %0, %1, 3\n\tv_readlane_b32 %0, %2, 4\n\tv_readlane_b32 %0, %3, 5\n\tv_readlane_b32 %0, %4, 6" : "=s"(r) : "v"(x0), "v"(x1), "v"(x2), "v"(x3));
            } else if (MODE == 6) {  // pk_fma all VGPR
                asm volatile("v_pk_fma_f32 %0, %0, %4, %0\n\tv_pk_fma_f32 %1, %1, %4, %1\n\tv_pk_fma_f32 %2, %2, %4, %2\n\tv_pk_fma_f32 %3, %3, %4, %3" : "+v"(c), "+v"(d), "+v"(e), "+v"(g) : "v"(b));
            }
        }
    }
    long long t1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
    out[blockIdx.x * 256 + threadIdx.x] = a.x + c.x + d.y + e.x + g.y + x0 + x1 + x2 + x3;
}

int main()
{
    float *sp, *out; long long* cyc;
    CK(hipMalloc(&sp, 8)); CK(hipMalloc(&out, 1 << 22)); CK(hipMalloc(&cyc, 8));
    float h[2] = {3.f, 7.f};
    CK(hipMemcpy(sp, h, 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_opsel, dim3(1), dim3(64), 0, 0, sp, out);
    float r[12];
    CK(hipMemcpy(r, out, sizeof(r), hipMemcpyDeviceToHost));
    printf("lane 0 (v = (0,100), s = (3,7)): plain (%g,%g)  low-half broadcast (%g,%g) [want (0,300)]  high-half broadcast (%g,%g) [want (0,700)]\n", r[0], r[1], r[2], r[3], r[4], r[5]);
    printf("lane 1 (v = (1,101)): plain (%g,%g)  low (%g,%g) [want (3,303)]  high (%g,%g) [want (7,707)]\n", r[6], r[7], r[8], r[9], r[10], r[11]);
    const char* names[] = {"v_pk_fma_f32 v,s", "v_fmac_f32 v,s", "v_mov_b32_dpp", "v_mov_b32", "v_pk_fma_f32 op_sel bcast", "v_readlane_b32", "v_pk_fma_f32 v,v"};
    const int iters = 2000;
    for (int mode = 0; mode < 7; mode++) {
        for (int wpb : {64, 256, 1024}) {  // 1 wave per CU (one SIMD busy), 1 per SIMD, 4 per SIMD
            long long c = 0;
            for (int rep = 0; rep < 2; rep++) {
                switch (mode) {
                    case 0: hipLaunchKernelGGL(k_rate<0>, dim3(256), dim3(wpb), 0, 0, out, iters, sp, cyc); break;
                    case 1: hipLaunchKernelGGL(k_rate<1>, dim3(256), dim3(wpb), 0, 0, out, iters, sp, cyc); break;
                    case 2: hipLaunchKernelGGL(k_rate<2>, dim3(256), dim3(wpb), 0, 0, out, iters, sp, cyc); break;
                    case 3: hipLaunchKernelGGL(k_rate<3>, dim3(256), dim3(wpb), 0, 0, out, iters, sp, cyc); break;
                    case 4: hipLaunchKernelGGL(k_rate<4>, dim3(256), dim3(wpb), 0, 0, out, iters, sp, cyc); break;
                    case 5: hipLaunchKernelGGL(k_rate<5>, dim3(256), dim3(wpb), 0, 0, out, iters, sp, cyc); break;
                    default: hipLaunchKernelGGL(k_rate<6>, dim3(256), dim3(wpb), 0, 0, out, iters, sp, cyc); break;
                }
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
            }
            printf("%-28s %4d threads/block: %.2f cycles per instruction per wave\n", names[mode], wpb, (double)c / (iters * 64.0));
        }
    }
    return 0;
}
