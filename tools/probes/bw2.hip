// probe: which ingredient of the streaming forward kernel costs time? (not product code)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
struct Map { int gx, nchunks, rpx; };
template <int HALO, bool XCD, int NFMA, int VLANES>
__global__ __launch_bounds__(256) void k(const float4* __restrict__ in, float2* __restrict__ a, float2* __restrict__ h, float2* __restrict__ v, float2* __restrict__ d, int Nr, int Nc4, int R, Map m, float s)
{
    int cy, bx;
    if (XCD) { int b = blockIdx.x; int xcd = b & 7, slot = b >> 3; cy = xcd * m.rpx + slot / m.gx; bx = slot % m.gx; if (cy >= min(m.nchunks, (xcd + 1) * m.rpx)) return; }
    else { cy = blockIdx.x / m.gx; bx = blockIdx.x % m.gx; if (cy >= m.nchunks) return; }
    int lane = threadIdx.x & 63, strip = bx * 4 + (threadIdx.x >> 6);
    int x = strip * VLANES + lane - (64 - VLANES) / 2; bool valid = lane >= (64 - VLANES) / 2 && lane < 64 - (64 - VLANES) / 2 && x < Nc4;
    if (strip * VLANES >= Nc4) return;
    x = (x + Nc4) % Nc4;
    int y0 = cy * R;
    float4 acc = make_float4(0, 0, 0, 0);
    // halo rows
    for (int r = 0; r < HALO; r++) { int yy = 2 * y0 - 3 + r; yy = yy < 0 ? yy + Nr : yy; float4 t = in[(size_t)yy * Nc4 + x]; acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w; }
    for (int q = 0; q < R; q += 4) {
        float4 r[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { int yy = 2 * (y0 + q) + HALO / 2 + u; yy = yy >= Nr ? yy - Nr : yy; r[u] = in[(size_t)yy * Nc4 + x]; }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            float4 s0 = r[2*u], s1 = r[2*u+1];
#pragma unroll
            for (int k = 0; k < NFMA; k++) { s0.x = __builtin_fmaf(s0.x, s, s1.x); s0.y = __builtin_fmaf(s0.y, s, s1.y); s0.z = __builtin_fmaf(s0.z, s, s1.z); s0.w = __builtin_fmaf(s0.w, s, s1.w);
                                            s1.x = __builtin_fmaf(s1.x, s, s0.x); s1.y = __builtin_fmaf(s1.y, s, s0.y); s1.z = __builtin_fmaf(s1.z, s, s0.z); s1.w = __builtin_fmaf(s1.w, s, s0.w); }
            size_t o = (size_t)(y0 + q + u) * Nc4 + x;
            if (valid) { a[o] = make_float2(s0.x + acc.x, s0.y); h[o] = make_float2(s0.z, s0.w + acc.y); v[o] = make_float2(s1.x, s1.y + acc.z); d[o] = make_float2(s1.z + acc.w, s1.w); }
        }
    }
}
template <typename F> float timeit(F f, int reps = 40) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) f();
    hipEventRecord(e0); for (int i = 0; i < reps; i++) f(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / reps;
}
template <int HALO, bool XCD, int NFMA, int VLANES>
void run(const char* tag, float* in, float** b, int N, int R) {
    int Nc4 = N / 4; int strips = (Nc4 + VLANES - 1) / VLANES; Map m; m.gx = (strips + 3) / 4; m.nchunks = N / 2 / R; m.rpx = (m.nchunks + 7) / 8;
    int grid = XCD ? 8 * m.rpx * m.gx : m.nchunks * m.gx;
    float us = timeit([&] { k<HALO, XCD, NFMA, VLANES><<<grid, 256>>>((float4*)in, (float2*)b[0], (float2*)b[1], (float2*)b[2], (float2*)b[3], N, Nc4, R, m, 0.5f); });
    printf("%-34s N=%d R=%2d grid %5d: %6.1f us  %.2f TB/s(alg)\n", tag, N, R, grid, us, 2.0 * N * N * 4 / us / 1e6);
}
int main() {
    const int N = 4096; size_t n = (size_t)N * N; float *in, *b[4];
    hipMalloc(&in, n * 4); for (int i = 0; i < 4; i++) hipMalloc(&b[i], n); hipMemset(in, 0, n * 4);
    for (int R : {8, 16, 32}) {
        run<0, false, 0, 64>("plain", in, b, N, R);
        run<0, false, 0, 62>("62 lanes", in, b, N, R);
        run<6, false, 0, 62>("62 lanes + halo", in, b, N, R);
        run<6, true, 0, 62>("62 lanes + halo + xcd", in, b, N, R);
        run<6, true, 8, 62>("62l+halo+xcd+ 64 fma/row", in, b, N, R);
        run<6, true, 16, 62>("62l+halo+xcd+128 fma/row", in, b, N, R);
        run<6, true, 32, 62>("62l+halo+xcd+256 fma/row", in, b, N, R);
        run<0, true, 16, 64>("64l+xcd+128 fma/row", in, b, N, R);
    }
    for (int M : {2048, 1024}) for (int R : {4, 8}) { run<6, true, 16, 62>("small 62l+halo+xcd+128fma", in, b, M, R); run<6, true, 0, 62>("small 62l+halo+xcd", in, b, M, R); }
    return 0;
}
