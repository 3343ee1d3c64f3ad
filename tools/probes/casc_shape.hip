// Access-pattern probe for the two-level forward cascade (not product code): same loads / stores per wave as k_fwd2d_casc,
// no filter arithmetic.  Variants isolate what the pattern costs: strip alignment, the 4-byte level-2 stores, halo rows,
// wave placement.  build: hipcc --offload-arch=gfx950 -O3 -o casc_shape casc_shape.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

struct P {
    int N;        // image is N x N
    int VL;       // producing lanes per strip
    int NBT;      // halo lanes per side (0 = aligned full-width strips)
    int rows;     // own input rows per chunk
    int halo;     // extra input rows read per chunk
    int strips, chunks;
    int l2mode;   // 0: four 4-byte level-2 stores per 4 rows, 1: none, 2: one 16-byte store by every 4th lane... (emulated as 1 store of 16 B masked)
    int vert;     // 1: the 4 waves of a block are stacked vertically in one strip; 0: four adjacent strips
    int l1mode;   // 0: three 8-byte stores per 2 rows; 1: none
};

template <int NVR>
__global__ __launch_bounds__(256) void k_shape(const float* __restrict__ in, float* __restrict__ H1, float* __restrict__ V1, float* __restrict__ D1,
                                               float* __restrict__ A2, float* __restrict__ H2, float* __restrict__ V2, float* __restrict__ D2, P p)
{
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int nblk = (p.strips * p.chunks + 3) / 4;
    const int per = (nblk + 7) / 8;
    const int blk = xcd * per + slot;
    if (slot >= per || blk >= nblk) return;
    int strip, chunk;
    if (p.vert) {  // block = 4 vertically adjacent chunks of one strip
        const int cg = p.chunks / 4;
        strip = blk % p.strips;
        chunk = (blk / p.strips) * 4 + w;
        if (blk / p.strips >= cg) return;
    } else {
        const int wi = blk * 4 + w;
        if (wi >= p.strips * p.chunks) return;
        chunk = wi / p.strips;
        strip = wi % p.strips;
    }
    const int N = p.N, N2 = N / 2, N4 = N / 4;
    const int x = strip * p.VL * 4 + 4 * (lane - p.NBT);
    const bool valid = lane >= p.NBT && lane < p.NBT + p.VL && x < N && x >= 0;
    int xo = x % N; if (xo < 0) xo += N;
    const int y0 = chunk * p.rows;
    const float* src = in + xo;
    const int nrows = p.rows + p.halo;
    float acc = 0.f;
    for (int r0 = 0; r0 < nrows; r0 += NVR) {
        float4 v[NVR];
#pragma unroll
        for (int u = 0; u < NVR; u++) {
            int y = y0 + r0 + u; if (y >= N) y -= N;
            v[u] = *reinterpret_cast<const float4*>(src + (size_t)y * N);
        }
#pragma unroll
        for (int u = 0; u < NVR; u += 2) {
            const int r = r0 + u;
            const bool own = r < p.rows;
            const float4 a = v[u], c = v[u + 1];
            if (own && p.l1mode == 2) {
                // paired rows: 16-byte stores, even lanes take row r/2, odd lanes the next one (emulated every other pair)
                if ((u & 2) == 0) {
                    const size_t o = (size_t)(((y0 + r) >> 1) + (lane & 1)) * N2 + ((x >> 1) & ~3) ;
                    if (valid) {
                    *reinterpret_cast<float4*>(H1 + o) = make_float4(a.x + c.x, a.y + c.y, a.z, c.w);
                    *reinterpret_cast<float4*>(V1 + o) = make_float4(a.z + c.z, a.w + c.w, a.x, c.y);
                    *reinterpret_cast<float4*>(D1 + o) = make_float4(a.x - c.x, a.y - c.y, c.x, a.y);
                    }
                }
            } else if (own && valid && p.l1mode == 0) {
                const size_t o = (size_t)((y0 + r) >> 1) * N2 + (x >> 1);
                *reinterpret_cast<float2*>(H1 + o) = make_float2(a.x + c.x, a.y + c.y);
                *reinterpret_cast<float2*>(V1 + o) = make_float2(a.z + c.z, a.w + c.w);
                *reinterpret_cast<float2*>(D1 + o) = make_float2(a.x - c.x, a.y - c.y);
            }
            acc += a.z - c.z + a.w - c.w;
            if ((u & 2) && own && valid) {
                const size_t o = (size_t)((y0 + r) >> 2) * N4 + (x >> 2);
                if (p.l2mode == 0) {
                    A2[o] = acc; H2[o] = acc + 1.f; V2[o] = acc + 2.f; D2[o] = acc + 3.f;
                } else if (p.l2mode == 3) {
                    // transposed: one 16-byte store per band per FOUR level-2 rows (lane = row r&3, 4 columns)
                    if (((r0 + u) & 15) == 14 || NVR == 8 && (((r0 + u) & 15) == 6) && false) {
                        const size_t o4 = (size_t)((((y0 + r) >> 2) & ~3) + (lane & 3)) * N4 + ((x >> 2) & ~63) + (lane >> 2) * 4;
                        *reinterpret_cast<float4*>(A2 + o4) = make_float4(acc, acc, acc, acc);
                        *reinterpret_cast<float4*>(H2 + o4) = make_float4(acc, acc + 1.f, acc, acc);
                        *reinterpret_cast<float4*>(V2 + o4) = make_float4(acc, acc, acc + 2.f, acc);
                        *reinterpret_cast<float4*>(D2 + o4) = make_float4(acc, acc, acc, acc + 3.f);
                    }
                } else if (p.l2mode == 2) {
                    if ((lane & 3) == 0) *reinterpret_cast<float4*>(A2 + ((o >> 2) << 2) * 4) = make_float4(acc, acc, acc, acc);
                }
            }
        }
    }
    if (acc == 12345.678f) H1[0] = acc;
}

template <typename F> float timeit(F f, int reps = 40) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; i++) f();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; i++) f();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / reps;
}

int main(int argc, char** argv) {
    const int N = 4096; size_t n = (size_t)N * N;
    float *in, *b[7];
    CK(hipMalloc(&in, n * 4));
    for (int i = 0; i < 7; i++) CK(hipMalloc(&b[i], n * 4));
    hipMemset(in, 1, n * 4);
    // settle clocks
    struct V { const char* name; int VL, NBT, waves, halo, l2mode, vert, l1mode, nvr; };
    std::vector<V> vs = {
        {"aligned 1024 w                        ", 64, 0, 1024, 0, 0, 0, 0, 8},
        {"aligned 1024 w, 16 rows               ", 64, 0, 1024, 0, 0, 0, 0, 16},
        {"aligned 2048 w                        ", 64, 0, 2048, 0, 0, 0, 0, 8},
        {"aligned 2048 w, 16 rows               ", 64, 0, 2048, 0, 0, 0, 0, 16},
        {"aligned 4096 w                        ", 64, 0, 4096, 0, 0, 0, 0, 8},
        {"aligned 4096 w, 16 rows               ", 64, 0, 4096, 0, 0, 0, 0, 16},
        {"aligned 4096 w, 16 rows, L2 none      ", 64, 0, 4096, 0, 1, 0, 0, 16},
        {"aligned 4096 w, 16 rows, L2 transposed", 64, 0, 4096, 0, 3, 0, 0, 16},
        {"aligned 4096 w, 16 rows, L2 tr, L1 16B", 64, 0, 4096, 0, 3, 0, 2, 16},
        {"aligned 4096 w, 16 rows, L2 4B, L1 16B", 64, 0, 4096, 0, 0, 0, 2, 16},
        {"aligned 2048 w, 16 rows, L2 tr, L1 16B", 64, 0, 2048, 0, 3, 0, 2, 16},
        {"aligned 1024 w, 16 rows, L2 tr, L1 16B", 64, 0, 1024, 0, 3, 0, 2, 16},
        {"aligned 4096 w, 16 rows, no L1, L2 4B ", 64, 0, 4096, 0, 0, 0, 1, 16},
        {"real 57/3, 2304 w, 16 rows            ", 57, 3, 2016, 0, 0, 0, 0, 16},
        {"real 57/3, 2304 w, 16 rows, L2 tr L1 16B", 57, 3, 2016, 0, 3, 0, 2, 16},
        {"real 57/3, 4608 w, 16 rows            ", 57, 3, 4032, 0, 0, 0, 0, 16},
        {"real 57/3, 4608 w, 16 rows, L2 tr L1 16B", 57, 3, 4032, 0, 3, 0, 2, 16},
    };
    for (int rep = 0; rep < 2; rep++)
    for (auto& v : vs) {
        P p; p.N = N; p.VL = v.VL; p.NBT = v.NBT; p.halo = v.halo; p.l2mode = v.l2mode; p.vert = v.vert; p.l1mode = v.l1mode;
        p.strips = (N / 4 + v.VL - 1) / v.VL;
        p.chunks = v.waves / p.strips; if (p.vert) p.chunks = (p.chunks / 4) * 4;
        p.rows = N / p.chunks; p.rows &= ~7;
        // rows must tile N exactly for a fair byte count: shrink chunks until they do
        while (N % p.chunks || (N / p.chunks) % 8) p.chunks--;
        p.rows = N / p.chunks;
        const int nblk = (p.strips * p.chunks + 3) / 4;
        dim3 g(8 * ((nblk + 7) / 8));
        float us;
        if (v.nvr == 8) us = timeit([&] { k_shape<8><<<g, 256>>>(in, b[0], b[1], b[2], b[3], b[4], b[5], b[6], p); });
        else us = timeit([&] { k_shape<16><<<g, 256>>>(in, b[0], b[1], b[2], b[3], b[4], b[5], b[6], p); });
        if (rep == 1) printf("%s strips %3d chunks %4d rows %3d waves %5d: %6.2f us  %.2f TB/s\n", v.name, p.strips, p.chunks, p.rows, p.strips * p.chunks, us, 2.0 * n * 4 / us / 1e6);
    }
    return 0;
}
