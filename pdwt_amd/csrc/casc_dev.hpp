// casc_dev.hpp -- geometry and argument structs shared by the multi-level-per-launch kernels
// (dwt_casc.hip: forward + independent-wave inverse; dwt_casc_invw.hip: workgroup-form inverse, two or three levels).
#pragma once
#include "common.hpp"

// diagnostic builds only (tools/casc_diag.sh): PDWT_CASC_DIAG & 1 folds every stored row, & 2 every loaded row onto 32 rows of its
// array (L2-resident): the results are wrong, the timings say what the cascade kernels cost without their memory traffic
#ifndef PDWT_CASC_DIAG
#define PDWT_CASC_DIAG 0
#endif
#define CASC_DIAG_ST(row) ((PDWT_CASC_DIAG & 1) ? ((row) & 31) : (row))
#define CASC_DIAG_LD(row) ((PDWT_CASC_DIAG & 2) ? ((row) & 31) : (row))

namespace pdwt {

template <int HLEN>
struct CascGeom {
    static constexpr int C = HLEN / 2 - 1;                      // halo samples per side, both levels
    static constexpr int NB1 = C > 0 ? (C + 3) / 4 : 0;         // halo lanes for the input window (4 columns per lane)
    static constexpr int NB2 = C > 0 ? (C + 1) / 2 : 0;         // halo lanes for the A1 window (2 columns per lane)
    static constexpr int NBT = NB1 + NB2;                       // lanes per side that produce no output
    static constexpr int WIN1 = 4 * (2 * NB1 + 1);
    static constexpr int WIN2 = 2 * (2 * NB2 + 1);
    static constexpr int MAXVL = 64 - 2 * NBT;
};

struct CascMap {
    int cpx;     // W == 1: chunk rows per XCD (all 8 XCDs get the same number); W > 1: workgroups per XCD
    int strips;  // strips per chunk row
    int gy;      // W > 1: workgroup-chunk rows (gy * strips workgroups in all)
    int flags;   // XCD weighting of the chunk split, see casc_chunk_start (0 = even split)
    const void* tbl;  // batched launch (gridDim.y = images): device array of CascBatchF / CascBatchI, one entry per image; NULL = one image
};
// Chunk boundaries of the workgroup-chunk rows of ONE strip, weighted by XCD.  In-kernel timelines (tools/casc_trace.py, three boxes) show
// the workgroups of the odd XCDs ending 0.7-1.7 us after those of the even ones, in both directions: the same rows take them ~5 % longer.
// Every strip is split over its cm.gy workgroups independently of the other strips, so a workgroup on an even XCD simply takes `d`
// units (level-2 rows forward, level-(l+1) row pairs inverse) more than the even split and one on an odd XCD `d` fewer: boundary g of
// strip s moves by d * (#even - #odd workgroups above it).  cm.flags = d (knob casc_xcdw), 0 = the even split.  Same function on the host
// (the launchers replay the split to decide whether every wave has a straight-line program).
__host__ __device__ inline int casc_chunk_start(int g, int strip, int units, int gy, int strips, int cpx, int d)
{
    if (g >= gy) return units;
    int b = (int)(((long long)g * units) / gy);
    if (d != 0) {
        int bal = 0;
        for (int q = 0; q < g; q++) bal += (((q * strips + strip) / cpx) & 1) ? -1 : 1;
        b += d * bal;
    }
    return b;
}

struct CascBands {
    float *H1, *V1, *D1, *A2, *H2, *V2, *D2;
};
// per-image pointers of a batched forward cascade launch (pdwt_batch2d_*, dwt.hip)
struct CascBatchF {
    const float* in;
    CascBands b;
};

template <int HLEN>
struct CascInvGeom {
    static constexpr int H2 = HLEN / 2;
    static constexpr int C = H2 / 2;
    static constexpr int SHIFT = (H2 & 1) ? 0 : 1;
    static constexpr int NB1 = C > 0 ? (C + 1) / 2 : 0;  // halo lanes, level l (2 coefficient columns per lane)
    static constexpr int NB2 = C;                         // halo lanes, level l+1 (1 column per lane)
    static constexpr int NBT = NB1 + NB2;
    static constexpr int WIN1 = 2 * (2 * NB1 + 1);
    static constexpr int WIN2 = 2 * NB2 + 1;
    static constexpr int MAXVL = 64 - 2 * NBT;
    // chunks start at level-l coefficient rows of this parity so that the first A_l row a chunk needs
    // (ya - C) is the first of the pair a level-(l+1) step produces (rows 2P-SHIFT, 2P+1-SHIFT)
    static constexpr int BASE = (C - SHIFT) & 1;
    static constexpr int VM_SB = H2 * (4 + 2 * (3 + 2));  // VMEM instructions per super-body
};
struct CascInvBands {
    const float *A2, *H2, *V2, *D2, *H1, *V1, *D1;
};
struct CascInv3B {
    const float *A3, *H3, *V3, *D3;
};
// per-image pointers of a batched inverse cascade launch
struct CascBatchI {
    CascInvBands b;
    CascInv3B b3;
    float* out;
};


// ---- in-kernel timeline (diagnostic builds only: tools/build_trace.sh compiles the two cascade files with -DPDWT_CASC_TRACE) ----
// Every wave keeps a few readings of the 100 MHz real-time counter in SGPRs and lane 0 stores them AFTER the final drain (the
// hand-counted vmcnt pipeline never sees an extra store), at float offset kCascTraceOff of the trash area: 8 x u64 per wave,
// wave id = blockIdx.x * W + wave (the inverse kernels one more kCascTraceOff further on, so that a forward / inverse pair can run
// back to back).  tools/casc_trace.py reads them back and prints the launch ramp, the prologue, the per-phase
// durations and the tail.
constexpr size_t kCascTraceOff = 1u << 20;
#ifdef PDWT_CASC_TRACE
#define CASC_TRACE_DECL unsigned long long casc_tr_[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define CASC_TRACE(i) casc_tr_[i] = wall_clock64()
#define CASC_TRACE_STORE(trash, wave_id, aux)                                                             \
    do {                                                                                                   \
        if ((threadIdx.x & 63) == 0) {                                                                     \
            unsigned long long* p_ = reinterpret_cast<unsigned long long*>((trash) + kCascTraceOff) + (size_t)(wave_id) * 8; \
            for (int i_ = 0; i_ < 7; i_++) p_[i_] = casc_tr_[i_];                                          \
            p_[7] = (unsigned long long)(aux);                                                             \
        }                                                                                                  \
    } while (0)
#else
#define CASC_TRACE_DECL
#define CASC_TRACE(i)
#define CASC_TRACE_STORE(trash, wave_id, aux)
#endif

// dwt_casc_inv3.hip: three levels, all streamed (A3 != NULL), or two; hlen 4 and 8; 1 = not taken
int inv2d_casc3_f32(const float* A2, const float* H2, const float* V2, const float* D2, const float* H1, const float* V1, const float* D1,
                    const float* A3, const float* H3, const float* V3, const float* D3, float* out, float* trash, int nr, int nc, int hlen,
                    const Taps2<float>& f, const CascBatchI* d_tbl = nullptr, int nimg = 1);

int inv2d_cascw_f32(const float* A2, const float* H2, const float* V2, const float* D2, const float* H1, const float* V1, const float* D1,
                    const float* A3, const float* H3, const float* V3, const float* D3, float* out, float* trash, int nr, int nc, int hlen,
                    const Taps2<float>& f);

}  // namespace pdwt
