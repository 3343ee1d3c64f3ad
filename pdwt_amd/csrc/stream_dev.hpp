// stream_dev.hpp -- device-side building blocks shared by the float32 streaming kernels
// (dwt_stream.hip: one level per launch; dwt_casc.hip: two levels per launch).
#pragma once
#include <stdlib.h>

#include <type_traits>

#include "common.hpp"

// The hand-counted memory pipelines below (and the EXEC = 0 stores counted in them) are validated on gfx950 only -- tools/probes/vmcnt_order.hip,
// pdwt_selfcheck_vmcnt_order() -- and the kernels are scheduled for CDNA4's issue rules: any other target is a build error, not a silent port.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "pdwt_amd streaming kernels: gfx950 (MI355X) only -- see stream_dev.hpp"
#endif

namespace pdwt {

// bound_ctrl:1 + no `old` operand: the lane without a source reads 0 and the compiler needs no
// initialising v_mov per shift (that lane is a halo lane and produces no output anyway)
__device__ __forceinline__ float dpp_shr1(float src)
{  // lane i <- lane i-1
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(src), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_shl1(float src)
{  // lane i <- lane i+1
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(src), 0x130, 0xf, 0xf, true));
}

__device__ __forceinline__ int wrapi(int s, int n)
{
    s %= n;
    return s < 0 ? s + n : s;
}
// single conditional wrap: valid for -n <= s < 2n (row indices of a chunk; the dispatcher guarantees n >= 2*hlen)
__device__ __forceinline__ int wrap1(int s, int n) { return s < 0 ? s + n : (s >= n ? s - n : s); }

// compile-time loop: ring slots must be constants for the rings to stay in registers
template <int I, int N, typename F>
__device__ __forceinline__ void static_for_impl(F&& fn)
{
    if constexpr (I < N) {
        fn(std::integral_constant<int, I>{});
        static_for_impl<I + 1, N>(fn);
    }
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& fn)
{
    static_for_impl<0, N>(fn);
}

// Packed-f32 arithmetic.  gfx950 issues v_pk_fma_f32 (2 FMAs per lane) in the slot of one VALU op and can
// broadcast either half of a 64-bit operand (op_sel), so the kernels are written on explicit float pairs:
//   forward : (lo,hi) += x * (L[k],H[k])   -- one input sample feeds both filters; taps travel as pairs
//   inverse : (c0,c1) += (band[c0],band[c1]) * tap   -- the two coefficient columns a lane owns
// Each scalar still accumulates its taps in the reference order with one FMA per tap (bit-exact vs the oracle).
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f splat(float x) { return v2f{x, x}; }

// acc += v * splat(one half of an SGPR pair).  v_pk_fma_f32 can feed either half of a 64-bit source to both lanes (op_sel /
// op_sel_hi), also when the source is an SGPR pair (tools/probes/pkfma.hip) -- hipcc does not use that for scalar operands: it
// materialises a pre-splatted SGPR pair (t, t) per tap, 2 SGPRs per tap on top of the natural pairs the row passes use, and the
// inverse cascade kernels then spill ~50 SGPRs to VGPR lanes (a v_readlane per use).  With these the taps live ONCE, as the
// aligned pairs (F[2i], F[2i+1]).  FIRST: the accumulator starts at 0 (same result as pk_fma(v, splat, {0, 0})).
template <int HALF, bool FIRST>
__device__ __forceinline__ v2f pk_fma_sbcast(v2f v, v2f spair, v2f acc)
{
    if constexpr (FIRST) {
        v2f r;
        if constexpr (HALF == 0) asm("v_pk_fma_f32 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(v), "s"(spair));
        else asm("v_pk_fma_f32 %0, %1, %2, 0 op_sel:[0,1,0] op_sel_hi:[1,1,0]" : "=v"(r) : "v"(v), "s"(spair));
        return r;
    } else {
        if constexpr (HALF == 0) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(v), "s"(spair));
        else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(v), "s"(spair));
        return acc;
    }
}

// acc += splat(one half of a VGPR pair) * (an SGPR pair as it is): the forward column passes, (A, H) += lo * (L[k], H[k]) and
// (V, D) += hi * (L[k], H[k]) with (lo, hi) a ring entry.  Written out because in straight-line code (the specialised wave programs of
// dwt_casc.hip) hipcc CSEs the splat of a ring value over the eight column passes that use it and keeps it as a materialised register
// PAIR (two v_mov per value, twice the ring's registers -> spills) instead of using the broadcast of the instruction.
template <int HALF, bool FIRST>
__device__ __forceinline__ v2f pk_fma_vbcast(v2f v, v2f spair, v2f acc)
{
    if constexpr (FIRST) {
        v2f r;
        if constexpr (HALF == 0) asm("v_pk_fma_f32 %0, %1, %2, 0 op_sel_hi:[0,1,0]" : "=v"(r) : "v"(v), "s"(spair));
        else asm("v_pk_fma_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,1,0]" : "=v"(r) : "v"(v), "s"(spair));
        return r;
    } else {
        if constexpr (HALF == 0) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(v), "s"(spair));
        else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0]" : "+v"(acc) : "v"(v), "s"(spair));
        return acc;
    }
}

// forward taps as (L[k], H[k]) pairs, by value in the kernarg segment (-> SGPR pairs)
struct TapsLH {
    v2f t[PDWT_MAX_FILTER_WIDTH];
};

// Block -> (chunk row, strip group) map.  Workgroups are dispatched round-robin over the 8 XCDs
// (block b -> XCD b % 8), each with a private L2.  Vertically adjacent chunks share hlen-2 halo rows, so
// every XCD gets a contiguous band of chunk rows and walks it top to bottom: the halo rows are then L2
// hits instead of a second trip to Infinity Cache / HBM.  Pure speed: any other placement is still correct.
struct ChunkMap {
    int gx;       // workgroups per chunk row
    int nchunks;  // chunk rows
    int rpx;      // chunk rows per XCD band = ceil(nchunks / 8)
};
__device__ __forceinline__ bool chunk_of_block(const ChunkMap& m, int& cy, int& bx)
{
    const int b = blockIdx.x;
    const int xcd = b & 7, slot = b >> 3;
    cy = xcd * m.rpx + slot / m.gx;
    bx = slot % m.gx;
    return cy < min(m.nchunks, (xcd + 1) * m.rpx);
}

typedef float v4f __attribute__((ext_vector_type(4)));
template <int N> struct VecOf;
template <> struct VecOf<4> { using type = v4f; };
template <> struct VecOf<2> { using type = v2f; };
__device__ __forceinline__ void vstore(float* p, const float (&a)[2]) { *reinterpret_cast<float2*>(p) = make_float2(a[0], a[1]); }
__device__ __forceinline__ void vstore(float* p, const float (&a)[1]) { *p = a[0]; }

// ---- hand-counted memory pipeline (steady-state loop only) -----------------------------------------
// hipcc sizes every s_waitcnt for the worst incoming edge and, on gfx9, counts stores in vmcnt too, so a
// compiler-scheduled loop ends every body with vmcnt(0): the wave drains its stores and its prefetched loads
// before it may compute again.  In the branch-free steady-state bodies all global loads and stores are
// therefore inline asm (invisible to hipcc's counting) and each consumer waits with an exact
// `s_waitcnt vmcnt(N)`, N = number of VMEM instructions the wave issues between that load and its use.
// Memory instructions retire in issue order, so "at most N outstanding" means the load has landed while the
// N younger loads/stores stay in flight.  Rules that keep the count exact: every lane-predicated store goes
// to a trash slot instead of being branched around; the loop is entered and left through vmcnt(0).
// The destination is a TIED operand ("+v"): the in-flight value keeps the physical register of the value it
// replaces, so the compiler never has to copy a register whose load has not landed yet (it believes the asm
// defines it at once; a v_mov of such a register -- e.g. a phi copy at a loop back-edge -- would read stale data).
__device__ __forceinline__ void asm_load(v4f& d, const float* p) { asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(d) : "v"(p) : "memory"); }
__device__ __forceinline__ void asm_load(v2f& d, const float* p) { asm volatile("global_load_dwordx2 %0, %1, off" : "+v"(d) : "v"(p) : "memory"); }
__device__ __forceinline__ void asm_load(float& d, const float* p) { asm volatile("global_load_dword %0, %1, off" : "+v"(d) : "v"(p) : "memory"); }
__device__ __forceinline__ void asm_store(float* p, v2f d) { asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(d) : "memory"); }
__device__ __forceinline__ void asm_store(float* p, float d) { asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"(d) : "memory"); }
__device__ __forceinline__ void asm_store(float* p, v4f d) { asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(d) : "memory"); }
// SGPR-base forms: address = uniform 64-bit base (SGPR pair, computed on the scalar unit) + per-lane 32-bit byte
// offset (a loop-invariant VGPR) -> no vector ALU work per access.  The masked stores run with EXEC = `mask`
// (lanes that own no output are switched off instead of being redirected); a partially masked VMEM instruction
// still counts as one in vmcnt, so the hand-counted waits stay exact.
typedef unsigned long long lanemask_t;
// (experiments only, tools/variant_lib.sh: -DPDWT_LD_NT=1 gives every streaming load of a translation unit the non-temporal hint)
#if defined(PDWT_LD_NT) && PDWT_LD_NT
#define PDWT_LD_POLICY " nt"
#else
#define PDWT_LD_POLICY ""
#endif
__device__ __forceinline__ void asm_load_s(v4f& d, const float* sbase, unsigned voff)
{
    asm volatile("global_load_dwordx4 %0, %1, %2" PDWT_LD_POLICY : "+v"(d) : "v"(voff), "s"(sbase) : "memory");
}
__device__ __forceinline__ void asm_load_s(v2f& d, const float* sbase, unsigned voff)
{
    asm volatile("global_load_dwordx2 %0, %1, %2" PDWT_LD_POLICY : "+v"(d) : "v"(voff), "s"(sbase) : "memory");
}
__device__ __forceinline__ void asm_load_s(float& d, const float* sbase, unsigned voff)
{
    asm volatile("global_load_dword %0, %1, %2" PDWT_LD_POLICY : "+v"(d) : "v"(voff), "s"(sbase) : "memory");
}
__device__ __forceinline__ void asm_store_sm(float* sbase, unsigned voff, v2f d, lanemask_t mask)
{
    lanemask_t saved;
    asm volatile("s_and_saveexec_b64 %0, %4\n\tglobal_store_dwordx2 %1, %2, %3\n\ts_mov_b64 exec, %0"
                 : "=&s"(saved) : "v"(voff), "v"(d), "s"(sbase), "s"(mask) : "memory", "scc");
}
__device__ __forceinline__ void asm_store_sm(float* sbase, unsigned voff, float d, lanemask_t mask)
{
    lanemask_t saved;
    asm volatile("s_and_saveexec_b64 %0, %4\n\tglobal_store_dword %1, %2, %3\n\ts_mov_b64 exec, %0"
                 : "=&s"(saved) : "v"(voff), "v"(d), "s"(sbase), "s"(mask) : "memory", "scc");
}
__device__ __forceinline__ void asm_store_sm(float* sbase, unsigned voff, v4f d, lanemask_t mask)
{
    lanemask_t saved;
    asm volatile("s_and_saveexec_b64 %0, %4\n\tglobal_store_dwordx4 %1, %2, %3\n\ts_nop 1\n\ts_mov_b64 exec, %0"
                 : "=&s"(saved) : "v"(voff), "v"(d), "s"(sbase), "s"(mask) : "memory", "scc");
}
// Several masked stores under ONE exec save / restore (2 scalar instructions per group instead of 2 per store; the scalar unit is what
// bounds the cascade kernels).  mask == 0 is allowed: the stores then write nothing but still count in vmcnt, in order
// (tools/probes/vmcnt_order.hip), so rows a wave does not own need neither a trash row nor a pointer select.
__device__ __forceinline__ void asm_store3_sm(float* b0, float* b1, float* b2, unsigned voff, v2f d0, v2f d1, v2f d2, lanemask_t mask)
{
    lanemask_t saved;
    asm volatile("s_and_saveexec_b64 %0, %8\n\tglobal_store_dwordx2 %1, %2, %5\n\tglobal_store_dwordx2 %1, %3, %6\n\tglobal_store_dwordx2 %1, %4, %7\n\ts_mov_b64 exec, %0"
                 : "=&s"(saved) : "v"(voff), "v"(d0), "v"(d1), "v"(d2), "s"(b0), "s"(b1), "s"(b2), "s"(mask) : "memory", "scc");
}
__device__ __forceinline__ void asm_store4_sm(float* b0, float* b1, float* b2, float* b3, unsigned voff, float d0, float d1, float d2, float d3, lanemask_t mask)
{
    lanemask_t saved;
    asm volatile("s_and_saveexec_b64 %0, %10\n\tglobal_store_dword %1, %2, %6\n\tglobal_store_dword %1, %3, %7\n\tglobal_store_dword %1, %4, %8\n\tglobal_store_dword %1, %5, %9\n\ts_mov_b64 exec, %0"
                 : "=&s"(saved) : "v"(voff), "v"(d0), "v"(d1), "v"(d2), "v"(d3), "s"(b0), "s"(b1), "s"(b2), "s"(b3), "s"(mask) : "memory", "scc");
}
// the same with the non-temporal hint (streaming data nobody reads soon: the detail bands of an SWT level)
__device__ __forceinline__ void asm_store_sm_nt(float* sbase, unsigned voff, v4f d, lanemask_t mask)
{
    lanemask_t saved;
    asm volatile("s_and_saveexec_b64 %0, %4\n\tglobal_store_dwordx4 %1, %2, %3 nt\n\ts_nop 1\n\ts_mov_b64 exec, %0"
                 : "=&s"(saved) : "v"(voff), "v"(d), "s"(sbase), "s"(mask) : "memory", "scc");
}
// Opaque register copies out of a load register.  A plain C++ copy may be coalesced with its source; the tied
// load that follows would then be given a fresh register and the loop back-edge a v_mov of the in-flight one.
__device__ __forceinline__ float asm_copy(const float& src)
{
    float d;
    asm volatile("v_mov_b32 %0, %1" : "=v"(d) : "v"(src));
    return d;
}
__device__ __forceinline__ v2f asm_copy(const v2f& src)
{
    v2f d;
    asm volatile("v_mov_b64 %0, %1" : "=v"(d) : "v"(src));
    return d;
}
template <int N, typename V>
__device__ __forceinline__ void asm_wait2(V& a, V& b) { asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory"); }
template <int N, typename V>
__device__ __forceinline__ void asm_wait4(V& a, V& b, V& c, V& d) { asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N) : "memory"); }
template <int N, typename V>
__device__ __forceinline__ void asm_wait3(V& a, V& b, V& c) { asm volatile("s_waitcnt vmcnt(%3)" : "+v"(a), "+v"(b), "+v"(c) : "n"(N) : "memory"); }
// Counted wait chosen at run time inside ONE asm statement: `counted` (uniform) ? vmcnt(N) : vmcnt(0).  An if / else around two tied
// waits makes the load registers PHI values, and hipcc may then copy a register whose load is still in flight.
template <int N, typename V>
__device__ __forceinline__ void asm_wait2_sel(bool counted, V& a, V& b)
{
    asm volatile("s_cmp_eq_u32 %2, 0\n\ts_cbranch_scc1 .Lws0_%=\n\ts_waitcnt vmcnt(%3)\n\ts_branch .Lws1_%=\n.Lws0_%=:\n\ts_waitcnt vmcnt(0)\n.Lws1_%=:"
                 : "+v"(a), "+v"(b) : "s"(__builtin_amdgcn_readfirstlane((int)counted)), "n"(N) : "memory", "scc");
}
template <int N, typename V>
__device__ __forceinline__ void asm_wait8_sel(bool counted, V& a, V& b, V& c, V& d, V& e, V& f, V& g, V& h)
{
    asm volatile("s_cmp_eq_u32 %8, 0\n\ts_cbranch_scc1 .Lws0_%=\n\ts_waitcnt vmcnt(%9)\n\ts_branch .Lws1_%=\n.Lws0_%=:\n\ts_waitcnt vmcnt(0)\n.Lws1_%=:"
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "s"(__builtin_amdgcn_readfirstlane((int)counted)), "n"(N) : "memory", "scc");
}
template <typename V>
__device__ __forceinline__ void asm_drain1(V& a) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(a) : : "memory"); }


// ---- host helpers ----------------------------------------------------------------------------------
static inline ChunkMap make_map(int gx, int nchunks, dim3* grid)
{
    ChunkMap m;
    m.gx = gx;
    m.nchunks = nchunks;
    m.rpx = idiv_up(nchunks, 8);
    *grid = dim3((unsigned)(8 * m.rpx * gx));
    return m;
}
static inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace pdwt
