// tapreg.hpp -- "tap register" helpers for long filter banks (double precision: 2*hlen taps do not fit a wave's SGPRs).
// Lane k of one VGPR holds tap k; a tap is broadcast with v_readlane right before the FMAs that use it, and an ordering
// point keeps the reads from being hoisted together (which would need every SGPR at once and spill them straight back
// into VGPR lanes: 2.5 v_readlane per FMA measured with the taps passed by value).  See cols_ring.inc / rows_tr.hip.
#pragma once
#include "common.hpp"

namespace pdwt {

constexpr int kTapG = 4;  // rows per tap-major group

__device__ __forceinline__ float lane_bcast(float v, int k) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), k)); }
__device__ __forceinline__ double lane_bcast(double v, int k)
{
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, k);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), k);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
template <typename T> __device__ __forceinline__ void opaque(T& v) { asm volatile("" : "+v"(v)); }
// zero-instruction ordering point: the tap registers pass through it together with one accumulator of the previous
// tap, so the next tap's v_readlanes cannot be scheduled above the previous tap's FMAs (hoisted together, the reads
// need every SGPR at once and hipcc spills them straight back into VGPR lanes)
template <typename T> __device__ __forceinline__ void tap_order(T& ta, T& tb, T (&a)[8])
{
    asm volatile("" : "+v"(ta), "+v"(tb), "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));
}

}  // namespace pdwt
