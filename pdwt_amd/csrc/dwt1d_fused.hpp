// dwt1d_fused.hpp -- all-levels-in-one-launch batched-1D DWT (dwt1d_fused.hip).
// Return PDWT_OK when launched, 1 when the signal does not fit the LDS budget / unsupported length
// (caller falls back to the per-level kernels), negative on error.
#pragma once
#include "common.hpp"

namespace pdwt {
namespace dflt {
template <typename T> int fwd1d_fused(const T* in, T** coeffs, const pdwt_info& w, const Taps2<T>& f);
template <typename T> int inv1d_fused(T* out, T** coeffs, const pdwt_info& w, const Taps2<T>& f);
}  // namespace dflt
namespace nt {  // float32 only: non-temporal row / band loads (dwt1d_fused_nt.hip)
template <typename T> int fwd1d_fused(const T* in, T** coeffs, const pdwt_info& w, const Taps2<T>& f);
template <typename T> int inv1d_fused(T* out, T** coeffs, const pdwt_info& w, const Taps2<T>& f);
}  // namespace nt
// batches that do not fit the Infinity Cache read their rows once: no line left behind (knob dwt1d_nt_mb: smallest image, in MB, that takes the nt variant; 0 = never)
template <typename T> inline bool dwt1d_use_nt(const pdwt_info& w)
{
    const int mb = knob(KN_DWT1D_NT_MB);
    return sizeof(T) == 4 && mb > 0 && (size_t)w.Nr * (size_t)w.Nc * sizeof(T) >= (size_t)mb << 20;
}
template <typename T> inline int fwd1d_fused(const T* in, T** coeffs, const pdwt_info& w, const Taps2<T>& f)
{
    if constexpr (sizeof(T) == 4) if (dwt1d_use_nt<T>(w)) return nt::fwd1d_fused<T>(in, coeffs, w, f);
    return dflt::fwd1d_fused<T>(in, coeffs, w, f);
}
template <typename T> inline int inv1d_fused(T* out, T** coeffs, const pdwt_info& w, const Taps2<T>& f)
{
    if constexpr (sizeof(T) == 4) if (dwt1d_use_nt<T>(w)) return nt::inv1d_fused<T>(out, coeffs, w, f);
    return dflt::inv1d_fused<T>(out, coeffs, w, f);
}
}  // namespace pdwt
