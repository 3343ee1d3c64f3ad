// dwt1d_fused.hpp -- all-levels-in-one-launch batched-1D DWT (dwt1d_fused.hip).
// Return PDWT_OK when launched, 1 when the signal does not fit the LDS budget / unsupported length
// (caller falls back to the per-level kernels), negative on error.
#pragma once
#include "common.hpp"

namespace pdwt {
template <typename T> int fwd1d_fused(const T* in, T** coeffs, const pdwt_info& w, const Taps2<T>& f);
template <typename T> int inv1d_fused(T* out, T** coeffs, const pdwt_info& w, const Taps2<T>& f);
}  // namespace pdwt
