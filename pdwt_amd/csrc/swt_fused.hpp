// swt_fused.hpp -- one forward SWT level per launch (swt_fused.hip).  Returns PDWT_OK when launched, 1 when the
// geometry is outside this path (caller runs the row pass + column pass kernels), < 0 on a HIP error.
#pragma once
#include "common.hpp"

namespace pdwt {
// in (Nr x Nc) -> cA, cH, cV, cD (Nr x Nc), tap spacing fct = 2^(level-1).  `in` must not alias an output.
int swt_fwd_fused_f32(const float* in, float* cA, float* cH, float* cV, float* cD, int Nr, int Nc, int hlen, int fct, const Taps2<float>& f);
}  // namespace pdwt
