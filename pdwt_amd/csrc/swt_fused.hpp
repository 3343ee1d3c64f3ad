// swt_fused.hpp -- one forward SWT level per launch (swt_fused.inc).  Returns PDWT_OK when launched, 1 when the
// geometry is outside this path (caller runs the row pass + column pass kernels), < 0 on a HIP error.
#pragma once
#include "common.hpp"

namespace pdwt {
// in (Nr x Nc) -> cA, cH, cV, cD (Nr x Nc), tap spacing fct = 2^(level-1).  `in` must not alias an output.
// d_tbl != NULL: a batch of nimg images in ONE launch (gridDim.z = image): device array of five pointers per image -- forward (in, cA, cH, cV,
// cD), inverse (cA, cH, cV, cD, out) -- every one 16-byte aligned, no input aliasing an output; the pointer arguments are then ignored.
int swt_fwd_fused_f32(const float* in, float* cA, float* cH, float* cV, float* cD, int Nr, int Nc, int hlen, int fct, const Taps2<float>& f,
                      const void* d_tbl = nullptr, int nimg = 1);
// bands (Nr x Nc) -> out (Nr x Nc); taps = the inverse bank already halved (taps_inv(filt, 0.5)).  `out` must not alias an input.
int swt_inv_fused_f32(const float* cA, const float* cH, const float* cV, const float* cD, float* out, int Nr, int Nc, int hlen, int fct,
                      const Taps2<float>& f, const void* d_tbl = nullptr, int nimg = 1);
// double precision (swt_fused_f64.inc): same contracts, Nc even
int swt_fwd_fused_f64(const double* in, double* cA, double* cH, double* cV, double* cD, int Nr, int Nc, int hlen, int fct, const Taps2<double>& f);
int swt_inv_fused_f64(const double* cA, const double* cH, const double* cV, const double* cD, double* out, int Nr, int Nc, int hlen, int fct,
                      const Taps2<double>& f);
}  // namespace pdwt
