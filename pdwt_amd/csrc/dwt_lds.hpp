// dwt_lds.hpp -- fused (row pass + column pass) level kernels with the ring in LDS / split register rings (dwt_lds.hip):
// double-precision banks of even length, float32 banks of more than 16 taps, odd sizes.
// Return PDWT_OK when the level was launched, 1 when the geometry / filter length is outside this path.
// taps_dev: unused (the tap table travels in the kernel argument segment); kept so that the level drivers need not change.
#pragma once
#include "common.hpp"

namespace pdwt {
int fwd2d_f64_lds(const double* in, double* cA, double* cH, double* cV, double* cD, double* taps_dev, int nr, int nc, int hlen,
                  const Taps2<double>& f);
int inv2d_f64_lds(const double* cA, const double* cH, const double* cV, const double* cD, double* out, double* taps_dev, int nri, int nci,
                  int nro, int nco, int hlen, const Taps2<double>& f);
// batched form (pdwt_batch2d_*_f64, dwt.hip): nimg images of one geometry in ONE launch (gridDim.y = image); d_tbl = device array of five
// pointers per image -- forward (in, cA, cH, cV, cD), inverse (cA, cH, cV, cD, out)
int fwd2d_f64_lds_batch(const void* d_tbl, int nimg, int nr, int nc, int hlen, const Taps2<double>& f);
int inv2d_f64_lds_batch(const void* d_tbl, int nimg, int nri, int nci, int nro, int nco, int hlen, const Taps2<double>& f);
// float32 banks of more than 16 taps (shorter ones belong to the cascade / streaming kernels)
int fwd2d_f32_lds(const float* in, float* cA, float* cH, float* cV, float* cD, int nr, int nc, int hlen, const Taps2<float>& f);
int inv2d_f32_lds(const float* cA, const float* cH, const float* cV, const float* cD, float* out, int nri, int nci, int nro, int nco,
                  int hlen, const Taps2<float>& f);
}  // namespace pdwt
