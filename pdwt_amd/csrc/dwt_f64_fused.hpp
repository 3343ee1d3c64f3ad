// dwt_f64_fused.hpp -- fused (row pass + column pass) level kernels for long double-precision banks (dwt_f64_fused.hip).
// Return PDWT_OK when the level was launched, 1 when the geometry / filter length is outside this path.
// taps_dev: >= 2*PDWT_MAX_FILTER_WIDTH doubles of device scratch the launch may overwrite (stream-ordered).
#pragma once
#include "common.hpp"

namespace pdwt {
int fwd2d_f64_fused(const double* in, double* cA, double* cH, double* cV, double* cD, double* taps_dev, int nr, int nc, int hlen,
                    const Taps2<double>& f);
int inv2d_f64_fused(const double* cA, const double* cH, const double* cV, const double* cD, double* out, double* taps_dev, int nri, int nci,
                    int nro, int nco, int hlen, const Taps2<double>& f);
// LDS-ring form (dwt_lds.hip): same contract
int fwd2d_f64_lds(const double* in, double* cA, double* cH, double* cV, double* cD, double* taps_dev, int nr, int nc, int hlen,
                  const Taps2<double>& f);
int inv2d_f64_lds(const double* cA, const double* cH, const double* cV, const double* cD, double* out, double* taps_dev, int nri, int nci,
                  int nro, int nco, int hlen, const Taps2<double>& f);
// float32 banks of more than 16 taps (shorter ones belong to the cascade / streaming kernels)
int fwd2d_f32_lds(const float* in, float* cA, float* cH, float* cV, float* cD, int nr, int nc, int hlen, const Taps2<float>& f);
int inv2d_f32_lds(const float* cA, const float* cH, const float* cV, const float* cD, float* out, int nri, int nci, int nro, int nco,
                  int hlen, const Taps2<float>& f);
// taps in the kernels' consumption order -> taps_dev (stream-ordered)
int f64_store_taps_fwd(const Taps2<double>& f, int hlen, double* taps_dev);
int f64_store_taps_inv(const Taps2<double>& f, int hlen, double* taps_dev);
}  // namespace pdwt
