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
}  // namespace pdwt
