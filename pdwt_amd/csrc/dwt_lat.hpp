// dwt_lat.hpp -- level kernels for orthogonal double-precision banks with the column pass as a paraunitary lattice (dwt_lat.hip).
// Return PDWT_OK when the level was launched, 1 when the bank has no lattice / the geometry is outside this path (-> dwt_lds.hip).
#pragma once
#include "common.hpp"

namespace pdwt {
int fwd2d_f64_lat(const double* in, double* cA, double* cH, double* cV, double* cD, int nr, int nc, int hlen, const Taps2<double>& f);
int inv2d_f64_lat(const double* cA, const double* cH, const double* cV, const double* cD, double* out, int nri, int nci, int nro, int nco, int hlen,
                  const Taps2<double>& f);
void stat_lat(int inverse);  // (test statistics: a lattice level kernel was launched; runtime.hip)
}  // namespace pdwt
