// rows_tr.hpp -- row passes for long double-precision banks (rows_tr.hip).  Return PDWT_OK when launched, 1 when the
// geometry / filter length is outside this path (caller uses k_ana_rows / k_syn_rows), < 0 on a HIP error.
#pragma once
#include "common.hpp"

namespace pdwt {
int ana_rows_tr_f64(const double* in, double* lo, double* hi, int Nr, int Nc, int hlen, const Taps2<double>& f);
int syn_rows_tr_f64(const double* a, const double* d, double* out, int Nr, int Nci, int Nco, int hlen, const Taps2<double>& f);
}  // namespace pdwt
