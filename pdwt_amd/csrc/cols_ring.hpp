// cols_ring.hpp -- register-ring column passes (cols_ring.inc).  Return PDWT_OK when launched, 1 when the
// filter length has no instantiation (caller falls back to the LDS-tiled kernels), negative on error.
#pragma once
#include "common.hpp"

namespace pdwt {
// (tB / caB != NULL: a second, independent branch of the same geometry in the same launch -- the (lo half -> A,H) and
// (hi half -> V,D) column passes of a 2-D level)
template <typename T> int ana_cols_ring(const T* t, T* lo, T* hi, int Nr, int Ncw, int hlen, const Taps2<T>& f, const T* tB = nullptr, T* loB = nullptr, T* hiB = nullptr);
template <typename T> int syn_cols_ring(const T* ca, const T* cd, T* out, int Nri, int Nc, int Nro, int hlen, const Taps2<T>& f, const T* caB = nullptr, const T* cdB = nullptr, T* outB = nullptr);
// stationary (a-trous) column passes at tap spacing fct (taps of the synthesis pre-halved by the caller)
template <typename T> int swt_ana_cols_ring(const T* t, T* lo, T* hi, int Nr, int Nc, int hlen, int fct, const Taps2<T>& f);
template <typename T> int swt_syn_cols_ring(const T* ca, const T* cd, T* out, int Nr, int Nc, int hlen, int fct, const Taps2<T>& f);
}  // namespace pdwt
