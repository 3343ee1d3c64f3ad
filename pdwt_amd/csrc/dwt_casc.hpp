// dwt_casc.hpp -- entry points of the two-levels-per-launch float32 streaming kernels (dwt_casc.hip).
// Return PDWT_OK when the two levels were launched, 1 when the geometry is outside this path
// (sizes not divisible by 4, small images, unsupported filter length, disabled) -> caller runs the
// levels one launch each.
#pragma once
#include "common.hpp"
#include "casc_dev.hpp"

namespace pdwt {
// forward levels l and l+1: in (nr x nc) -> H1,V1,D1 (nr/2 x nc/2) and A2,H2,V2,D2 (nr/4 x nc/4).
// The level-l approximation never goes to memory.  `trash` as in dwt_stream.hpp.
// d_tbl != NULL: nimg images in one launch (gridDim.y = image), the per-image pointers in a device-side table; the pointer arguments are
// those of image 0 (checked for alignment).  Workgroup form only: 1 when the geometry would take the independent-wave kernels.
int fwd2d_casc_f32(const float* in, float* H1, float* V1, float* D1, float* A2, float* H2, float* V2, float* D2, float* trash, int nr,
                   int nc, int hlen, const Taps2<float>& f, const CascBatchF* d_tbl = nullptr, int nimg = 1);
// inverse levels l+1 and l: A2,H2,V2,D2 (nr/4 x nc/4) + H1,V1,D1 (nr/2 x nc/2) -> out (nr x nc)
int inv2d_casc_f32(const float* A2, const float* H2, const float* V2, const float* D2, const float* H1, const float* V1, const float* D1,
                   float* out, float* trash, int nr, int nc, int hlen, const Taps2<float>& f);
}  // namespace pdwt
