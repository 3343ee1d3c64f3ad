// dwt_tail.hpp -- levels >= 2 of a 2D DWT fused into one launch per direction (dwt_tail.hip), float32.
// Return PDWT_OK when launched, 1 when the geometry / filter length is outside the path, negative on error.
#pragma once
#include "common.hpp"

namespace pdwt {
// forward: `in` = approximation of level first_level-1 (nr x nc); fills bands of levels first_level..nlevels-1 and c[0]
int fwd2d_tail_f32(const float* in, float** c, int first_level, int nlevels, int nr, int nc, int hlen, const Taps2<float>& f);
// inverse: reconstructs the approximation of level first_level-1 (nr x nc) into `out` from levels nlevels-1..first_level
int inv2d_head_f32(float* out, float** c, int first_level, int nlevels, int nr, int nc, int hlen, const Taps2<float>& f);
void tail_set_enabled(int on);
}  // namespace pdwt
