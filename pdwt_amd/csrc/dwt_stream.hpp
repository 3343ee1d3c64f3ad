// dwt_stream.hpp -- entry points of the float32 streaming level kernels (dwt_stream.hip).
// Return PDWT_OK when the level was launched, 1 when the geometry is outside the fast path
// (odd sizes, misaligned pointers, unsupported filter length) -> caller uses the tiled kernels.
#pragma once
#include "common.hpp"

namespace pdwt {
// `trash`: scratch that lane-predicated stores of halo lanes are redirected to: `trash_floats` >= 1024 floats (one
// 1024-float slot per workgroup modulo a power of two, up to 256 slots = kStreamTrashFloats)
constexpr size_t kStreamTrashFloats = 256 * 1024;
int fwd2d_stream_f32(const float* in, float* cA, float* cH, float* cV, float* cD, float* trash, size_t trash_floats, int nr, int nc, int hlen,
                     const Taps2<float>& f);
int inv2d_stream_f32(const float* cA, const float* cH, const float* cV, const float* cD, float* out, int nri, int nci, int nro, int nco,
                     int hlen, const Taps2<float>& f);
bool stream_enabled();

// ---- batched form: ONE launch runs the same level of `nimg` equally sized images (gridDim.y = image).  The per-image pointers live in a
// device-side table (built once per batch object: every pointer of a transform is fixed for its life).  Small images are
// launch-bound -- six launches per pair whatever the size -- so a batch of them amortises the launches (dwt.hip: pdwt_batch2d_*).
struct StreamBatchF {
    const float* in;
    float *cA, *cH, *cV, *cD, *trash;
};
struct StreamBatchI {
    const float *cA, *cH, *cV, *cD;
    float* out;
};
int fwd2d_stream_batch_f32(const StreamBatchF* d_tab, int nimg, size_t trash_floats, int nr, int nc, int hlen, const Taps2<float>& f);
int inv2d_stream_batch_f32(const StreamBatchI* d_tab, int nimg, int nri, int nci, int hlen, const Taps2<float>& f);
bool fwd2d_stream_takes(int nr, int nc, int hlen);   // geometry / filter length inside the streaming forward path
bool inv2d_stream_takes(int nri, int nci, int hlen);
}  // namespace pdwt
