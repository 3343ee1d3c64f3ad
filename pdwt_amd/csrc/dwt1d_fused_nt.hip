// dwt1d_fused_nt.hip -- the fused batched-1D kernels once more with non-temporal loads (float32; see the head of dwt1d_fused.hip)
#define PDWT_1D_VARIANT nt
#define PDWT_1D_NT 5
#define PDWT_1D_FLOAT_ONLY 1
#include "dwt1d_fused.hip"
