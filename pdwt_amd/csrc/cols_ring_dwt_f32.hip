// cols_ring_dwt_f32.hip -- one slice of the register-ring column kernels (see cols_ring.inc)
#define PDWT_RING_PART 1
#include "cols_ring.inc"
