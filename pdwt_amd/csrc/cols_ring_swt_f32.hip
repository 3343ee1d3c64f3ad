// cols_ring_swt_f32.hip -- one slice of the register-ring column kernels (see cols_ring.inc)
#define PDWT_RING_PART 3
#include "cols_ring.inc"
