// cols_ring_dwt_f64.hip -- one slice of the register-ring column kernels (see cols_ring.inc)
#define PDWT_RING_PART 2
#include "cols_ring.inc"
