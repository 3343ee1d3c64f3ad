// cols_ring_swt_f64.hip -- one slice of the register-ring column kernels (see cols_ring.inc)
#define PDWT_RING_PART 4
#include "cols_ring.inc"
