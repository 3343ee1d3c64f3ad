// swt_fused_l2_inv1.hip -- inverse SWT levels for float32 banks of 22 ... 40 taps with the tap spacing 1 fixed at compile time
// (swt_fused_l2.inc, part 3)
#define PDWT_SWTL2_PART 3
#define PDWT_SWTL2_FSEL 1
#include "swt_fused_l2.inc"
