// swt_fused_l2_inv4.hip -- inverse SWT levels for float32 banks of 22 ... 40 taps with the tap spacing 4 fixed at compile time
// (swt_fused_l2.inc, part 3)
#define PDWT_SWTL2_PART 3
#define PDWT_SWTL2_FSEL 4
#include "swt_fused_l2.inc"
