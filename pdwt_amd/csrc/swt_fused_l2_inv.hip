// swt_fused_l2_inv.hip -- inverse SWT levels for float32 banks of 22 ... 40 taps (swt_fused_l2.inc, part 2)
#define PDWT_SWTL2_PART 2
#include "swt_fused_l2.inc"
