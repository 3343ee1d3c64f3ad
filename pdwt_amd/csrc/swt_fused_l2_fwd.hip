// swt_fused_l2_fwd.hip -- forward SWT levels for float32 banks of 22 ... 40 taps (swt_fused_l2.inc, part 1)
#define PDWT_SWTL2_PART 1
#include "swt_fused_l2.inc"
