// swt_fused_fwd_long.hip -- forward levels for banks of 18 and 20 taps (swt_fused.inc, part 4)
#define PDWT_SWTF_PART 4
#include "swt_fused.inc"
