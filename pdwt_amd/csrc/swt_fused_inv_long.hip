// swt_fused_inv_long.hip -- inverse levels for banks of 18 and 20 taps (swt_fused.inc, part 5)
#define PDWT_SWTF_PART 5
#include "swt_fused.inc"
