// swt_fused_f64_inv.hip -- inverse half of swt_fused_f64.inc
#define PDWT_SWTD_PART 2
#include "swt_fused_f64.inc"
