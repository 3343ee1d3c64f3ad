// swt_fused_f64_fwd.hip -- forward half of swt_fused_f64.inc
#define PDWT_SWTD_PART 1
#include "swt_fused_f64.inc"
