// swt_fused_inv.hip -- inverse half of swt_fused.inc
#define PDWT_SWTF_PART 2
#include "swt_fused.inc"
