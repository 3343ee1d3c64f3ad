// swt_fused_invp.hip -- inverse levels with residue-major LDS rows (swt_fused.inc, part 3)
#define PDWT_SWTF_PART 3
#include "swt_fused.inc"
