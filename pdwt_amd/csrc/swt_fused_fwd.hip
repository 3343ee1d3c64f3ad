// swt_fused_fwd.hip -- forward half of swt_fused.inc (separate translation units: the unrolled bodies compile in parallel)
#define PDWT_SWTF_PART 1
#include "swt_fused.inc"
