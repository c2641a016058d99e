// Translation unit of the plain-bf16 comparison kernel (see the end of pndf_kernel_split.hip): same source, the one-term
// instantiation with bfloat16 operands, compiled in parallel with the others.
#define PNDF_BF16_TU 1
#define PNDF_TU_TAG bf16
#include "pndf_kernel_split.hip"
