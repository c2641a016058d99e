// Translation unit of the two-term split kernels (see the end of pndf_kernel_split.hip): same source, other instantiations,
// compiled in parallel with the three-term ones.
#define PNDF_SPLIT_X2_TU 1
#include "pndf_kernel_split.hip"
