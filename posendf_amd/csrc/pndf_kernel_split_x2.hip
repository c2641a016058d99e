// Translation unit of the two-term split kernels (see the end of pndf_kernel_split.hip): same source, other instantiations,
// compiled in parallel with the three-term ones.
#define PNDF_SPLIT_X2_TU 1
#define PNDF_TU_TAG split_x2
#define PNDF_TU_RING_PIECES 2   // these kernels never read a lo tile: their trunk slots fetch the hi tiles only (pndf_device.h)
#include "pndf_kernel_split.hip"
