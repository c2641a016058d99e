// The instrumented split-precision kernels as their own translation unit: the same source as pndf_kernel_split.hip with the
// ring's sampled event stamps compiled in (pndf_device.h: PNDF_RING_STAMPS), so that the product kernels carry none of it.
#define PNDF_TU_RING_STAMPS 1
#define PNDF_TU_TAG split_timing
#define PNDF_SPLIT_TIMING_TU
#include "pndf_kernel_split.hip"
