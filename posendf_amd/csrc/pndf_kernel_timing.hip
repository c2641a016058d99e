// The instrumented exact-fp32 kernel as its own translation unit: the same source as pndf_kernel.hip with the ring's sampled
// event stamps compiled in (pndf_device.h: PNDF_RING_STAMPS), so that the product kernels carry none of it.
#define PNDF_TU_RING_STAMPS 1
#define PNDF_TU_TAG fp32_timing
#define PNDF_TIMING_TU
#include "pndf_kernel.hip"
