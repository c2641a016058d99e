// The stage-dump build of the exact-fp32 kernel as its own translation unit (DEBUG library): per-stage register dumps of workgroup 0
// for tools/gpu_selfcheck.py / tests (pndf_debug_forward_grad).
#define PNDF_TU_TAG fp32_dbg
#define PNDF_DBG_TU
#include "pndf_kernel.hip"
