// libposendf_amd_debug.so -- the bring-up / profiling / measurement aids of include/posendf_amd_debug.h, in a library of their own
// (the product library carries no instrumented kernel, no stage-dump kernel, no probe and no pndf_debug_* symbol).
//
// What lives here: the s_memtime builds of the fused kernels (pndf_kernel_timing.hip, pndf_kernel_split_timing.hip), the stage-dump
// build of the exact-fp32 kernel (pndf_kernel_dbg.hip), the memory probes (pndf_probe.hip) and the entry points below.  They reach an
// engine of the PRODUCT library through its three pndf_internal_* hooks (pndf_capi.hip), whose addresses posendf_amd.engine hands
// to pndf_debug_bind() after loading both libraries -- no link-time dependency, so a variant product build (PNDF_LIBRARY) is paired
// with its own debug build and never with the default one.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/posendf_amd_debug.h"
#include "pndf_args.h"
#include "pndf_layout.h"

extern "C" __global__ void pndf_fused_split_relu_kernel_timing(PndfKernelArgs args);
extern "C" __global__ void pndf_fused_split_softplus_kernel_timing(PndfKernelArgs args);
extern "C" __global__ void pndf_fused_half_relu_kernel_timing(PndfKernelArgs args);
extern "C" __global__ void pndf_fused_relu_kernel_timing(PndfKernelArgs args);
extern "C" __global__ void pndf_fused_relu_kernel_dbg(PndfKernelArgs args);
extern "C" int pndf_kernel_timing_regions();
extern "C" int pndf_kernel_timing_layout(int what);
extern "C" int pndf_kernel_dbg_floats();

namespace {

typedef int (*launch_fn)(pndf_handle, int, const float*, const float*, float*, float*, int64_t, int, float*, void*, const void*);
typedef int (*describe_fn)(pndf_handle, int*, int);
typedef int (*fail_fn)(pndf_handle, int, const char*);
launch_fn g_launch = nullptr;
describe_fn g_describe = nullptr;
fail_fn g_fail = nullptr;

struct Desc {
    int precision, act, lo_all_zero, noenc, generic, resident, lds;
};
int describe(pndf_handle h, Desc& d) {
    if (!g_launch || !g_describe || !g_fail) return PNDF_ERR_BAD_ARG;      // pndf_debug_bind has not been called
    int w[7];
    const int rc = g_describe(h, w, 7);
    if (rc != PNDF_OK) return rc;
    d = Desc{w[0], w[1], w[2], w[3], w[4], w[5], w[6]};
    return PNDF_OK;
}
// the instrumented kernels need the same dynamic LDS as the product ones
int allow_lds(pndf_handle h, const void* kernel, int bytes) {
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) {
        (void)hipGetLastError();
        return g_fail(h, PNDF_ERR_HIP, "hipFuncSetAttribute failed for an instrumented kernel");
    }
    return PNDF_OK;
}

}  // namespace

extern "C" int pndf_debug_bind(void* internal_launch, void* internal_describe, void* internal_fail) {
    if (!internal_launch || !internal_describe || !internal_fail) return PNDF_ERR_BAD_ARG;
    g_launch = (launch_fn)internal_launch;
    g_describe = (describe_fn)internal_describe;
    g_fail = (fail_fn)internal_fail;
    return PNDF_OK;
}

extern "C" int64_t pndf_debug_floats(void) { return pndf_kernel_dbg_floats(); }

extern "C" int pndf_debug_forward_grad(pndf_handle h, const float* q, float* d, float* dq, int64_t B, float* dump, void* stream) {
    Desc ds;
    int rc = describe(h, ds);
    if (rc != PNDF_OK) return rc;
    if (!dump) return g_fail(h, PNDF_ERR_BAD_ARG, "dump is null");
    if (ds.generic) return g_fail(h, PNDF_ERR_UNSUPPORTED, "stage dumps and region timing exist for the amass.yaml-shaped kernels only");
    if (ds.precision != PNDF_PREC_FP32) return g_fail(h, PNDF_ERR_UNSUPPORTED, "the stage-dump kernel exists for fp32 precision only");
    if (ds.act == PNDF_ACT_SOFTPLUS) return g_fail(h, PNDF_ERR_UNSUPPORTED, "the debug dump exists for the relu-family kernel only");
    if (ds.noenc) return g_fail(h, PNDF_ERR_UNSUPPORTED, "the stage-dump kernel expects the structure encoder");
    const void* k = (const void*)pndf_fused_relu_kernel_dbg;
    if ((rc = allow_lds(h, k, ds.lds)) != PNDF_OK) return rc;
    return g_launch(h, MODE_FORWARD_GRAD, q, nullptr, dq, d, B, 1, dump, stream, k);
}

extern "C" int pndf_debug_timing_regions(void) { return pndf_kernel_timing_regions(); }
extern "C" int pndf_debug_timing_layout(int what) { return pndf_kernel_timing_layout(what); }

// project() through the instrumented build of the handle's kernel; cycles[(wg * 4 + wave) * regions + r] = shader cycles.  The
// instrumented builds exist for the three-term split kernels (relu family and softplus), the plain-f16 and the fp32 relu-family
// kernel: everything else is refused rather than timing another kernel than pndf_kernel_name() reports.
extern "C" int pndf_debug_project_timing(pndf_handle h, const float* q_in, float* q_out, int64_t B, int steps, unsigned long long* cycles,
                                         void* stream) {
    Desc ds;
    int rc = describe(h, ds);
    if (rc != PNDF_OK) return rc;
    if (!cycles) return g_fail(h, PNDF_ERR_BAD_ARG, "cycles is null");
    if (ds.generic) return g_fail(h, PNDF_ERR_UNSUPPORTED, "stage dumps and region timing exist for the amass.yaml-shaped kernels only");
    const bool sp = ds.act == PNDF_ACT_SOFTPLUS;
    if (ds.precision == PNDF_PREC_F16X3 && ds.lo_all_zero)
        return g_fail(h, PNDF_ERR_UNSUPPORTED, "no instrumented build of the two-term split kernels (set PNDF_THREE_TERMS=1 to time the three-term ones)");
    if (sp && ds.precision != PNDF_PREC_F16X3) return g_fail(h, PNDF_ERR_UNSUPPORTED, "softplus timing kernel: f16x3 only");
    if (ds.precision == PNDF_PREC_BF16) return g_fail(h, PNDF_ERR_UNSUPPORTED, "no instrumented build of the plain-bf16 comparison kernel");
    const void* k = sp ? (const void*)pndf_fused_split_softplus_kernel_timing
                    : ds.precision == PNDF_PREC_F16X3 ? (const void*)pndf_fused_split_relu_kernel_timing
                    : ds.precision == PNDF_PREC_F16 ? (const void*)pndf_fused_half_relu_kernel_timing
                                                     : (const void*)pndf_fused_relu_kernel_timing;
    if ((rc = allow_lds(h, k, ds.lds)) != PNDF_OK) return rc;
    return g_launch(h, MODE_PROJECT, q_in, nullptr, q_out, nullptr, B, steps, (float*)cycles, stream, k);
}

// the experiment words of this library's translation units (csrc/pndf_experiment.h): 0 in a product build
PNDF_EXPORT_EXPERIMENT_WORD(debug)
extern "C" {
extern const unsigned pndf_experiment_word_fp32_timing, pndf_experiment_word_split_timing, pndf_experiment_word_fp32_dbg;
}
extern "C" unsigned pndf_debug_experiment_word(void) {
    return pndf_experiment_word_debug | pndf_experiment_word_fp32_timing | pndf_experiment_word_split_timing | pndf_experiment_word_fp32_dbg;
}
