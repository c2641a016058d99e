/* posendf_amd_debug.h -- bring-up, profiling and measurement aids: the entry points of libposendf_amd_debug.so.
 *
 * NOT part of the drop-in boundary (include/posendf_amd.h) and NOT in the product library: libposendf_amd.so exports no pndf_debug_*
 * symbol and carries no instrumented kernel, stage-dump kernel or probe.  Nothing here has a counterpart in the reference, no
 * caller of the reference's path needs it, and a maintainer binding the library binds the public header only.  These entry points
 * exist for the repository's own tests (tests/test_timing_probe.py, tools/gpu_selfcheck.py) and for bench.py --diagnostics; they
 * may change or disappear between versions.  Plain C99 like the public header.
 *
 * The debug library has no link-time dependency on the product library: after loading both, the host hands it the addresses of
 * the product library's three pndf_internal_* hooks (pndf_debug_bind; posendf_amd.engine.load_library does it), so that a variant
 * product build is always instrumented by its own debug build.
 */
#ifndef POSENDF_AMD_DEBUG_H
#define POSENDF_AMD_DEBUG_H

#include "posendf_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* addresses of pndf_internal_launch, pndf_internal_describe, pndf_internal_fail of the PRODUCT library that created the handles
 * the calls below will see; every handle-taking entry point returns PNDF_ERR_BAD_ARG before this has been called */
int pndf_debug_bind(void* internal_launch, void* internal_describe, void* internal_fail);
/* csrc/pndf_experiment.h: the OR of this library's per-translation-unit experiment words; 0 in a product build */
unsigned pndf_debug_experiment_word(void);

/* Bring-up aid: forward_grad on the first 64 poses with per-stage register dumps of workgroup 0.
 * `dump` is a device buffer of pndf_debug_floats() floats. */
int pndf_debug_forward_grad(pndf_handle h, const float* q, float* d, float* dq, int64_t B, float* dump,
                            void* stream);
int64_t pndf_debug_floats(void);

/* Performance analysis aid: pndf_project through a kernel instrumented with s_memtime stamps.
 * cycles[(workgroup * 4 + wave) * pndf_debug_timing_regions() + r] = shader cycles of that wave, summed over the steps
 * (device buffer of ceil(B/64) * 4 * regions uint64).  A row is [regions of a step | per-group stamps | ring events]:
 * pndf_debug_timing_layout(0 / 1 / 2) = the three lengths.  Ring events (the weight ring's two synchronous events, sampled
 * every pndf_debug_timing_layout(3)-th slot): [0] cycles in the counted vmcnt wait = the slot's DMA had not landed, i.e. the
 * look-ahead of pndf_debug_timing_layout(4) - 1 slots did not cover the fetch latency; [1] cycles in the barrier; [2] sampled
 * slots; [3] what two stamps back to back measure (the floor of [0] / [2]). */
int pndf_debug_project_timing(pndf_handle h, const float* q_in, float* q_out, int64_t B, int steps,
                              unsigned long long* cycles, void* stream);
int pndf_debug_timing_regions(void);
int pndf_debug_timing_layout(int what);

/* Measurement aid: the memory subsystem of `device` as the fused kernels' weight stream sees it -- dependent-load latency
 * (ns) with the walked footprint resident in L2 (1 MiB), in the Infinity Cache (64 MiB) and in HBM (1 GiB): out[0..2];
 * streaming read bandwidth (GB/s): out[3]; wall-clock counter rate (MHz): out[4]; hops timed: out[5].  n_out >= 6; with
 * n_out >= 8 also the weight ring ALONE -- one workgroup per compute unit streaming 11 MB through a five-slot LDS ring with the
 * kernels' own instructions, waits and barriers, no arithmetic: GB/s delivered per compute unit while all of them run (out[6];
 * the f16x3 kernel consumes ~51) and ns per 16-KiB slot (out[7]).
 * Allocates 1 GiB for the duration of the call and synchronises the device (bench.py's `box` block). */
int pndf_debug_mem_probe(int device, double* out, int n_out);
/* The ring-only stream of the probe as a sustained load (`passes` walks of 11 MB by one workgroup per compute unit, synchronous;
 * seconds per pass in *sec_per_pass, may be NULL): tools/power_window.py --ring-only reads the package power under it. */
int pndf_debug_ring_stream(int device, int passes, double* sec_per_pass);

#ifdef __cplusplus
}
#endif
#endif /* POSENDF_AMD_DEBUG_H */
