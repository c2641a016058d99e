// The lab is quarantined from the product.
//
// The fused kernels carry compile-time arms that exist for same-box A/B measurements: ablations that compute WRONG results on
// purpose (what does the weight ring / a tile read / an MFMA term cost?), alternative schedules and cache policies that compute
// the right ones.  A product build takes NONE of them:
//   * every such macro has its product default HERE and nowhere else;
//   * setting any of them on the command line without -DPNDF_EXPERIMENT=1 does not compile (the #error below);
//   * `__graft_entry__.build()` never passes a flag, `build_library()` refuses flags for the product path and adds the umbrella
//     for variant builds (tools/build_variants.py -> gpurun_ab/lib_<name>.so, loaded through PNDF_LIBRARY);
//   * every translation unit that includes this header exports `pndf_experiment_word_<tu>`: one bit per macro that differs from
//     its product default.  `pndf_experiment_word()` ORs them, `pndf_version()` prints the word, and tests/test_cabi.py reads
//     all of them from the built library: a product library reports experiments=0x00000000.
// Structural parameters of a wrapper translation unit (pndf_kernel_split_x2.hip: two DMA pieces per trunk slot; the *_timing.hip
// units: ring stamps) are set by the wrapper through PNDF_TU_* and are not experiments; overriding them with -D is one.
#pragma once

#ifndef PNDF_EXPERIMENT
#define PNDF_EXPERIMENT 0
#endif

#if !PNDF_EXPERIMENT
#if defined(PNDF_ABLATE) || defined(PNDF_SP_DIAG) || defined(PNDF_RING_ALIAS_F) || defined(PNDF_WRAP_SLOTS) || defined(PNDF_RING_SLOTS) || \
    defined(PNDF_RING_PIECES) || defined(PNDF_RING_STAMPS) || defined(PNDF_GROUP_STAMPS) || defined(PNDF_NT_MODE) || defined(PNDF_DMA_EARLY) || \
    defined(PNDF_MFMA_ORDER) || defined(PNDF_BIG_CT) || defined(PNDF_SPLIT_FOUR) || defined(PNDF_SP_FORM) || defined(PNDF_SP_FORM_OUT) || \
    defined(PNDF_SP_FORM_ENC) || defined(PNDF_SP_FORM_TILES) || defined(PNDF_SP_FORM_CHUNK) || defined(PNDF_SP_NT) || defined(PNDF_EXP_LO_BITS) || \
    defined(PNDF_LBS_DIAG) || defined(PNDF_LBS_FLA) || defined(PNDF_LBS_RLA) || defined(PNDF_LBS_PAIR_READS) || defined(PNDF_STAGGER) || \
    defined(PNDF_SP_WRAP) || defined(PNDF_GEN_ABLATE)
#error "an experiment macro is set without -DPNDF_EXPERIMENT=1: the product library takes no tuning / ablation macros (pndf_experiment.h)"
#endif
#endif

// ---- product defaults (the meaning of the other values is documented where each macro is used)
#ifndef PNDF_ABLATE
#define PNDF_ABLATE 0            // pndf_device.h / pndf_kernel_split.hip: timing ablations, WRONG results
#endif
#ifndef PNDF_SP_DIAG
#define PNDF_SP_DIAG 0           // pndf_kernel_split.hip (PNDF_SP_DIAG_DOC): softplus timing diagnostics, WRONG results
#endif
#ifdef PNDF_RING_ALIAS_F         // pndf_device.h: a sixth ring buffer over the pose tile, WRONG results
#define PNDF_X_RING_ALIAS_F 1
#else
#define PNDF_X_RING_ALIAS_F 0
#endif
#ifdef PNDF_WRAP_SLOTS           // pndf_device.h: wrapped stream footprint (with PNDF_ABLATE & 32), WRONG results
#define PNDF_X_WRAP_SLOTS 1
#else
#define PNDF_X_WRAP_SLOTS 0
#endif
#ifndef PNDF_RING_SLOTS
#define PNDF_RING_SLOTS 5        // pndf_device.h: ring depth (look-ahead 4); 3 / 4 are the arms of the latency-margin curve
#endif
#ifndef PNDF_TU_RING_PIECES      // structural: pndf_kernel_split_x2.hip sets 2 (hi tiles only)
#define PNDF_TU_RING_PIECES 4
#endif
#ifndef PNDF_RING_PIECES
#define PNDF_RING_PIECES PNDF_TU_RING_PIECES
#endif
#ifndef PNDF_TU_RING_STAMPS      // structural: the *_timing.hip units set 1
#define PNDF_TU_RING_STAMPS 0
#endif
#ifndef PNDF_RING_STAMPS
#define PNDF_RING_STAMPS PNDF_TU_RING_STAMPS
#endif
#ifndef PNDF_GROUP_STAMPS
#define PNDF_GROUP_STAMPS 0      // pndf_kernel_split.hip: per-group stamps of the instrumented kernel (a documented dead end)
#endif
#ifndef PNDF_NT_MODE
#define PNDF_NT_MODE 0           // pndf_kernel_split.hip: cache-policy bits on the slot fetches
#endif
#ifndef PNDF_DMA_EARLY
#define PNDF_DMA_EARLY 3         // pndf_kernel_split.hip: where the four DMA pieces of a slot are issued
#endif
#ifndef PNDF_MFMA_ORDER
#define PNDF_MFMA_ORDER 3        // pndf_kernel_split.hip: order of the MFMAs of a group (pair-major)
#endif
#ifndef PNDF_BIG_CT
#define PNDF_BIG_CT 2            // pndf_layout.h: chunk tiles of the two big phases
#endif
#ifndef PNDF_SPLIT_FOUR
#define PNDF_SPLIT_FOUR 0        // pndf_kernel_split.hip: the four-instruction operand split of rounds 2-4
#endif
#ifndef PNDF_SP_FORM
#define PNDF_SP_FORM 1           // pndf_device.h: softplus evaluation form (1 = packed pairs)
#endif
#ifndef PNDF_SP_FORM_OUT         // per-site overrides for bisection builds (default: PNDF_SP_FORM everywhere)
#define PNDF_SP_FORM_OUT PNDF_SP_FORM
#endif
#ifndef PNDF_SP_FORM_ENC
#define PNDF_SP_FORM_ENC PNDF_SP_FORM
#endif
#ifndef PNDF_SP_FORM_TILES
#define PNDF_SP_FORM_TILES PNDF_SP_FORM
#endif
#ifndef PNDF_SP_FORM_CHUNK
#define PNDF_SP_FORM_CHUNK PNDF_SP_FORM
#endif
#ifndef PNDF_SP_NT
#define PNDF_SP_NT 0             // pndf_device.h: non-temporal accesses to the softplus derivative scratch
#endif
#ifdef PNDF_EXP_LO_BITS          // pndf_capi.hip: the packer masks the lo halves of the weights (energy experiment), WRONG results
#define PNDF_X_EXP_LO_BITS 1
#else
#define PNDF_X_EXP_LO_BITS 0
#endif
#ifndef PNDF_LBS_DIAG
#define PNDF_LBS_DIAG 0          // pndf_lbs.hip: timing diagnostics of the body-model kernel, WRONG results
#endif
#ifndef PNDF_LBS_FLA
#define PNDF_LBS_FLA 2           // pndf_lbs.hip: forward steps whose LDS reads run ahead
#endif
#ifndef PNDF_LBS_RLA
#define PNDF_LBS_RLA 1           // pndf_lbs.hip: reverse row tiles whose LDS reads run ahead
#endif
#ifndef PNDF_LBS_PAIR_READS
#define PNDF_LBS_PAIR_READS 0
#endif
#ifndef PNDF_SP_WRAP
#define PNDF_SP_WRAP 0           // pndf_device.h: softplus derivative slots wrap after this many (same bytes, smaller footprint: does the
#endif                           // scratch cost what it costs because 216 MB + the rest overflow the 256 MB Infinity Cache?), WRONG results
#ifndef PNDF_GEN_ABLATE
#define PNDF_GEN_ABLATE 0        // pndf_generic.hip, timing arms only (WRONG results): 1 = no activation / derivative / gradient stores in the
#endif                           // layer epilogues, 2 = no derivative loads in the backward epilogues, 4 = no operand-tile DMA after a pass's
                                 // first two, 8 = no bias loads, 16 = no activation arithmetic, 32 = no vmcnt(0) at a pass's start, 64 = no weight-tile reads, 128 = no ring events; 256 = plain instead of non-temporal epilogue stores, 512 / 1024 = plain instead of non-temporal operand-tile fetches / derivative loads (correct results)
#ifndef PNDF_STAGGER
#define PNDF_STAGGER 0           // pndf_kernel_split.hip: workgroups of XCD x start x * PNDF_STAGGER sleeps (~4 us each) late (round 6:
#endif                           // does a chip whose XCDs are in different phases of a step sit closer to the power cap?)

// one bit per macro that differs from the product default of THIS translation unit
#define PNDF_EXPERIMENT_WORD                                                                                                        \
    (((PNDF_ABLATE) != 0 ? 1u << 0 : 0u) | ((PNDF_SP_DIAG) != 0 ? 1u << 1 : 0u) | (PNDF_X_RING_ALIAS_F ? 1u << 2 : 0u) |            \
     (PNDF_X_WRAP_SLOTS ? 1u << 3 : 0u) | ((PNDF_RING_SLOTS) != 5 ? 1u << 4 : 0u) | ((PNDF_RING_PIECES) != (PNDF_TU_RING_PIECES) ? 1u << 5 : 0u) | \
     ((PNDF_RING_STAMPS) != (PNDF_TU_RING_STAMPS) ? 1u << 6 : 0u) | ((PNDF_GROUP_STAMPS) != 0 ? 1u << 7 : 0u) |                     \
     ((PNDF_NT_MODE) != 0 ? 1u << 8 : 0u) | ((PNDF_DMA_EARLY) != 3 ? 1u << 9 : 0u) | ((PNDF_MFMA_ORDER) != 3 ? 1u << 10 : 0u) |     \
     ((PNDF_BIG_CT) != 2 ? 1u << 11 : 0u) | ((PNDF_SPLIT_FOUR) != 0 ? 1u << 12 : 0u) |                                              \
     (((PNDF_SP_FORM) != 1 || (PNDF_SP_FORM_OUT) != 1 || (PNDF_SP_FORM_ENC) != 1 || (PNDF_SP_FORM_TILES) != 1 || (PNDF_SP_FORM_CHUNK) != 1) ? 1u << 13 : 0u) | \
     ((PNDF_SP_NT) != 0 ? 1u << 14 : 0u) | (PNDF_X_EXP_LO_BITS ? 1u << 15 : 0u) | ((PNDF_LBS_DIAG) != 0 ? 1u << 16 : 0u) |          \
     (((PNDF_LBS_FLA) != 2 || (PNDF_LBS_RLA) != 1 || (PNDF_LBS_PAIR_READS) != 0) ? 1u << 17 : 0u) | ((PNDF_STAGGER) != 0 ? 1u << 18 : 0u) | ((PNDF_SP_WRAP) != 0 ? 1u << 19 : 0u) | ((PNDF_GEN_ABLATE) != 0 ? 1u << 20 : 0u) | ((PNDF_EXPERIMENT) != 0 ? 1u << 31 : 0u))

// `PNDF_EXPORT_EXPERIMENT_WORD(tag)` in a translation unit: its word as an exported constant of the shared library
#define PNDF_EXPORT_EXPERIMENT_WORD_(tag)                                                                  \
    extern "C" {                                                                                          \
    extern __attribute__((visibility("default"))) const unsigned pndf_experiment_word_##tag;             \
    __attribute__((used)) const unsigned pndf_experiment_word_##tag = PNDF_EXPERIMENT_WORD;              \
    }
#define PNDF_EXPORT_EXPERIMENT_WORD(tag) PNDF_EXPORT_EXPERIMENT_WORD_(tag)
