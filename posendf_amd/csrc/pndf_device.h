// Device-side building blocks shared by the fp32 kernel (pndf_kernel.hip) and the split-precision kernel
// (pndf_kernel_split.hip): LDS carve, weight ring + LDS-DMA, activation helpers, the MFMA encoder.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pndf_layout.h"
#include "pndf_args.h"

using namespace pndf;

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define PNDF_GLOBAL __attribute__((address_space(1)))
#define PNDF_LDS __attribute__((address_space(3)))

namespace {

constexpr int SLOT_BYTES = SLOT_TILES * TILE_BYTES;          // 16 KiB
// LDS carve.  Every region is addressed as (one base register) + (16-bit immediate): hundreds of distinct
// constant LDS addresses above 64 KiB would each be materialised in an SGPR, hoisted out of the step loop
// and spilled.  The DMA ring sits at the bottom so that its M0 base stays below 64 KiB.
constexpr int FSTRIDE = 132;                                 // floats per pose in the feature buffer (bank skew)
// PNDF_RING_SLOTS (pndf_experiment.h): product 5 buffers = a slot is fetched FOUR slots ahead.  3 / 4 (look-ahead 2 / 3) are the arms
// of the latency-margin curve (profiles/r05/ring_margin.txt); 6 only with -DPNDF_RING_ALIAS_F (timing only: the feature buffer
// then overlays the pose tile -- WRONG results -- to make room for a sixth buffer)
constexpr int RING_SLOTS = PNDF_RING_SLOTS;
static_assert(RING_SLOTS >= 3 && RING_SLOTS <= 6, "ring depth");
constexpr int MASK_ROWS = 48;                                // u8 [48][256]: x1 8 chunks, x3 32 chunks, x5 4 chunks x 2 bytes
constexpr int LDS_RING = 0;                                  // 5 slots of 16 KiB (DMA target, lowest addresses)
constexpr int LDS_BIAS = LDS_RING + RING_SLOTS * SLOT_BYTES; // BIAS_FLOATS floats (trunk + encoder biases)
constexpr int LDS_MASK = LDS_BIAS + BIAS_FLOATS * 4;         // chunk-layer sign bits
constexpr int LDS_Q = LDS_MASK + MASK_ROWS * WG_THREADS;     // float [64][84]  the pose tile
#ifdef PNDF_RING_ALIAS_F
constexpr int LDS_F = LDS_Q;                                 // (timing-only arm of the margin curve, see PNDF_RING_SLOTS)
#else
constexpr int LDS_F = LDS_Q + WG_POSES * NQ * 4;             // float [64][FSTRIDE]  features, then d d / d feature,
#endif
constexpr int LDS_GN = LDS_F;                                //   then (same rows) d d / d n of the pose
constexpr int LDS_TOTAL = LDS_F + WG_POSES * FSTRIDE * 4;
static_assert(LDS_BIAS % 16 == 0 && LDS_MASK % 16 == 0 && LDS_Q % 16 == 0 && LDS_F % 16 == 0, "16-byte LDS carve");
static_assert(NQ <= FSTRIDE, "GN aliases the feature rows");
static_assert(LDS_TOTAL <= 160 * 1024, "LDS budget");


// debug dump stage offsets (floats per thread)
enum {
    DBG_FEAT = 0, DBG_X2 = 126, DBG_X4 = DBG_X2 + 128, DBG_X6 = DBG_X4 + 128, DBG_D = DBG_X6 + 16,
    DBG_G4 = DBG_D + 1, DBG_G2 = DBG_G4 + 128, DBG_G0 = DBG_G2 + 128, DBG_GN = DBG_G0 + 32,
    DBG_DQ = DBG_GN + 84, DBG_TOTAL = DBG_DQ + 84
};

// Weight ring: RING_SLOTS (5) slots of 16 tiles; slot i lives in buffer i % 5 and is fetched FOUR slots ahead.
// The one barrier per slot sits in the MIDDLE of the slot being consumed (tile 8): every wave has then left
// slot i-1, so its buffer can take the DMA of slot i+4; before the barrier each wave waits with a COUNTED
// vmcnt until its share of slot i+1 has landed (the DMAs of slots i+2, i+3 stay in flight).  Crossing a slot
// boundary needs no synchronisation and tile prefetch runs straight through.  Depth matters for the
// split-precision kernel, which consumes a slot in ~300 cycles: with 3 slots the latency budget of a DMA was
// one slot time and every L2 miss of the weight stream (3 %, i.e. almost every slot) stalled all four waves.
//
// Bookkeeping is kept to what one wave per SIMD can afford (every SALU / VALU instruction is ~4 cycles of the critical
// path): byte offsets instead of indices, and NO wrap test on the stream side -- the device copy of the stream is
// followed by a replica of its first RING_SLOTS - 1 slots, the fetch offset just keeps growing through a step and is
// pulled back by one step's length at the top of the next one (ring_next_step).  Per slot that is
//   boundary : s_add, s_cmp, s_cselect (buffer offset with wrap) + one v_add (this lane's read pointer)
//   mid slot : s_add (M0 = LDS destination) + one v_add (this lane's fetch offset)
struct Ring {
    const char* gstream;   // packed weight stream (global), followed by a replica of its first 4 slots
    char* smem;
    uint32_t fetch_off;    // per lane: stream byte offset of this lane's 16 B of the slot fetched LAST
    uint32_t cur_off;      // uniform: LDS byte offset (within the ring) of the buffer being consumed
    uint32_t prev_off;     // uniform: buffer of the previous slot = the one the mid-slot fetch refills
    uint32_t dst_base;     // uniform: LDS address of this wave's 4 KiB window of buffer 0
    int lane;
    uint32_t nfetch;       // (PNDF_ABLATE & 32 only) uniform: fetches issued so far in this step
    uint32_t lane_off;     // (PNDF_ABLATE & 32 only) per lane: this lane's byte offset inside a slot
    // (PNDF_RING_STAMPS only: the instrumented kernels' translation units) what the ring's two synchronous events cost this wave:
    unsigned long long st_wait, st_bar;   // shader cycles spent in the counted vmcnt wait / in the barrier, summed over the SAMPLED slots
    uint32_t st_n, st_k;                  // sampled slots, all slots
};
// PNDF_RING_STAMPS 1 (the *_timing.hip translation units): every RING_STAMP_PERIOD-th mid-slot event is bracketed by s_memtime stamps
constexpr uint32_t RING_STAMP_PERIOD = 16;      // 670 slots per step: the sampled positions rotate from step to step

// LDS-DMA of one 16 KiB slot: each wave moves 4 tiles (global_load_lds_dwordx4 = 1 KiB per instruction,
// LDS destination = M0 + lane * 16).  Issued from inline asm on purpose: when hipcc sees the builtin it
// degrades every `s_waitcnt lgkmcnt(N)` of the tile prefetch to lgkmcnt(0), which serialises ds_read and
// MFMA.  Consequence (cdna_hip_programming.md 5.7): the compiler does not count these loads, so every
// consumer-side barrier is preceded by an explicit `s_waitcnt vmcnt(N)`.
// One 1-KiB piece (tile 4*wave + j of the slot).  Piece 0 points M0 at the wave's LDS window, pieces 1..3 reuse
// it (the instruction offset moves both the global and the LDS address).
// Addressing: SGPR base (the stream pointer) + one 32-bit VGPR byte offset -- measured 21 issue cycles per piece
// between MFMAs against 30 for the 64-bit-VGPR-address form (tools/ubench/dma_cost.hip).
struct DmaSrc {
    const char* base;      // uniform
    uint32_t off;          // per lane: slot * SLOT_BYTES + wave * 4 KiB + lane * 16
};
// M0 is declared clobbered and never restored: hipcc treats M0 as a reserved scratch register that it sets right
// before each of its own uses (there is none in these kernels: the only M0 writes in the ISA are the ones below), so
// saving / restoring it only costs issue slots.  Pieces 1..3 rely on M0 still holding piece 0's value.
// PNDF_ABLATE (pndf_experiment.h; timing experiments ONLY, wrong results): 1 = no mid-slot barrier, 2 = no slot fetches after the
// first four, 4 = no counted vmcnt wait -- what each ring event costs (profiles/r02/ablation.txt); 8 / 16 = no third MFMA term / no
// lo-tile reads; 32 = the backward half of a step walks the forward half's slots in reverse (L2 reuse experiment; +64 = its control
// arm, -DPNDF_WRAP_SLOTS=N = wrapped footprint), 256 / 512 = every tile read / slot fetch of the split kernel issued twice
// (profiles/r02/ab_mirror_walk.txt, ab_additive.txt); 1024 / 2048 = no slot fetch in the backward / forward big phase (round 6)
__device__ __forceinline__ void ring_dma_piece(const DmaSrc& src, uint32_t dst, int j) {
    if (PNDF_ABLATE & 2) return;
    if (j == 0)
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                     : : "v"(src.off), "s"(src.base), "s"(dst) : "memory", "m0");
    else if (j == 1)
        asm volatile("global_load_lds_dwordx4 %0, %1 offset:1024" : : "v"(src.off), "s"(src.base) : "memory");
    else if (j == 2)
        asm volatile("global_load_lds_dwordx4 %0, %1 offset:2048" : : "v"(src.off), "s"(src.base) : "memory");
    else
        asm volatile("global_load_lds_dwordx4 %0, %1 offset:3072" : : "v"(src.off), "s"(src.base) : "memory");
}

// source / destination of this wave's share of the next slot to fetch (into the buffer of the previous slot)
__device__ __forceinline__ void ring_dma_begin(Ring& r, DmaSrc& src, uint32_t& dst) {
    if (PNDF_ABLATE & 32) {   // (L2 experiment: the backward half of a step re-reads the forward half's slots in reverse)
        const uint32_t k = r.nfetch;
        uint32_t zero;        // (PNDF_ABLATE & 64: control arm -- the same instructions, but the ordinary forward walk)
        asm volatile("s_mov_b32 %0, 0" : "=s"(zero));
#ifdef PNDF_WRAP_SLOTS     // (footprint experiment: the walk wraps after PNDF_WRAP_SLOTS slots, a power of two)
        const uint32_t s = (k + zero) & (uint32_t)(PNDF_WRAP_SLOTS - 1);
#else
        const uint32_t s = k < (uint32_t)FWD_SLOTS ? k : (PNDF_ABLATE & 64) ? zero + k : (uint32_t)STEP_SLOTS - 1 - k;
#endif
        r.nfetch = (k + 1 == (uint32_t)STEP_SLOTS) ? 0u : k + 1;
        r.fetch_off = r.lane_off + s * SLOT_BYTES - SLOT_BYTES;
    }
    r.fetch_off += SLOT_BYTES;
    src.base = r.gstream;
    src.off = r.fetch_off;
    dst = r.dst_base + r.prev_off;
}

__device__ __forceinline__ void ring_dma(Ring& r) {
    DmaSrc src;
    uint32_t dst;
    ring_dma_begin(r, src, dst);
#pragma unroll
    for (int j = 0; j < 4; ++j) ring_dma_piece(src, dst, j);
}

__device__ __forceinline__ void ring_wait_dma() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// at most two slot fetches (4 pieces each) of this wave may still be in flight
// Pieces (1 KiB each, per wave) that a TRUNK slot's fetch issues.  4 = all of the wave's four tiles.  2 (the two-term kernels'
// translation unit, pndf_kernel_split_x2.hip): only the hi tiles -- those kernels run networks whose lo tiles are all zero and
// never read them, so their half of the stream need not be delivered at all (round 5: the delivery of the weight stream into LDS
// is a fifth of a launch's energy and what pushes the kernel over the power cap).  Encoder slots and the ring's start always fetch all four.
static_assert(PNDF_RING_PIECES == 4 || PNDF_RING_PIECES == 2, "pieces per trunk slot");
static_assert(PAIR_HI_TILE == 0 && PAIR_LO_TILE == 1, "PNDF_RING_PIECES == 2 skips the ODD pieces of a wave's window: they must be the lo tiles (pndf_layout.h)");
// the counted wait before the mid-slot barrier: at most this many of the wave's fetch operations may still be in flight -- the
// pieces of the slots after the next one, counted with the SMALLEST number a slot can issue (a slot that issued more only makes
// the wait stricter)
constexpr int RING_INFLIGHT = PNDF_RING_PIECES * (RING_SLOTS - 3);      // product: 8
__device__ __forceinline__ void ring_wait_next_slot() {
    if (PNDF_SP_DIAG & 32) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");      // (timing diagnostic only: see pndf_kernel_split.hip)
    else asm volatile("s_waitcnt vmcnt(%0)" : : "n"(RING_INFLIGHT) : "memory");
}

// slots 0..3 into buffers 0..3; the caller waits vmcnt(0) + barrier.  The first slot boundary makes buffer 0 current
// and buffer 4 "previous", so the first mid-slot fetch (slot 4) fills buffer 4.
__device__ __forceinline__ void ring_start(Ring& r, int wave) {
    r.dst_base = (uint32_t)(size_t)(PNDF_LDS char*)(r.smem + LDS_RING) + wave * (4 * TILE_BYTES);
    r.fetch_off = wave * (4 * TILE_BYTES) + r.lane * 16 - SLOT_BYTES;
    r.nfetch = 0;
    r.st_k = 0;
    r.lane_off = wave * (4 * TILE_BYTES) + r.lane * 16;
    r.prev_off = 0;
#pragma unroll
    for (int b = 0; b < RING_SLOTS - 1; ++b) {
        ring_dma(r);          // (PNDF_ABLATE & 2: the ring then keeps whatever it held -- timing only)
        r.prev_off += SLOT_BYTES;
    }
    r.cur_off = (RING_SLOTS - 1) * SLOT_BYTES;
}

// PNDF_RING_PIECES == 2 only.  The trunk's slot fetches run RING_SLOTS - 1 slots ahead, i.e. the last ones of the backward
// trunk target the encoder's backward section and the first slot of the next step -- fp32 tiles, all of them needed -- and
// brought only their even tiles.  Once per step, between the trunk and the encoder: fetch the odd tiles of those slots (the
// buffers after the current one, in ring order; the fetch pointer stands at the last of them), then drain.  The caller's
// workgroup barrier publishes them.
__device__ __forceinline__ void ring_complete_lookahead(Ring& r) {
    if constexpr (PNDF_RING_PIECES == 2) {
        uint32_t boff = r.cur_off;
#pragma unroll
        for (int k = 0; k < RING_SLOTS - 1; ++k) {
            boff = (boff == (RING_SLOTS - 1) * SLOT_BYTES) ? 0u : boff + SLOT_BYTES;
            const uint32_t off = r.fetch_off - (uint32_t)(RING_SLOTS - 2 - k) * SLOT_BYTES;
            const uint32_t dst = r.dst_base + boff;
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024\n\t"
                         "global_load_lds_dwordx4 %0, %1 offset:3072"
                         : : "v"(off), "s"(r.gstream), "s"(dst) : "memory", "m0");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

// top of projection step 1, 2, ...: the fetch pointer has run one step's length (into the replica of the first slots)
__device__ __forceinline__ void ring_next_step(Ring& r) { r.fetch_off -= (uint32_t)STEP_SLOTS * SLOT_BYTES; }

// tile 0 of a slot: switch buffers (no barrier needed, see above)
__device__ __forceinline__ void ring_boundary(Ring& r) {
    r.prev_off = r.cur_off;
    r.cur_off = (r.cur_off == (RING_SLOTS - 1) * SLOT_BYTES) ? 0u : r.cur_off + SLOT_BYTES;
}

// tile 8 of a slot: one barrier, then prefetch four slots ahead into the buffer of the previous slot
// A raw s_barrier, not __syncthreads(): the latter's fence adds `s_waitcnt lgkmcnt(0)`, i.e. it waits for the
// tile prefetch issued a few instructions earlier.  What the barrier has to order is already ordered: every
// wave's DMA share of the next slot has landed (its own counted vmcnt above), and every read of the previous slot
// returned long ago (its data has been consumed by MFMAs issued before this point).
__device__ __forceinline__ void ring_midslot_sync(Ring& r) {
    if constexpr (PNDF_RING_STAMPS != 0) {
        // Instrumented kernels: how long does this wave sit in the counted wait (= the slot's DMA had not landed: the ring's
        // look-ahead did not cover the fetch latency) and in the barrier (= the other waves were not there yet)?  One asm
        // statement, so that nothing is scheduled between the stamps; s_memtime returns through lgkmcnt, i.e. a sampled event
        // also drains the tile prefetch -- hence every RING_STAMP_PERIOD-th event only.
        if ((r.st_k++ % RING_STAMP_PERIOD) == 0) {
            unsigned long long t0, t1, t2;
            asm volatile("s_memtime %0\n\ts_waitcnt vmcnt(%3)\n\ts_memtime %1\n\ts_barrier\n\ts_memtime %2\n\ts_waitcnt lgkmcnt(0)"
                         : "=&s"(t0), "=&s"(t1), "=&s"(t2) : "n"(RING_INFLIGHT) : "memory");
            r.st_wait += t1 - t0;
            r.st_bar += t2 - t1;
            r.st_n++;
            return;
        }
    }
    if (!(PNDF_ABLATE & 4)) ring_wait_next_slot();
    if (!(PNDF_ABLATE & 1)) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ f32x4 ring_tile(const Ring& r, int t_in_slot) {
    return *(const f32x4*)(r.smem + LDS_RING + r.cur_off + t_in_slot * TILE_BYTES + r.lane * 16);
}

// Lanes of ONE wave exchange data through LDS (lane group 0 stores, all lane groups load).  The hardware
// executes a wave's LDS operations in order, but for the compiler this is inter-thread communication: without
// a fence it may satisfy the later loads from before the (other lanes') stores -- it did: lane groups 1..3 read
// stale d d / d n.  A wavefront-scope fence costs no instructions and restores the ordering.
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// chunk-layer sign bits: one byte per lane per 8 bits; a CT = 4 chunk (16 bits) uses two consecutive rows
template <int CT>
__device__ __forceinline__ void store_chunk_bits(uint8_t* mask, int c, uint32_t bits) {
    if constexpr (CT == 2) {
        mask[c * WG_THREADS] = (uint8_t)bits;
    } else {
        mask[(2 * c) * WG_THREADS] = (uint8_t)bits;
        mask[(2 * c + 1) * WG_THREADS] = (uint8_t)(bits >> 8);
    }
}
template <int CT>
__device__ __forceinline__ uint32_t load_chunk_bits(const uint8_t* mask, int c) {
    if constexpr (CT == 2) return mask[c * WG_THREADS];
    else return (uint32_t)mask[(2 * c) * WG_THREADS] | ((uint32_t)mask[(2 * c + 1) * WG_THREADS] << 8);
}

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// relu family: slope = 0 (relu) or 0.01 (lrelu).  PyTorch conventions: relu'(0) = 0, lrelu'(0) = slope.
__device__ __forceinline__ float act_relu(float z, float slope, bool& pos) {
    pos = z > 0.0f;
    return pos ? z : z * slope;
}

// Branch-free form used on the MFMA path: step(z) = (z > 0 ? 1.0f : 0.0f) EXACTLY for every finite z, built
// from two multiplies with the free [0,1] clamp output modifier (no v_cmp / VCC / v_cndmask chain).
// z = +-0 -> 0, z < 0 -> 0, smallest denormal 2^-149 * 2^64 * 2^127 >= 1 -> 1.
__device__ __forceinline__ float step01(float z) {
    float t;
    asm("v_mul_f32 %0, %1, %2 clamp" : "=v"(t) : "v"(z), "s"(0x1p64f));
    float u;
    asm("v_mul_f32 %0, %1, %2 clamp" : "=v"(u) : "v"(t), "s"(0x1p127f));
    return u;
}
// derivative factor (1 or slope) from the step; activation = z * factor (relu: -0 for z < 0)
__device__ __forceinline__ float relu_factor(float step, float slope) { return fmaf(step, 1.0f - slope, slope); }

// nn.Softplus(beta, threshold=20) (net_modules.py:39-40): x if beta x > 20 else log1p(exp(beta x)) / beta;
// derivative as PyTorch's softplus_backward: e / (e + 1) with e = exp(beta x), 1 above the threshold.
// On the hardware transcendentals (v_exp_f32 / v_log_f32 / v_rcp_f32, ~1 ulp each) instead of libm's expf / log1pf /
// IEEE division (~70 instructions per value: they made the softplus kernel 57 % slower than the relu one).
//
// PNDF_SP_FORM 1 (round 4, default): 10 plain instructions per value instead of 16, written on PAIRS of values so that they
// issue as packed fp32 (v_pk_mul / v_pk_fma / v_pk_add: two values per issue slot) -- the split softplus kernel is bound by
// VALU issue, the three quarter-rate transcendentals per value are the floor:
//   x = min(z beta log2 e, 20 log2 e);  e = 2^x;  u = 1 + e;  ru = 1 / u;
//   softplus = max(z, (ln 2 / beta) log2 u + ((e - (u - 1)) ru) / beta);   derivative = e ru
// * the clamp replaces both selects of form 0: above the threshold u = e exactly (e > 2^24), so the clamped value is
//   (20 + 2e-9) / beta < z and max() returns z as PyTorch does, while e ru = 1 to an ulp; below it softplus(z) > z, and max()
//   returns the computed value (at beta z within rounding of 20 the two agree to an ulp);
// * (e - (u - 1)) ru puts the rounding of 1 + e back to first order: a saturated-low unit keeps softplus = e / beta, not 0.
// PNDF_SP_FORM 0: rounds 1-3 (kept for same-box A/B builds).
typedef float f32x2 __attribute__((ext_vector_type(2)));
struct SpK {            // uniform constants of the activation
    float beta, b2, c, invb;      // beta, beta log2(e), ln 2 / beta, 1 / beta
};
__device__ __forceinline__ SpK sp_consts(float beta) {
    const float invb = __builtin_amdgcn_rcpf(beta);
    return SpK{beta, beta * 1.44269504088896341f, 0.693147180559945309f * invb, invb};
}
constexpr float SP_CLAMP_LOG2 = 28.8539008177792681f;      // 20 log2(e): the threshold of nn.Softplus in the exponent's unit
// one v_min / v_max: fminf() / fmaxf() make hipcc quiet both operands first (a second instruction per value)
__device__ __forceinline__ float vmin1(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float vmax1(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// The same v_max whose result may be the A / B operand of the NEXT instruction, an MFMA: a VGPR written by a VALU
// instruction needs wait states before an MFMA reads it as an operand, hipcc pads them for its own instructions but not for
// the inside of an asm statement (cdna_hip_programming.md 5.7 item 2) -- without the pad the encoder's MFMAs read the
// PREVIOUS contents of the register (found by bisecting the sites of form 1 on the hardware, round 4: every site whose
// result passes through another VALU instruction first was right, the encoder -- activation straight into the next
// layer's MFMA -- was wrong).
__device__ __forceinline__ float vmax1_mfma(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2\n\ts_nop 1" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float vmax3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// (per-site overrides for bisection builds, PNDF_SP_FORM_{OUT,ENC,TILES,CHUNK}: pndf_experiment.h)
template <int FORM = PNDF_SP_FORM>
__device__ __forceinline__ float act_softplus(float z, const SpK& k, float& deriv) {
  if constexpr (FORM == 0) {
    const float bz = z * k.beta;
    const float e = __builtin_amdgcn_exp2f(fminf(bz, 20.0f) * 1.44269504088896341f);
    const bool lin = bz > 20.0f;
    const float u = 1.0f + e;
    const float ru = __builtin_amdgcn_rcpf(u);
    const float l = fmaf(e - (u - 1.0f), ru, __builtin_amdgcn_logf(u) * 0.693147180559945309f);
    deriv = lin ? 1.0f : e * ru;
    return lin ? z : l * k.invb;
  } else {
    const float e = __builtin_amdgcn_exp2f(vmin1(z * k.b2, SP_CLAMP_LOG2));
    const float u = 1.0f + e;
    const float ru = __builtin_amdgcn_rcpf(u);
    const float t = (e - (u - 1.0f)) * ru;
    deriv = e * ru;
    return vmax1(z, fmaf(t, k.invb, __builtin_amdgcn_logf(u) * k.c));
  }
}

// two values at once: the plain instructions as packed fp32
// TO_MFMA: the activated values are MFMA operands as they are (the encoder, the fp32 kernel's trunk)
template <int FORM = PNDF_SP_FORM, bool TO_MFMA = false>
__device__ __forceinline__ f32x2 act_softplus2(f32x2 z, const SpK& k, f32x2& deriv) {
  if constexpr (FORM == 0) {
    f32x2 y;
    float d0, d1;
    y[0] = act_softplus<0>(z[0], k, d0);
    y[1] = act_softplus<0>(z[1], k, d1);
    deriv = f32x2{d0, d1};
    return y;
  } else {
    const f32x2 x = z * k.b2;
    f32x2 e;
    e[0] = __builtin_amdgcn_exp2f(vmin1(x[0], SP_CLAMP_LOG2));
    e[1] = __builtin_amdgcn_exp2f(vmin1(x[1], SP_CLAMP_LOG2));
    const f32x2 u = e + 1.0f;
    f32x2 ru, lg;
    ru[0] = __builtin_amdgcn_rcpf(u[0]);
    ru[1] = __builtin_amdgcn_rcpf(u[1]);
    lg[0] = __builtin_amdgcn_logf(u[0]);
    lg[1] = __builtin_amdgcn_logf(u[1]);
    const f32x2 t = (e - (u - 1.0f)) * ru;
    // which of the two products is fused with the sum is pinned (the two-term and three-term split kernels must agree bit
    // for bit: left to -ffp-contract each instantiation chose for itself)
    const f32x2 sp = __builtin_elementwise_fma(t, f32x2{k.invb, k.invb}, lg * k.c);
    deriv = e * ru;
    if constexpr (TO_MFMA) return f32x2{vmax1_mfma(z[0], sp[0]), vmax1_mfma(z[1], sp[1])};
    else return f32x2{vmax1(z[0], sp[0]), vmax1(z[1], sp[1])};
  }
}
// a whole C/D tile register set (4 values per lane)
template <int FORM = PNDF_SP_FORM, bool TO_MFMA = false>
__device__ __forceinline__ void act_softplus4(f32x4& z, const SpK& k, f32x4& deriv) {
    f32x2 d0, d1;
    const f32x2 y0 = act_softplus2<FORM, TO_MFMA>(f32x2{z[0], z[1]}, k, d0), y1 = act_softplus2<FORM, TO_MFMA>(f32x2{z[2], z[3]}, k, d1);
    z = f32x4{y0[0], y0[1], y1[0], y1[1]};
    deriv = f32x4{d0[0], d0[1], d1[0], d1[1]};
}

// Activation parameters + where derivatives are parked between the forward and the backward pass.
//   relu family : sign bits (chunk layers: one u16 per lane per chunk in LDS; accumulator layers: registers)
//   softplus    : fp32 derivatives in a per-workgroup global scratch, one float4 per lane per tile ("slot")
// The derivative scratch of a workgroup ([slot][256 lanes] float4) is addressed as ONE uniform 64-bit base (an SGPR pair)
// + a 32-bit per-lane byte offset: `global_load/store v_off, .., s[base:base+1]`.  As per-lane 64-bit pointers (round 1)
// the ~200 slot addresses were computed ahead, hoisted and spilled -- and every spill reload is a VMEM load whose
// vmcnt(0) drains the ring's DMA.
// bytes of a derivative tile per lane: 16 = four fp32 values.  PNDF_SP_DIAG & 256 (timing arm, WRONG results): 12 -- what a
// 3-byte storage format of the derivative could buy at most (its bytes without its pack / unpack instructions; the fourth value
// of a tile is then not stored at all); profiles/r06/small_arms.txt
typedef float f32x3 __attribute__((ext_vector_type(3)));
constexpr uint32_t SP_LANE_BYTES = (PNDF_SP_DIAG & 256) ? 12u : 16u;
struct SpRef {
    const char* base;   // uniform: this workgroup's block of the scratch
    uint32_t off;       // per lane: tid * SP_LANE_BYTES
    __device__ __forceinline__ f32x4* slot(int s) const {
        if constexpr (PNDF_SP_WRAP != 0) s %= PNDF_SP_WRAP;      // (footprint experiment, pndf_experiment.h: WRONG results)
        return (f32x4*)(const_cast<char*>(base) + (uint32_t)(off + (uint32_t)s * (WG_THREADS * SP_LANE_BYTES)));
    }
    // the parked derivatives are written once and read once, 843 KB per workgroup and step: PNDF_SP_NT marks these accesses
    // non-temporal (1 = the chunk layers' stores, 2 = every store, 4 = the loads) so that they do not push the weight stream out of L2
    template <int KIND>
    __device__ __forceinline__ void put(int s, const f32x4& v) const {
        if constexpr (SP_LANE_BYTES == 12) *(f32x3*)slot(s) = f32x3{v[0], v[1], v[2]};
        else if constexpr ((PNDF_SP_NT & KIND) != 0) __builtin_nontemporal_store(v, slot(s));
        else *slot(s) = v;
    }
    __device__ __forceinline__ f32x4 get(int s) const {
        if constexpr (SP_LANE_BYTES == 12) {
            const f32x3 v = *(const f32x3*)slot(s);
            return f32x4{v[0], v[1], v[2], v[2]};
        } else if constexpr ((PNDF_SP_NT & 4) != 0) return __builtin_nontemporal_load(slot(s));
        else return *slot(s);
    }
};
struct ActP {
    float slope;        // relu family
    SpK k;              // softplus
    SpRef sp;           // softplus: this thread's column of the scratch, else {null, 0}
    char* stage;        // softplus backward: this wave's LDS staging window for derivative tiles (its own F rows,
    int lane;           //   free between the forward trunk and the end of the backward trunk); lane id
};

// One parked derivative tile of this wave (64 lanes x 16 B, contiguous in the scratch) -> LDS by DMA.  The chunk
// layers' derivatives must not come in as ordinary global loads: those share the in-order vmcnt queue with the ring's
// inline-asm DMA, which hipcc cannot see, so its own `s_waitcnt vmcnt(0)` at the point of use would drain every DMA
// piece in flight.  The consumer waits with `vmcnt(N)`, N = number of ring pieces certainly issued in between.
__device__ __forceinline__ void stage_derivative_tile(const SpRef& sp, int slot, char* stage_tile) {
    // wave-uniform by construction (the wave's window); readfirstlane keeps it in an SGPR whatever hipcc infers
    const uint32_t dst = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(PNDF_LDS char*)stage_tile);
    if constexpr (PNDF_SP_WRAP != 0) slot %= PNDF_SP_WRAP;
    const uint32_t off = sp.off + (uint32_t)slot * (WG_THREADS * SP_LANE_BYTES);
    if constexpr (SP_LANE_BYTES == 12)      // (timing arm: 12 bytes per lane land at dst + lane * 12)
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx3 %0, %1" : : "v"(off), "s"(sp.base), "s"(dst) : "memory", "m0");
    else if constexpr ((PNDF_SP_NT & 4) != 0)
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt" : : "v"(off), "s"(sp.base), "s"(dst) : "memory", "m0");
    else
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(off), "s"(sp.base), "s"(dst) : "memory", "m0");
}
// this lane's values of staged tile `ci` of the wave's window
__device__ __forceinline__ f32x4 staged_derivative_tile(const char* stage, int ci, int lane) {
    if constexpr (SP_LANE_BYTES == 12) {
        const float* p = (const float*)(stage + ci * 1024 + lane * 12);
        return f32x4{p[0], p[1], p[2], p[2]};
    } else {
        return *(const f32x4*)(stage + ci * 1024 + lane * 16);
    }
}
template <int YOUNGER>
__device__ __forceinline__ void wait_staged_derivatives() {
    static_assert(YOUNGER >= 0 && YOUNGER < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(YOUNGER) : "memory");
}
constexpr int SP_SLOT_CHUNK[3] = {0, 16, 80};      // chunk layers x1 (8x2), x3 (32x2), x5 (4x4)
constexpr int SP_SLOT_X2 = 96, SP_SLOT_X4 = 128, SP_SLOT_X6 = 160, SP_SLOT_ENC = 164;   // encoder: 2 tiles per joint
constexpr int SP_SLOTS = SP_SLOT_ENC + 2 * NJ;
constexpr int SP_WG_FLOATS = SP_SLOTS * WG_THREADS * 4;

constexpr int TIMING_REGIONS = 12;
constexpr int TIMING_GROUPS = 32;     // per-group stamps inside the (lin2,lin3) phase
constexpr int TIMING_RING = 4;        // ring events (PNDF_RING_STAMPS): wait cycles, barrier cycles, sampled slots, cycles of two stamps back to back
// s_memtime stamps at region boundaries (TIMING instantiation only): accumulates shader cycles per region
struct RegionClock {
    unsigned long long acc[TIMING_REGIONS];
    unsigned long long grp[TIMING_GROUPS];
    unsigned long long last;
};
// the ring's sampled events of this wave -> out[0 .. TIMING_RING) (zeros from a kernel built without PNDF_RING_STAMPS)
__device__ __forceinline__ void ring_stamps_out(const Ring& r, unsigned long long* out) {
    unsigned long long a = 0, b = 0;
    if constexpr (PNDF_RING_STAMPS != 0)      // what two stamps back to back measure: the floor of `st_wait / st_n`
        asm volatile("s_memtime %0\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)" : "=&s"(a), "=&s"(b) : : "memory");
    out[0] = PNDF_RING_STAMPS ? r.st_wait : 0ull;
    out[1] = PNDF_RING_STAMPS ? r.st_bar : 0ull;
    out[2] = PNDF_RING_STAMPS ? (unsigned long long)r.st_n : 0ull;
    out[3] = b - a;
}
template <bool TIMING>
__device__ __forceinline__ void tick(RegionClock& rc, int region) {
    if constexpr (TIMING) {
        const unsigned long long now = __builtin_amdgcn_s_memtime();
        rc.acc[region] += now - rc.last;
        rc.last = now;
    }
}

// does the tile group [T0, T0 + GT) contain the mid-slot tile?
template <int GT, int T0>
constexpr bool group_has_mid() {
    for (int i = 0; i < GT; ++i)
        if ((T0 + i) % SLOT_TILES == SLOT_TILES / 2) return true;
    return false;
}

template <int GT, int T0>
__device__ __forceinline__ void load_group(f32x4 (&a)[GT], Ring& ring) {
#pragma unroll
    for (int i = 0; i < GT; ++i) {
        const int t = (T0 + i) % SLOT_TILES;    // compile-time: T0 is a template constant, i unrolled
        if (t == 0) ring_boundary(ring);
        if (t == SLOT_TILES / 2) ring_midslot_sync(ring);
        a[i] = ring_tile(ring, t);
    }
}

// ------------------------------------------------------------------ encoder on the MFMA pipe
// Joint J, forward (reference net_modules.py:75-111,162-168), batched over the wave's 16 poses:
//   X  (B layout, k = 4g+s): lane group 0 = normalised quaternion of the joint, groups 1-2 = the parent's
//       six features -- which is exactly where the parent's output tile holds them (rows 4..9)
//   H  = act(W1 X + b1)    rows 0..9  = hidden units        (tile 2J of the encoder slots)
//   F  = act(W2 H + b2)    rows 4..9  = features            (tile 2J+1)
// All padded rows/columns carry zero weights, so padding values never reach a real output.
// Derivatives: relu family = 8 sign bits per lane per joint (registers); softplus = two scratch slots.
template <int T>
__device__ __forceinline__ f32x4 enc_tile(Ring& ring) {
    constexpr int t = T % SLOT_TILES;
    if (t == 0) ring_boundary(ring);
    if (t == SLOT_TILES / 2) {
        ring_midslot_sync(ring);
        ring_dma(ring);
    }
    return ring_tile(ring, t);
}

// per-component denominators of F.normalize(pose, dim=1): max(||q[:, c]||_2 over joints, eps)
// POISON (softplus instantiations): also returns 0 * (every component of the pose), i.e. +0 for a finite pose and NaN for
// one that holds a NaN or an infinity -- as F.normalize makes of it (inf / inf).  nn.Softplus carries a NaN through
// (softplus(NaN) = NaN); the hardware v_min / v_max of the activation return their other operand, so the kernels add this
// to the pose's distance and gradient seed instead (0 + x is exact for every finite x).
template <bool POISON = false>
__device__ __forceinline__ float joint_axis_norms(const float* my_q, float (&ss)[4]) {
    ss[0] = ss[1] = ss[2] = ss[3] = 0.f;
    float poison = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const f32x4 v = *(const f32x4*)(my_q + 4 * j);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            ss[c] = fmaf(v[c], v[c], ss[c]);
            if constexpr (POISON) poison = fmaf(v[c], 0.0f, poison);
        }
    }
    return poison;
}

// activation of one encoder tile; returns 4 derivative bits (relu family) or stores the derivative (softplus)
template <bool SP>
__device__ __forceinline__ float enc_act(f32x4& z, const ActP& ap, int spslot) {
    float bitsum = 0.f;
    if constexpr (SP) {
        f32x4 dv;
        act_softplus4<PNDF_SP_FORM_ENC, true>(z, ap.k, dv);
        ap.sp.put<2>(spslot, dv);
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float st = step01(z[r]);
            z[r] = z[r] * relu_factor(st, ap.slope);
            bitsum = fmaf(st, (float)(1u << r), bitsum);
        }
    }
    return bitsum;
}

template <bool SP>
__device__ __forceinline__ void enc_dact(f32x4& gz, uint32_t bits4, const ActP& ap, const f32x4& dpre) {
    if constexpr (SP) {
        gz = gz * dpre;              // derivative fetched from the scratch at the start of encoder_backward
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) gz[r] = gz[r] * relu_factor((float)((bits4 >> r) & 1u), ap.slope);
    }
}

// Joints J and J+1 are processed as a PAIR when they are independent (J+1 is not a child of J): their MFMA chains
// (four dependent 16x16x4 steps per layer, 40 cycles of latency each at 32 cycles of issue) and activation VALU are
// interleaved by hand, which hides the dependent latency that made the encoder 9.5 % of a split-precision step.
// SMPL order: pairs (0,1) (2,3) ... (14,15), then the chain 16 -> 17 -> 18 -> 19 -> 20 one joint at a time.
constexpr bool enc_pairable(int j) { return j + 1 < NJ && PARENT[j + 1] != j; }

template <int J>
__device__ __forceinline__ f32x4 enc_input(const f32x4& qj, const float (&inv)[4], const f32x4 (&F)[NJ], int g) {
    f32x4 X;
#pragma unroll
    for (int c = 0; c < 4; ++c) X[c] = qj[c] * inv[c];         // posendf.py:71 (x / max(norm, eps), as x * (1 / .))
    if constexpr (PARENT[J] >= 0) X = (g == 0) ? X : F[PARENT[J]];   // cat(quat, parent feature), net_modules.py:167
    else X = (g == 0) ? X : f32x4{0.f, 0.f, 0.f, 0.f};
    return X;
}

template <int J>
__device__ __forceinline__ void enc_store_features(float* my_f, const f32x4& Fj, int g) {
    // features of the joint -> per-pose buffer (rows 4..7 live in lane group 1, rows 8..9 in lane group 2):
    // two predicated 8-byte stores
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    float* dst = my_f + FEAT * J + ((g == 2) ? 4 : 0);
    if (g == 1 || g == 2) *(f32x2*)dst = f32x2{Fj[0], Fj[1]};
    if (g == 1) *(f32x2*)(dst + 2) = f32x2{Fj[2], Fj[3]};
}

// what a step needs from LDS, fetched one step ahead: weight tiles (2 per joint), biases (2 per joint), quaternions
struct EncIn {
    f32x4 t[4];
    f32x4 b[4];
    f32x4 q[2];
};
template <int J, int NJOINTS>
__device__ __forceinline__ void enc_fetch(EncIn& in, const float* my_q, const float* encb, Ring& ring, int g) {
    in.t[0] = enc_tile<2 * J>(ring);
    in.t[1] = enc_tile<2 * J + 1>(ring);
    in.b[0] = *(const f32x4*)(encb + 32 * J + 4 * g);
    in.b[1] = *(const f32x4*)(encb + 32 * J + 16 + 4 * g);
    in.q[0] = *(const f32x4*)(my_q + 4 * J);
    if constexpr (NJOINTS == 2) {
        in.t[2] = enc_tile<2 * J + 2>(ring);
        in.t[3] = enc_tile<2 * J + 3>(ring);
        in.b[2] = *(const f32x4*)(encb + 32 * (J + 1) + 4 * g);
        in.b[3] = *(const f32x4*)(encb + 32 * (J + 1) + 16 + 4 * g);
        in.q[1] = *(const f32x4*)(my_q + 4 * (J + 1));
    }
}

// NT tiles (2 per joint) starting at stream tile T0 of the encoder section
template <int T0, int NT>
__device__ __forceinline__ void enc_tiles(f32x4 (&t)[4], Ring& ring) {
    if constexpr (NT >= 1) t[0] = enc_tile<T0>(ring);
    if constexpr (NT >= 2) t[1] = enc_tile<T0 + 1>(ring);
    if constexpr (NT >= 3) t[2] = enc_tile<T0 + 2>(ring);
    if constexpr (NT >= 4) t[3] = enc_tile<T0 + 3>(ring);
}

template <int J, bool SP>
__device__ __forceinline__ void enc_fwd_step(const float* my_q, float* my_f, const float* encb, const float (&inv)[4],
                                             f32x4 (&F)[NJ], float (&ebf)[NJ], const EncIn& in, Ring& ring,
                                             const ActP& ap, int g) {
    constexpr bool PAIR = enc_pairable(J);
    constexpr int JN = J + (PAIR ? 2 : 1);                     // first joint of the next step
    EncIn nx = in;
    if constexpr (JN < NJ) enc_fetch<JN, enc_pairable(JN) ? 2 : 1>(nx, my_q, encb, ring, g);   // next step's operands
    if constexpr (PAIR) {
        const f32x4 Xa = enc_input<J>(in.q[0], inv, F, g), Xb = enc_input<J + 1>(in.q[1], inv, F, g);
        f32x4 Ha = in.b[0], Hb = in.b[2];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            Ha = mfma4(in.t[0][s], Xa[s], Ha);
            Hb = mfma4(in.t[2][s], Xb[s], Hb);
        }
        const float hba = enc_act<SP>(Ha, ap, SP_SLOT_ENC + 2 * J);
        const float hbb = enc_act<SP>(Hb, ap, SP_SLOT_ENC + 2 * (J + 1));
        f32x4 Fa = in.b[1], Fb = in.b[3];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            Fa = mfma4(in.t[1][s], Ha[s], Fa);
            Fb = mfma4(in.t[3][s], Hb[s], Fb);
        }
        const float fba = enc_act<SP>(Fa, ap, SP_SLOT_ENC + 2 * J + 1);
        const float fbb = enc_act<SP>(Fb, ap, SP_SLOT_ENC + 2 * (J + 1) + 1);
        ebf[J] = fmaf(fba, 16.f, hba);          // 8 derivative bits of the joint, as an exact small float
        ebf[J + 1] = fmaf(fbb, 16.f, hbb);
        F[J] = Fa;
        F[J + 1] = Fb;
        enc_store_features<J>(my_f, Fa, g);
        enc_store_features<J + 1>(my_f, Fb, g);
    } else {
        const f32x4 X = enc_input<J>(in.q[0], inv, F, g);
        f32x4 H = in.b[0];
#pragma unroll
        for (int s = 0; s < 4; ++s) H = mfma4(in.t[0][s], X[s], H);
        const float hb = enc_act<SP>(H, ap, SP_SLOT_ENC + 2 * J);
        f32x4 Fj = in.b[1];
#pragma unroll
        for (int s = 0; s < 4; ++s) Fj = mfma4(in.t[1][s], H[s], Fj);
        const float fb = enc_act<SP>(Fj, ap, SP_SLOT_ENC + 2 * J + 1);
        ebf[J] = fmaf(fb, 16.f, hb);
        F[J] = Fj;
        enc_store_features<J>(my_f, Fj, g);
    }
    if constexpr (JN < NJ) enc_fwd_step<JN, SP>(my_q, my_f, encb, inv, F, ebf, nx, ring, ap, g);
}

template <bool SP>
__device__ __forceinline__ float encoder_forward(const float* my_q, float* my_f, const float* encb,
                                                 uint32_t (&eb)[6], Ring& ring, const ActP& ap, int g) {
    float ss[4], inv[4];
    const float poison = joint_axis_norms<SP>(my_q, ss);
#pragma unroll
    for (int c = 0; c < 4; ++c) inv[c] = 1.0f / fmaxf(sqrtf(ss[c]), 1e-12f);
    f32x4 F[NJ];
    float ebf[NJ];
    EncIn in;
    enc_fetch<0, enc_pairable(0) ? 2 : 1>(in, my_q, encb, ring, g);
    enc_fwd_step<0, SP>(my_q, my_f, encb, inv, F, ebf, in, ring, ap, g);
    if (g == 0) {
        my_f[126] = 0.f;
        my_f[127] = 0.f;
    }
    wave_lds_fence();      // features written by lane groups 1-2 are read by all lane groups (x0)
#pragma unroll
    for (int w = 0; w < 6; ++w) {
        uint32_t v = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (4 * w + k < NJ) v |= (uint32_t)ebf[4 * w + k] << (8 * k);
        eb[w] = v;
        asm volatile("" : "+v"(eb[w]));      // pin the packing here (see act_tiles)
    }
    return poison;
}

// Joint J, backward: GF[J] (rows 4..9 = d d / d feature, from the trunk plus the children) ->
//   gz2 = GF * act'(z2);  GH = W2^T gz2 (rows = hidden);  gz1 = GH * act'(z1);  GI = W1^T gz1
//   GI rows 0..3 = d d / d n_J (lane group 0 -> LDS), rows 4..9 = contribution to the parent's GF.
// Stream order is joint 20 .. 0; joints J and J-1 are paired when J is not a child of J-1 (then GF[J-1] is already
// complete: all its children have higher indices and were processed in earlier steps).
constexpr bool enc_bwd_pairable(int j) { return j >= 1 && PARENT[j] != j - 1; }

template <int J, bool SP>
__device__ __forceinline__ void enc_bwd_step(float* my_gn, f32x4 (&GF)[NJ], const uint32_t (&eb)[6], const f32x4 (&t)[4],
                                             Ring& ring, const ActP& ap, int g, const f32x4 (&D)[SP ? 2 * NJ : 1]) {
    constexpr auto dslot = [](int tile) { return SP ? tile : 0; };
    constexpr bool PAIR = enc_bwd_pairable(J);
    constexpr int JN = J - (PAIR ? 2 : 1);                     // first joint of the next step (may be < 0)
    f32x4 n[4] = {t[0], t[1], t[2], t[3]};
    if constexpr (JN >= 0) enc_tiles<2 * (NJ - 1 - JN), enc_bwd_pairable(JN) ? 4 : 2>(n, ring);
    if constexpr (PAIR) {
        constexpr int K = J - 1;
        const uint32_t ba = (eb[J / 4] >> (8 * (J % 4))) & 0xffu, bb = (eb[K / 4] >> (8 * (K % 4))) & 0xffu;
        f32x4 za = GF[J], zb = GF[K];
        enc_dact<SP>(za, ba >> 4, ap, D[dslot(2 * J + 1)]);
        enc_dact<SP>(zb, bb >> 4, ap, D[dslot(2 * K + 1)]);
        f32x4 Ha = f32x4{0.f, 0.f, 0.f, 0.f}, Hb = Ha;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            Ha = mfma4(t[0][s], za[s], Ha);
            Hb = mfma4(t[2][s], zb[s], Hb);
        }
        enc_dact<SP>(Ha, ba & 0xfu, ap, D[dslot(2 * J)]);
        enc_dact<SP>(Hb, bb & 0xfu, ap, D[dslot(2 * K)]);
        f32x4 Ia = f32x4{0.f, 0.f, 0.f, 0.f}, Ib = Ia;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            Ia = mfma4(t[1][s], Ha[s], Ia);
            Ib = mfma4(t[3][s], Hb[s], Ib);
        }
        if (g == 0) {
            *(f32x4*)(my_gn + 4 * J) = Ia;
            *(f32x4*)(my_gn + 4 * K) = Ib;
        }
        if constexpr (PARENT[J] >= 0) GF[PARENT[J]] = GF[PARENT[J]] + Ia;   // rows 0..3 of GF are never read back
        if constexpr (PARENT[K] >= 0) GF[PARENT[K]] = GF[PARENT[K]] + Ib;
    } else {
        const uint32_t byte = (eb[J / 4] >> (8 * (J % 4))) & 0xffu;
        f32x4 gz2 = GF[J];
        enc_dact<SP>(gz2, byte >> 4, ap, D[dslot(2 * J + 1)]);
        f32x4 GH = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s) GH = mfma4(t[0][s], gz2[s], GH);
        enc_dact<SP>(GH, byte & 0xfu, ap, D[dslot(2 * J)]);
        f32x4 GI = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s) GI = mfma4(t[1][s], GH[s], GI);
        if (g == 0) *(f32x4*)(my_gn + 4 * J) = GI;
        if constexpr (PARENT[J] >= 0) GF[PARENT[J]] = GF[PARENT[J]] + GI;
    }
    if constexpr (JN >= 0) enc_bwd_step<JN, SP>(my_gn, GF, eb, n, ring, ap, g, D);
}

// consumes d d / d feature from my_f, leaves d d / d n in my_gn
template <bool SP>
__device__ __forceinline__ void encoder_backward(float* my_f, float* my_gn, const uint32_t (&eb)[6], Ring& ring,
                                                 const ActP& ap, int g) {
    f32x4 GF[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        // rows 4..7 <- lane group 1, rows 8..9 <- lane group 2, everything else 0
        const float* src = my_f + FEAT * j + ((g == 2) ? 4 : 0);
        const float a = src[0], b = src[1];
        const float c = (g == 1) ? src[2] : 0.f, d = (g == 1) ? src[3] : 0.f;
        const bool live = (g == 1) || (g == 2);
        GF[j] = live ? f32x4{a, b, c, d} : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // softplus: all 42 parked derivatives of the encoder are fetched up front (at the point of use every joint
    // stalled for two global-memory round trips)
    f32x4 D[SP ? 2 * NJ : 1];
    if constexpr (SP) {
#pragma unroll
        for (int i = 0; i < 2 * NJ; ++i) D[i] = ap.sp.get(SP_SLOT_ENC + i);
    }
    f32x4 t[4];
    enc_tiles<0, enc_bwd_pairable(NJ - 1) ? 4 : 2>(t, ring);
    enc_bwd_step<NJ - 1, SP>(my_gn, GF, eb, t, ring, ap, g, D);
    wave_lds_fence();      // d d / d n written by lane group 0 is read by all lane groups
}

// ---- model.StrEnc.use = False (reference model/posendf.py:40-42,73-74): no encoder, DFNet's input is the normalised
// pose itself, x0[4 j + c] = n[j][c] (p.reshape(len(p), -1), net_modules.py:49), zero padded to 128 rows; lin0's packed
// tiles carry zero columns beyond 84.  The stream keeps its (now all-zero) encoder sections so that every phase starts on
// the same slot as with the encoder: they are skipped slot by slot with the ring's usual events.
__device__ __forceinline__ void ring_skip_encoder_section(Ring& ring) {
#pragma unroll 1
    for (int s = 0; s < ENC_TILES_PADDED / SLOT_TILES; ++s) {
        ring_boundary(ring);
        ring_midslot_sync(ring);
        ring_dma(ring);
    }
}
template <bool SP = false>
__device__ __forceinline__ float noenc_forward(const float* my_q, float* my_f, int g) {
    float ss[4], inv[4];
    const float poison = joint_axis_norms<SP>(my_q, ss);
#pragma unroll
    for (int c = 0; c < 4; ++c) inv[c] = 1.0f / fmaxf(sqrtf(ss[c]), 1e-12f);
    for (int j = g; j < 32; j += 4) {          // rows 84..127 of x0 are zero
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (j < NJ) {
            const f32x4 qj = *(const f32x4*)(my_q + 4 * j);
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = qj[c] * inv[c];
        }
        *(f32x4*)(my_f + 4 * j) = v;
    }
    wave_lds_fence();      // written by one lane group each, read by all (x0)
    return poison;
}

// q <- q - d * grad exactly as the reference evaluates it (experiments/sample_poses.py:74: the product is rounded to
// fp32, then subtracted).  hipcc's default -ffp-contract=fast would fuse the two into one fma (and __fmul_rn is a plain
// `*` in this toolchain); an opaque multiply keeps the two roundings, so that project() equals the reference's loop
// around forward + gradient bit for bit (tests/test_reference_callers.py).
__device__ __forceinline__ float project_update(float q, float d, float dq) {
    float prod;
    asm("v_mul_f32 %0, %1, %2" : "=v"(prod) : "v"(d), "v"(dq));
    return q - prod;
}

template <int NT>
__device__ __forceinline__ void load_bias(f32x4 (&acc)[NT], const float* bias, int g) {
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = *(const f32x4*)(bias + 16 * t + 4 * g);
}

// activation of an accumulator layer; relu family: derivative bits kept in registers (NT*4 bits);
// softplus: derivatives to the scratch slots [spslot, spslot + NT)
template <int NT>
__device__ __forceinline__ void dump_tiles(float* dbg, int off, const f32x4 (&x)[NT], int tid) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) dbg[(size_t)(off + 4 * t + r) * WG_THREADS + tid] = x[t][r];
    }
}

}  // namespace
