// Fused Pose-NDF distance / gradient / projection kernel for gfx950 (MI355X, CDNA4), exact fp32.
//
// One workgroup = 4 waves = 64 poses; one wave = 16 poses, one wave per SIMD, whole 512-register file.
// Per projection step (reference experiments/sample_poses.py:70-74) a wave runs, entirely on chip:
//   normalise over joints (model/posendf.py:71) -> 21 BoneMLPs along the kinematic tree
//   (model/network/net_modules.py:162-169; VALU, weights via scalar loads) -> trunk forward
//   (net_modules.py:46-72; v_mfma_f32_16x16x4_f32, activations resident in registers, weights streamed
//   global -> LDS by DMA) -> d -> trunk backward (same MFMA, transposed weight tiles) -> encoder backward
//   -> normalise backward -> q <- q - d * grad.
// See pndf_layout.h for the register/tile layout and DESIGN.md for the roofline.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pndf_layout.h"

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
constexpr int RING_SLOTS = 3;
constexpr int LDS_RING = 0;                                  // 3 slots of 16 KiB (DMA target, lowest addresses)
constexpr int LDS_BIAS = LDS_RING + RING_SLOTS * SLOT_BYTES; // BIAS_FLOATS floats (trunk + encoder biases)
constexpr int LDS_MASK = LDS_BIAS + BIAS_FLOATS * 4;         // u16 [MASK_CHUNKS][256]; aliased by GN after the trunk
constexpr int LDS_GN = LDS_MASK;                             // float [64][84]  d d / d n per pose
constexpr int LDS_Q = LDS_MASK + MASK_CHUNKS * WG_THREADS * 2;   // float [64][84]  the pose tile
constexpr int LDS_F = LDS_Q + WG_POSES * NQ * 4;             // float [64][FSTRIDE]  features, then d d / d feature
constexpr int LDS_TOTAL = LDS_F + WG_POSES * FSTRIDE * 4;
static_assert(LDS_BIAS % 16 == 0 && LDS_MASK % 16 == 0 && LDS_Q % 16 == 0 && LDS_F % 16 == 0, "16-byte LDS carve");
static_assert(WG_POSES * NQ * 4 <= MASK_CHUNKS * WG_THREADS * 2, "GN aliases the chunk-mask region");
static_assert(LDS_TOTAL <= 160 * 1024, "LDS budget");

enum { MODE_FORWARD = 0, MODE_FORWARD_GRAD = 1, MODE_PROJECT = 2 };

// debug dump stage offsets (floats per thread)
enum {
    DBG_FEAT = 0, DBG_X2 = 126, DBG_X4 = DBG_X2 + 128, DBG_X6 = DBG_X4 + 128, DBG_D = DBG_X6 + 16,
    DBG_G4 = DBG_D + 1, DBG_G2 = DBG_G4 + 128, DBG_G0 = DBG_G2 + 128, DBG_GN = DBG_G0 + 32,
    DBG_DQ = DBG_GN + 84, DBG_TOTAL = DBG_DQ + 84
};

// Weight ring: 3 slots of 16 tiles.  Slot i lives in buffer i % 3.  The one barrier per slot sits in the
// MIDDLE of the slot being consumed (tile 8): at that point every wave has left slot i-1, so its buffer
// can take the DMA of slot i+2, and the DMA of slot i+1 (issued one slot earlier) has landed for everybody.
// Crossing a slot boundary therefore needs no synchronisation and tile prefetch runs straight through.
struct Ring {
    const char* gstream;   // packed weight stream (global)
    char* smem;
    int nslots;            // slots per step (wrap point)
    int next;              // next slot to DMA
    int cur;               // buffer holding the slot being consumed
    int wave;              // wave id (uniform)
    int lane;
};

// LDS-DMA of one 16 KiB slot: each wave moves 4 tiles (global_load_lds_dwordx4 = 1 KiB per instruction,
// LDS destination = M0 + lane * 16).  Issued from inline asm on purpose: when hipcc sees the builtin it
// degrades every `s_waitcnt lgkmcnt(N)` of the tile prefetch to lgkmcnt(0), which serialises ds_read and
// MFMA.  Consequence (cdna_hip_programming.md 5.7): the compiler does not count these loads, so every
// consumer-side barrier is preceded by an explicit `s_waitcnt vmcnt(0)`.
// One 1-KiB piece (tile 4*wave + j of the slot).  M0 is written in the same statement that uses it
// (cdna_hip_programming.md 5.7: M0 is compiler-owned outside the statement).
__device__ __forceinline__ void ring_dma_piece(const char* src, uint32_t dst, int j) {
    uint32_t keep;
    if (j == 0)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
    else if (j == 1)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off offset:1024\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
    else if (j == 2)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off offset:2048\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off offset:3072\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
}

// source / destination of this wave's share of the next slot to fetch; advances r.next
__device__ __forceinline__ void ring_dma_begin(Ring& r, int buf, const char*& src, uint32_t& dst) {
    src = r.gstream + (size_t)r.next * SLOT_BYTES + r.wave * (4 * TILE_BYTES) + r.lane * 16;
    const uint32_t lds_base = (uint32_t)(size_t)(PNDF_LDS char*)(r.smem + LDS_RING);
    dst = lds_base + buf * SLOT_BYTES + r.wave * (4 * TILE_BYTES);
    r.next = (r.next + 1 == r.nslots) ? 0 : r.next + 1;
}

__device__ __forceinline__ void ring_dma(Ring& r, int buf) {
    const char* src;
    uint32_t dst;
    ring_dma_begin(r, buf, src, dst);
#pragma unroll
    for (int j = 0; j < 4; ++j) ring_dma_piece(src, dst, j);
}

__device__ __forceinline__ void ring_wait_dma() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__device__ __forceinline__ void ring_start(Ring& r) {
    r.next = 0;
    ring_dma(r, 0);
    ring_dma(r, 1);
    r.cur = 2;             // the first slot boundary makes it 0
}

// tile 0 of a slot: switch buffers (no barrier needed, see above)
__device__ __forceinline__ void ring_boundary(Ring& r) { r.cur = (r.cur == 2) ? 0 : r.cur + 1; }

// tile 8 of a slot: one barrier, then prefetch two slots ahead into the buffer of the previous slot
// A raw s_barrier, not __syncthreads(): the latter's fence adds `s_waitcnt lgkmcnt(0)`, i.e. it waits for the
// tile prefetch issued a few instructions earlier.  What the barrier has to order is already ordered: every
// wave's DMA share of the next slot has landed (its own vmcnt(0) above), and every read of the previous slot
// returned long ago (its data has been consumed by MFMAs issued before this point).
__device__ __forceinline__ void ring_midslot_sync(Ring& r) {
    ring_wait_dma();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ f32x4 ring_tile(const Ring& r, int t_in_slot) {
    return *(const f32x4*)(r.smem + LDS_RING + r.cur * SLOT_BYTES + t_in_slot * TILE_BYTES + r.lane * 16);
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
__device__ __forceinline__ float act_softplus(float z, float beta, float& deriv) {
    const float bz = z * beta;
    const float e = expf(fminf(bz, 20.0f));
    const bool lin = bz > 20.0f;
    deriv = lin ? 1.0f : e / (e + 1.0f);
    return lin ? z : log1pf(e) / beta;
}

// Activation parameters + where derivatives are parked between the forward and the backward pass.
//   relu family : sign bits (chunk layers: one u16 per lane per chunk in LDS; accumulator layers: registers)
//   softplus    : fp32 derivatives in a per-workgroup global scratch, one float4 per lane per tile ("slot")
struct ActP {
    float slope;        // relu family
    float beta;         // softplus
    f32x4* sp;          // softplus: this thread's column of the scratch ([slot][256] float4), else null
};
constexpr int SP_SLOT_CHUNK[3] = {0, 16, 80};      // chunk layers x1 (8x2), x3 (32x2), x5 (4x4)
constexpr int SP_SLOT_X2 = 96, SP_SLOT_X4 = 128, SP_SLOT_X6 = 160, SP_SLOT_ENC = 164;   // encoder: 2 tiles per joint
constexpr int SP_SLOTS = SP_SLOT_ENC + 2 * NJ;
constexpr int SP_WG_FLOATS = SP_SLOTS * WG_THREADS * 4;

// One fused layer pair.  xin: KA input tiles (B operands); acc: NB output tiles (accumulators).
// Forward: chunk accumulators start from the A-layer bias, get the activation, and their sign bits are
// parked in LDS; backward: chunk accumulators start at 0 and are multiplied by the parked derivative.
// Weight tiles are consumed in groups of GT = 2 CT tiles; the group after the current one is read from
// LDS before the current group's MFMAs are issued (hipcc does not software-pipeline this by itself).
constexpr int TIMING_REGIONS = 12;
constexpr int TIMING_GROUPS = 32;     // per-group stamps inside the (lin2,lin3) phase
// s_memtime stamps at region boundaries (TIMING instantiation only): accumulates shader cycles per region
struct RegionClock {
    unsigned long long acc[TIMING_REGIONS];
    unsigned long long grp[TIMING_GROUPS];
    unsigned long long last;
};
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

template <int KA, int CT, int NC, int NB, bool BWD, bool SP, bool GTIME = false>
struct PhaseBody {
    static constexpr int GT = 2 * CT;                 // tiles per group
    static constexpr int NGA = KA / 2;                // part-A groups (two k-tiles each)
    static constexpr int NGB = NB / 2;                // part-B groups (two output tiles each)
    static constexpr int NG = NGA + NGB;
    static constexpr int CHUNK_TILES = CT * KA + NB * CT;
    static_assert(CHUNK_TILES % SLOT_TILES == 0, "chunk body must be whole slots");
    static_assert(KA % 2 == 0 && NB % 2 == 0, "tiles are consumed in pairs");

    // groups GI .. NG-1 of one chunk; `cur` holds group GI's tiles on entry and the next chunk's group 0
    // (or garbage after the very last group) on exit.
    template <int GI>
    static __device__ __forceinline__ void groups(const f32x4 (&xin)[KA], f32x4 (&acc)[NB], f32x4 (&ch)[CT],
                                                  f32x4 (&cur)[GT], Ring& ring, uint16_t* mask, const ActP& ap,
                                                  int spslot, int c, RegionClock* rc) {
        if constexpr (GI < NG) {
            f32x4 nxt[GT];
            constexpr int TNEXT = ((GI + 1) * GT) % CHUNK_TILES;
            constexpr bool MID = group_has_mid<GT, TNEXT>();
            const bool loaded = (GI + 1 < NG || c + 1 < NC);
            if (loaded) load_group<GT, TNEXT>(nxt, ring);
            // the slot fetch that follows a mid-slot barrier: four 1-KiB DMA instructions, each issued behind
            // an MFMA of this group (a VMEM issue blocks the wave ~16+ cycles; back to back they starve the
            // MFMA pipe, behind an MFMA they are free)
            const char* dsrc = nullptr;
            uint32_t ddst = 0;
            if (MID && loaded) ring_dma_begin(ring, (ring.cur == 0) ? 2 : ring.cur - 1, dsrc, ddst);
            int piece = 0;
            __builtin_amdgcn_sched_barrier(0);   // keep the prefetch ABOVE this group's MFMAs
            if constexpr (GI < NGA) {
                // ---- part A: two k-tiles of the chunk rows; tile order (kt, ci)
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2) {
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
#pragma unroll
                        for (int ci = 0; ci < CT; ++ci)
                            ch[ci] = mfma4(cur[k2 * CT + ci][s], xin[2 * GI + k2][s], ch[ci]);
                        if (MID && s < 2 && loaded) {      // 2 k-tiles x 2 = 4 pieces
                            __builtin_amdgcn_sched_barrier(0);
                            ring_dma_piece(dsrc, ddst, piece++);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
                if constexpr (GI == NGA - 1) {
                    // ---- chunk epilogue
                    if constexpr (SP) {
#pragma unroll
                        for (int ci = 0; ci < CT; ++ci) {
                            f32x4* slot = ap.sp + (size_t)(spslot + c * CT + ci) * WG_THREADS;
                            if (!BWD) {
                                f32x4 dv;
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    float dr;
                                    ch[ci][r] = act_softplus(ch[ci][r], ap.beta, dr);
                                    dv[r] = dr;
                                }
                                *slot = dv;
                            } else {
                                ch[ci] = ch[ci] * *slot;
                            }
                        }
                    } else if (!BWD) {
                        // sign bits are summed as st * 2^k in ONE fp32 chain (exact below 2^24): a float chain cannot
                        // be reassociated or deferred piecewise, so every step value dies at once
                        float bitsum = 0.f;
#pragma unroll
                        for (int ci = 0; ci < CT; ++ci) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const float st = step01(ch[ci][r]);
                                ch[ci][r] = ch[ci][r] * relu_factor(st, ap.slope);
                                bitsum = fmaf(st, (float)(1u << (ci * 4 + r)), bitsum);
                            }
                        }
                        mask[c * WG_THREADS] = (uint16_t)(uint32_t)bitsum;
                    } else {
                        const uint32_t bits = mask[c * WG_THREADS];
#pragma unroll
                        for (int ci = 0; ci < CT; ++ci) {
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                ch[ci][r] = ch[ci][r] * relu_factor((float)((bits >> (ci * 4 + r)) & 1u), ap.slope);
                        }
                    }
                }
            } else {
                // ---- part B: two output tiles get this chunk's contribution; tile order (ci, h)
                constexpr int nbp = GI - NGA;
#pragma unroll
                for (int ci = 0; ci < CT; ++ci) {
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
#pragma unroll
                        for (int h = 0; h < 2; ++h)
                            acc[2 * nbp + h] = mfma4(cur[ci * 2 + h][s], ch[ci][s], acc[2 * nbp + h]);
                        if (MID && ci < 2 && s < 2 && loaded) {   // 4 pieces
                            __builtin_amdgcn_sched_barrier(0);
                            ring_dma_piece(dsrc, ddst, piece++);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < GT; ++i) cur[i] = nxt[i];
            if constexpr (GTIME) {
                const unsigned long long now = __builtin_amdgcn_s_memtime();
                rc->grp[GI] += now - rc->last;
                rc->last = now;
            }
            groups<GI + 1>(xin, acc, ch, cur, ring, mask, ap, spslot, c, rc);
        }
    }
};

template <int KA, int CT, int NC, int NB, bool BWD, bool SP, bool GTIME = false>
__device__ __forceinline__ void run_phase(const f32x4 (&xin)[KA], f32x4 (&acc)[NB], Ring& ring,
                                          const float* biasA, uint16_t* mask, const ActP& ap, int spslot, int g,
                                          RegionClock* rc = nullptr) {
    using Body = PhaseBody<KA, CT, NC, NB, BWD, SP, GTIME>;
    f32x4 cur[Body::GT];
    load_group<Body::GT, 0>(cur, ring);
    for (int c = 0; c < NC; ++c) {
        f32x4 ch[CT];
#pragma unroll
        for (int ci = 0; ci < CT; ++ci) {
            if (!BWD) ch[ci] = *(const f32x4*)(biasA + 16 * (c * CT + ci) + 4 * g);
            else ch[ci] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if constexpr (GTIME) rc->last = __builtin_amdgcn_s_memtime();
        Body::template groups<0>(xin, acc, ch, cur, ring, mask, ap, spslot, c, rc);
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
        ring_dma(ring, (ring.cur == 0) ? 2 : ring.cur - 1);
    }
    return ring_tile(ring, t);
}

// per-component denominators of F.normalize(pose, dim=1): max(||q[:, c]||_2 over joints, eps)
__device__ __forceinline__ void joint_axis_norms(const float* my_q, float (&ss)[4]) {
    ss[0] = ss[1] = ss[2] = ss[3] = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const f32x4 v = *(const f32x4*)(my_q + 4 * j);
#pragma unroll
        for (int c = 0; c < 4; ++c) ss[c] = fmaf(v[c], v[c], ss[c]);
    }
}

// activation of one encoder tile; returns 4 derivative bits (relu family) or stores the derivative (softplus)
template <bool SP>
__device__ __forceinline__ float enc_act(f32x4& z, const ActP& ap, int spslot) {
    float bitsum = 0.f;
    if constexpr (SP) {
        f32x4 dv;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float dr;
            z[r] = act_softplus(z[r], ap.beta, dr);
            dv[r] = dr;
        }
        ap.sp[(size_t)spslot * WG_THREADS] = dv;
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
__device__ __forceinline__ void enc_dact(f32x4& gz, uint32_t bits4, const ActP& ap, int spslot) {
    if constexpr (SP) {
        gz = gz * ap.sp[(size_t)spslot * WG_THREADS];
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) gz[r] = gz[r] * relu_factor((float)((bits4 >> r) & 1u), ap.slope);
    }
}

template <int J, bool SP>
__device__ __forceinline__ void enc_fwd_joint(const float* my_q, float* my_f, const float* encb,
                                              const float (&denom)[4], f32x4 (&F)[NJ], float (&ebf)[NJ],
                                              f32x4 t1, f32x4 t2, Ring& ring, const ActP& ap, int g) {
    f32x4 n1 = t1, n2 = t2;
    if constexpr (J + 1 < NJ) {            // prefetch the next joint's two tiles
        n1 = enc_tile<2 * (J + 1)>(ring);
        n2 = enc_tile<2 * (J + 1) + 1>(ring);
    }
    const f32x4 qj = *(const f32x4*)(my_q + 4 * J);
    f32x4 X;
#pragma unroll
    for (int c = 0; c < 4; ++c) X[c] = qj[c] / denom[c];       // posendf.py:71
    if constexpr (PARENT[J] >= 0) {
        X = (g == 0) ? X : F[PARENT[J]];                       // cat(quat, parent feature), net_modules.py:167
    } else {
        X = (g == 0) ? X : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    f32x4 H = *(const f32x4*)(encb + 32 * J + 4 * g);
#pragma unroll
    for (int s = 0; s < 4; ++s) H = mfma4(t1[s], X[s], H);
    const float hb = enc_act<SP>(H, ap, SP_SLOT_ENC + 2 * J);
    f32x4 Fj = *(const f32x4*)(encb + 32 * J + 16 + 4 * g);
#pragma unroll
    for (int s = 0; s < 4; ++s) Fj = mfma4(t2[s], H[s], Fj);
    const float fb = enc_act<SP>(Fj, ap, SP_SLOT_ENC + 2 * J + 1);
    ebf[J] = fmaf(fb, 16.f, hb);            // 8 derivative bits of this joint, as an exact small float
    F[J] = Fj;
    // features of the joint -> per-pose buffer (rows 4..7 live in lane group 1, rows 8..9 in lane group 2)
    if (g == 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) my_f[FEAT * J + r] = Fj[r];
    } else if (g == 2) {
        my_f[FEAT * J + 4] = Fj[0];
        my_f[FEAT * J + 5] = Fj[1];
    }
    if constexpr (J + 1 < NJ) enc_fwd_joint<J + 1, SP>(my_q, my_f, encb, denom, F, ebf, n1, n2, ring, ap, g);
}

template <bool SP>
__device__ __forceinline__ void encoder_forward(const float* my_q, float* my_f, const float* encb,
                                                uint32_t (&eb)[6], Ring& ring, const ActP& ap, int g) {
    float ss[4], denom[4];
    joint_axis_norms(my_q, ss);
#pragma unroll
    for (int c = 0; c < 4; ++c) denom[c] = fmaxf(sqrtf(ss[c]), 1e-12f);
    f32x4 F[NJ];
    float ebf[NJ];
    const f32x4 t1 = enc_tile<0>(ring);
    const f32x4 t2 = enc_tile<1>(ring);
    enc_fwd_joint<0, SP>(my_q, my_f, encb, denom, F, ebf, t1, t2, ring, ap, g);
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
}

// Joint J, backward: GF[J] (rows 4..9 = d d / d feature, from the trunk plus the children) ->
//   gz2 = GF * act'(z2);  GH = W2^T gz2 (rows = hidden);  gz1 = GH * act'(z1);  GI = W1^T gz1
//   GI rows 0..3 = d d / d n_J (lane group 0 -> LDS), rows 4..9 = contribution to the parent's GF.
template <int J, bool SP>
__device__ __forceinline__ void enc_bwd_joint(float* my_gn, f32x4 (&GF)[NJ], const uint32_t (&eb)[6],
                                              f32x4 t1, f32x4 t2, Ring& ring, const ActP& ap, int g) {
    f32x4 n1 = t1, n2 = t2;
    if constexpr (J > 0) {
        n1 = enc_tile<2 * (NJ - J)>(ring);          // tiles of joint J-1: stream order is joint 20 .. 0
        n2 = enc_tile<2 * (NJ - J) + 1>(ring);
    }
    const uint32_t byte = (eb[J / 4] >> (8 * (J % 4))) & 0xffu;
    f32x4 gz2 = GF[J];
    enc_dact<SP>(gz2, byte >> 4, ap, SP_SLOT_ENC + 2 * J + 1);
    f32x4 GH = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 4; ++s) GH = mfma4(t1[s], gz2[s], GH);
    enc_dact<SP>(GH, byte & 0xfu, ap, SP_SLOT_ENC + 2 * J);
    f32x4 GI = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 4; ++s) GI = mfma4(t2[s], GH[s], GI);
    if (g == 0) *(f32x4*)(my_gn + 4 * J) = GI;
    if constexpr (PARENT[J] >= 0) GF[PARENT[J]] = GF[PARENT[J]] + GI;   // rows 0..3 of GF are never read back
    if constexpr (J > 0) enc_bwd_joint<J - 1, SP>(my_gn, GF, eb, n1, n2, ring, ap, g);
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
    const f32x4 t1 = enc_tile<0>(ring);
    const f32x4 t2 = enc_tile<1>(ring);
    enc_bwd_joint<NJ - 1, SP>(my_gn, GF, eb, t1, t2, ring, ap, g);
    wave_lds_fence();      // d d / d n written by lane group 0 is read by all lane groups
}

template <int NT>
__device__ __forceinline__ void load_bias(f32x4 (&acc)[NT], const float* bias, int g) {
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = *(const f32x4*)(bias + 16 * t + 4 * g);
}

// activation of an accumulator layer; relu family: derivative bits kept in registers (NT*4 bits);
// softplus: derivatives to the scratch slots [spslot, spslot + NT)
template <int NT, bool SP>
__device__ __forceinline__ void act_tiles(f32x4 (&x)[NT], uint32_t (&m)[(NT * 4 + 31) / 32], const ActP& ap, int spslot) {
    if constexpr (SP) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            f32x4 dv;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float dr;
                x[t][r] = act_softplus(x[t][r], ap.beta, dr);
                dv[r] = dr;
            }
            ap.sp[(size_t)(spslot + t) * WG_THREADS] = dv;
        }
    } else {
        constexpr int NW = (NT * 4 + 31) / 32;
        float lo[NW], hi[NW];      // 16 sign bits each, summed as st * 2^k in fp32 chains (see run_phase)
#pragma unroll
        for (int w = 0; w < NW; ++w) lo[w] = hi[w] = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float st = step01(x[t][r]);
                x[t][r] = x[t][r] * relu_factor(st, ap.slope);
                const int w = (t * 4 + r) / 32, b = (t * 4 + r) % 32;
                if (b < 16) lo[w] = fmaf(st, (float)(1u << b), lo[w]);
                else hi[w] = fmaf(st, (float)(1u << (b - 16)), hi[w]);
            }
        }
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            m[w] = (uint32_t)lo[w] | ((uint32_t)hi[w] << 16);
            // pin the packing HERE: LLVM otherwise sinks the whole chain to its first use in the backward pass
            // and keeps (spills) all 128 step values until then (each reload = a memory round trip)
            asm volatile("" : "+v"(m[w]));
        }
    }
}

template <int NT, bool SP>
__device__ __forceinline__ void dact_tiles(f32x4 (&gx)[NT], const uint32_t (&m)[(NT * 4 + 31) / 32], const ActP& ap, int spslot) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if constexpr (SP) {
            gx[t] = gx[t] * ap.sp[(size_t)(spslot + t) * WG_THREADS];
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                gx[t][r] = gx[t][r] * relu_factor((float)((m[(t * 4 + r) / 32] >> ((t * 4 + r) % 32)) & 1u), ap.slope);
        }
    }
}

template <int NT>
__device__ __forceinline__ void dump_tiles(float* dbg, int off, const f32x4 (&x)[NT], int tid) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) dbg[(size_t)(off + 4 * t + r) * WG_THREADS + tid] = x[t][r];
    }
}

}  // namespace

struct PndfKernelArgs {
    const float* q_in;      // [B,84]
    float* q_out;           // [B,84]  projected poses (PROJECT) or dd/dq * grad_out (FORWARD_GRAD)
    float* d_out;           // [B]
    const float* grad_out;  // [B] or null (FORWARD_GRAD only)
    const char* stream;     // packed trunk weights, STEP_TILES KiB
    const float* bias;      // BIAS_FLOATS
    float* dbg;             // null, or DBG_TOTAL*256 floats written by workgroup 0 (first step)
    long long B;
    int steps;
    int mode;
    float slope;            // 0 = relu, 0.01 = lrelu
    float beta;             // softplus beta
    float* scratch;         // softplus: gridDim.x * SP_WG_FLOATS floats of derivative scratch, else null
    int dbg_nslots;         // 0, or (timing experiments only, wrong results) wrap the weight stream after n slots
};

template <bool DBG, bool SP, bool TIMING = false>
__device__ __forceinline__ void pndf_fused_body(const PndfKernelArgs& args) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4;          // lane group = k_local of the MFMA
    const int p = lane & 15;          // pose within the wave
    const int wp = wave * 16 + p;     // pose within the workgroup
    ActP ap;
    ap.slope = args.slope;
    ap.beta = args.beta;
    float* const wg_scratch = SP ? args.scratch + (size_t)blockIdx.x * SP_WG_FLOATS : nullptr;
    ap.sp = SP ? (f32x4*)wg_scratch + tid : nullptr;
    const long long pose0 = (long long)blockIdx.x * WG_POSES;
    float* const lds_bias = (float*)(smem + LDS_BIAS);
    uint16_t* const lds_mask = (uint16_t*)(smem + LDS_MASK) + tid;
    float* const lds_q = (float*)(smem + LDS_Q);
    float* const my_q = lds_q + wp * NQ;
    float* const my_f = (float*)(smem + LDS_F) + wp * FSTRIDE;
    float* const my_gn = (float*)(smem + LDS_GN) + wp * NQ;
    float* const dbg = (DBG && blockIdx.x == 0) ? args.dbg : nullptr;

    Ring ring;
    ring.gstream = args.stream;
    ring.smem = smem;
    ring.nslots = (args.mode == MODE_FORWARD) ? FWD_SLOTS : STEP_SLOTS;
    if (args.dbg_nslots > 0) ring.nslots = args.dbg_nslots;
    ring.wave = wave;
    ring.lane = lane;
    ring_start(ring);   // slots 0 and 1 in flight; the __syncthreads() below makes them visible

    // ---- stage biases and this workgroup's poses in LDS (coalesced)
    for (int i = tid; i < BIAS_FLOATS / 4; i += WG_THREADS)
        ((f32x4*)lds_bias)[i] = ((const f32x4*)args.bias)[i];
    {
        long long nvalid = args.B - pose0;
        if (nvalid > WG_POSES) nvalid = WG_POSES;
        const f32x4* src = (const f32x4*)(args.q_in + pose0 * NQ);
        const int nvec = (int)nvalid * (NQ / 4);
        for (int i = tid; i < WG_POSES * (NQ / 4); i += WG_THREADS) {
            // poses past the end of the batch replicate the last valid pose (never written back)
            const int src_i = i < nvec ? i : (nvec - (NQ / 4) + (i % (NQ / 4)));
            ((f32x4*)lds_q)[i] = src[src_i];
        }
    }
    ring_wait_dma();
    __syncthreads();

    const int nsteps = (args.mode == MODE_PROJECT) ? args.steps : 1;
    float dval = 0.f;
    RegionClock rc;
    if constexpr (TIMING) {
#pragma unroll
        for (int i = 0; i < TIMING_REGIONS; ++i) rc.acc[i] = 0;
#pragma unroll
        for (int i = 0; i < TIMING_GROUPS; ++i) rc.grp[i] = 0;
        rc.last = __builtin_amdgcn_s_memtime();
    }
    for (int step = 0; step < nsteps; ++step) {
        uint32_t eb[6];
        uint32_t m2[4], m4[4], m6[1];
        f32x4 x6[4];
        f32x4 x4[32];
        {
            f32x4 x2[32];
            {
                // ---------------- normalise + encoder forward (posendf.py:71, net_modules.py:162-169)
                encoder_forward<SP>(my_q, my_f, lds_bias + ENCB_OFF, eb, ring, ap, g);
                if (DBG && dbg && step == 0) {
                    for (int i = 0; i < NFEAT; ++i) dbg[(size_t)(DBG_FEAT + i) * WG_THREADS + tid] = my_f[i];
                }
                // B operands of lin0: lane (g, p) holds features 16 kt + 4 g + s of pose p
                f32x4 x0[8];
#pragma unroll
                for (int kt = 0; kt < 8; ++kt) x0[kt] = *(const f32x4*)(my_f + 16 * kt + 4 * g);
                tick<TIMING>(rc, 0);
                // ---------------- trunk forward
                load_bias<32>(x2, lds_bias + BIAS_OFF[1], g);
                run_phase<8, 2, 8, 32, false, SP>(x0, x2, ring, lds_bias + BIAS_OFF[0],
                                                  lds_mask + MASK_BASE[0] * WG_THREADS, ap, SP_SLOT_CHUNK[0], g);
            }
            tick<TIMING>(rc, 1);
            act_tiles<32, SP>(x2, m2, ap, SP_SLOT_X2);
            tick<TIMING>(rc, 2);
            if (DBG && dbg && step == 0) dump_tiles<32>(dbg, DBG_X2, x2, tid);
            load_bias<32>(x4, lds_bias + BIAS_OFF[3], g);
            run_phase<32, 2, 32, 32, false, SP, TIMING>(x2, x4, ring, lds_bias + BIAS_OFF[2],
                                                        lds_mask + MASK_BASE[1] * WG_THREADS, ap, SP_SLOT_CHUNK[1], g, &rc);
        }
        tick<TIMING>(rc, 3);
        act_tiles<32, SP>(x4, m4, ap, SP_SLOT_X4);
        tick<TIMING>(rc, 4);
        if (DBG && dbg && step == 0) dump_tiles<32>(dbg, DBG_X4, x4, tid);
        load_bias<4>(x6, lds_bias + BIAS_OFF[5], g);
        run_phase<32, 4, 4, 4, false, SP>(x4, x6, ring, lds_bias + BIAS_OFF[4],
                                          lds_mask + MASK_BASE[2] * WG_THREADS, ap, SP_SLOT_CHUNK[2], g);
        tick<TIMING>(rc, 5);
        act_tiles<4, SP>(x6, m6, ap, SP_SLOT_X6);
        if (DBG && dbg && step == 0) dump_tiles<4>(dbg, DBG_X6, x6, tid);

        // ---------------- lin6 (64 -> 1) + output ReLU  (net_modules.py:64-69)
        f32x4 w6[4];
        load_bias<4>(w6, lds_bias + W6_OFF, g);
        float part = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) part = fmaf(w6[t][r], x6[t][r], part);
        }
        part += __shfl_xor(part, 16);
        part += __shfl_xor(part, 32);
        const float z7 = part + lds_bias[BIAS_OFF[6]];
        float gz7;
        if constexpr (SP) {
            dval = act_softplus(z7, ap.beta, gz7);      // output Softplus, net_modules.py:39-41,69
        } else {
            dval = fmaxf(z7, 0.f);                       // output ReLU for relu AND lrelu, net_modules.py:30-37
            gz7 = (z7 > 0.f) ? 1.f : 0.f;
        }
        if (DBG && dbg && step == 0) dbg[(size_t)DBG_D * WG_THREADS + tid] = dval;
        if (args.mode == MODE_FORWARD) break;

        // ---------------- trunk backward: d d / d x, masks from the forward pass
        if (args.mode == MODE_FORWARD_GRAD && args.grad_out) {
            long long pidx = pose0 + wp;
            if (pidx >= args.B) pidx = args.B - 1;
            gz7 *= args.grad_out[pidx];
        }
        f32x4 g6[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) g6[t] = w6[t] * gz7;
        dact_tiles<4, SP>(g6, m6, ap, SP_SLOT_X6);
        tick<TIMING>(rc, 6);
        {
            f32x4 g0[8];
            {
                f32x4 g2[32];
                {
                    f32x4 g4[32];
#pragma unroll
                    for (int t = 0; t < 32; ++t) g4[t] = f32x4{0.f, 0.f, 0.f, 0.f};
                    run_phase<4, 4, 4, 32, true, SP>(g6, g4, ring, nullptr, lds_mask + MASK_BASE[2] * WG_THREADS, ap,
                                                     SP_SLOT_CHUNK[2], g);
                    dact_tiles<32, SP>(g4, m4, ap, SP_SLOT_X4);
                    tick<TIMING>(rc, 7);
                    if (DBG && dbg && step == 0) dump_tiles<32>(dbg, DBG_G4, g4, tid);
#pragma unroll
                    for (int t = 0; t < 32; ++t) g2[t] = f32x4{0.f, 0.f, 0.f, 0.f};
                    run_phase<32, 2, 32, 32, true, SP>(g4, g2, ring, nullptr, lds_mask + MASK_BASE[1] * WG_THREADS, ap,
                                                       SP_SLOT_CHUNK[1], g);
                }
                dact_tiles<32, SP>(g2, m2, ap, SP_SLOT_X2);
                tick<TIMING>(rc, 8);
                if (DBG && dbg && step == 0) dump_tiles<32>(dbg, DBG_G2, g2, tid);
#pragma unroll
                for (int t = 0; t < 8; ++t) g0[t] = f32x4{0.f, 0.f, 0.f, 0.f};
                run_phase<32, 2, 8, 8, true, SP>(g2, g0, ring, nullptr, lds_mask + MASK_BASE[0] * WG_THREADS, ap,
                                                 SP_SLOT_CHUNK[0], g);
            }
            if (DBG && dbg && step == 0) dump_tiles<8>(dbg, DBG_G0, g0, tid);
            // d d / d feature back to the per-pose buffer: lane (g, p) owns features 16 t + 4 g + r
#pragma unroll
            for (int t = 0; t < 8; ++t) *(f32x4*)(my_f + 16 * t + 4 * g) = g0[t];
        }
        // The chunk masks (aliased by GN) are dead for every wave only after all waves have left the last
        // backward phase; GN/F of a pose are touched by its own wave only.
        __syncthreads();

        tick<TIMING>(rc, 9);
        // ---------------- encoder backward + normalise backward + update
        encoder_backward<SP>(my_f, my_gn, eb, ring, ap, g);
        if (DBG && dbg && step == 0) {
            for (int i = 0; i < NQ; ++i) dbg[(size_t)(DBG_GN + i) * WG_THREADS + tid] = my_gn[i];
        }
        tick<TIMING>(rc, 10);
        {
            // backward of x / clamp_min(||x||_joints, eps)  (model/posendf.py:71)
            float ss[4], dot[4], denom[4], kk[4];
            ss[0] = ss[1] = ss[2] = ss[3] = 0.f;
            dot[0] = dot[1] = dot[2] = dot[3] = 0.f;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const f32x4 qv = *(const f32x4*)(my_q + 4 * j);
                const f32x4 gv = *(const f32x4*)(my_gn + 4 * j);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    ss[c] = fmaf(qv[c], qv[c], ss[c]);
                    dot[c] = fmaf(gv[c], qv[c], dot[c]);
                }
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float norm = sqrtf(ss[c]);
                denom[c] = fmaxf(norm, 1e-12f);
                kk[c] = (norm > 1e-12f) ? dot[c] / (denom[c] * denom[c] * norm) : 0.f;
            }
            // lane group g handles joints g, g+4, ...; results replace the pose tile in LDS
            for (int j = g; j < NJ; j += 4) {
                const f32x4 qv = *(const f32x4*)(my_q + 4 * j);
                const f32x4 gv = *(const f32x4*)(my_gn + 4 * j);
                f32x4 o;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float dq = gv[c] / denom[c] - qv[c] * kk[c];
                    if (DBG && dbg && step == 0) dbg[(size_t)(DBG_DQ + 4 * j + c) * WG_THREADS + tid] = dq;
                    // q <- q - d * grad  (experiments/sample_poses.py:74: product rounded, then subtracted)
                    o[c] = (args.mode == MODE_PROJECT) ? __fsub_rn(qv[c], __fmul_rn(dval, dq)) : dq;
                }
                *(f32x4*)(my_q + 4 * j) = o;
            }
        }
        // GN (aliasing the chunk masks) must be consumed by every wave before the next step's masks land
        __syncthreads();
        tick<TIMING>(rc, 11);
    }
    if constexpr (TIMING) {
        if (lane == 0 && args.dbg) {
            unsigned long long* out = (unsigned long long*)args.dbg + ((size_t)blockIdx.x * 4 + wave) * (TIMING_REGIONS + TIMING_GROUPS);
#pragma unroll
            for (int i = 0; i < TIMING_REGIONS; ++i) out[i] = rc.acc[i];
#pragma unroll
            for (int i = 0; i < TIMING_GROUPS; ++i) out[TIMING_REGIONS + i] = rc.grp[i];
        }
    }

    // ---------------- write back (coalesced)
    {
        long long pidx = pose0 + wp;
        if (g == 0 && pidx < args.B && args.d_out) args.d_out[pidx] = dval;
    }
    if (args.mode != MODE_FORWARD) {
        __syncthreads();
        long long nvalid = args.B - pose0;
        if (nvalid > WG_POSES) nvalid = WG_POSES;
        f32x4* dst = (f32x4*)(args.q_out + pose0 * NQ);
        const int nvec = (int)nvalid * (NQ / 4);
        for (int i = tid; i < nvec; i += WG_THREADS) dst[i] = ((const f32x4*)lds_q)[i];
    }
    // drain the DMA prefetch that is still in flight before the workgroup's LDS is released
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

extern "C" __global__ void __launch_bounds__(WG_THREADS, 1)
pndf_fused_relu_kernel(PndfKernelArgs args) {
    pndf_fused_body<false, false>(args);
}

// Softplus(beta) variant (the reference's published checkpoints): fp32 derivatives go through `scratch`.
extern "C" __global__ void __launch_bounds__(WG_THREADS, 1)
pndf_fused_softplus_kernel(PndfKernelArgs args) {
    pndf_fused_body<false, true>(args);
}

// relu-family kernel with per-stage register dumps from workgroup 0 (tests / bring-up only).
extern "C" __global__ void __launch_bounds__(WG_THREADS, 1)
pndf_fused_relu_kernel_dbg(PndfKernelArgs args) {
    pndf_fused_body<true, false>(args);
}

// relu-family kernel with s_memtime region stamps (performance analysis only)
extern "C" __global__ void __launch_bounds__(WG_THREADS, 1)
pndf_fused_relu_kernel_timing(PndfKernelArgs args) {
    pndf_fused_body<false, false, true>(args);
}

extern "C" int pndf_kernel_timing_regions() { return TIMING_REGIONS + TIMING_GROUPS; }
extern "C" int pndf_kernel_lds_bytes() { return LDS_TOTAL; }
extern "C" int pndf_kernel_dbg_floats() { return DBG_TOTAL * WG_THREADS; }
extern "C" long long pndf_kernel_softplus_scratch_floats_per_wg() { return SP_WG_FLOATS; }
