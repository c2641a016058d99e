// Runtime-planned Pose-NDF kernel for gfx950: any DFNet the reference can build.
//
// reference model/network/net_modules.py:14-28 takes the hidden widths of DFNet as a free list (`dims`), and the README points
// users at checkpoints of configs other than configs/amass.yaml.  The fused kernels of pndf_kernel*.hip are compile-time plans of
// that one architecture (pair-fused phases, register-resident activations); everything else -- 2 .. 8 linear layers, hidden widths
// 1 .. 1024 -- runs here, with the same ownership (a workgroup = 4 waves = 64 poses for the whole call, one persistent launch for
// all projection steps), the same MFMA encoder, normalisation and update code (pndf_device.h), and the trunk LAYER BY LAYER from a
// plan that travels in the kernel arguments:
//   * arithmetic: precision fp32 -- exact fp32 (v_mfma_f32_16x16x4_f32, fp32 accumulate from the bias), the same operation order as
//     pndf_fused_relu_kernel; precision f16x3 / f16 -- the same plan on split-precision fp16 MFMAs (pndf_generic_split_*_kernel,
//     "the same trunk on split-precision fp16 MFMAs" below), unless a layer cannot be scaled into the fp16 range;
//   * transposed like the fused kernels: D tile = 16 rows x the wave's 16 poses, and a D tile is the B operand of the next
//     layer as it stands (pndf_layout.h), so activations are never transposed;
//   * what does not fit the register file at run-time widths lives in a per-workgroup global scratch that only ITS OWN lane ever
//     touches (every lane stores and re-loads exactly its 16 bytes of a tile: private memory with a coalesced layout -- no
//     barrier, no fence): activations ping / pong (64 tiles each) and the activation derivative of every hidden unit (fp32
//     factor: 1 | slope, or softplus' e / (1 + e)) for the backward pass;
//   * the accumulators of up to 32 output tiles (512 rows: 128 registers) stay resident, so an operand tile is loaded once per
//     layer (twice for layers wider than 512) and prefetched a whole k step (up to 128 MFMAs) ahead -- it comes back from L2 /
//     Infinity Cache;
//     the first form of this kernel (one block of four output tiles per pass: the operand re-read once per block, prefetched
//     16 MFMAs ahead) ran at 0.22 of the fp32 MFMA peak, waiting for those loads (profiles/r06/generic_arch_v1.jsonl);
//   * weights: fp32 tiles `tile(M, nt, kt)` in consumption order [pass][k tile][tile of the pass], ONE stream per step (encoder forward |
//     trunk forward | trunk backward | encoder backward) through the same five-slot LDS-DMA ring as the fused kernels (pndf_device.h:
//     slots of 16 tiles shared by the four waves, fetched four slots = 8,192 cycles of matrix pipe ahead, counted vmcnt wait + one
//     barrier in the middle of a slot); see "the trunk's weight ring" and gen_layer below for the forms that were measured before it.
// Roofline: fp32 MFMA (157.3 TFLOP/s); algorithmic work per pose-step 4 x sum_l in_l out_l FLOP.  Not the benchmark path
// (BASELINE.json names amass.yaml).  Measured (tools/bench_generic.py, B = 65,536 x 10 steps, profiles/r06/generic_arch*.jsonl), as a
// fraction of the fp32 MFMA peak: 0.80 on configs/amass.yaml itself (PNDF_FORCE_GENERIC=1; the fused exact-fp32 kernel: 0.89), 0.73
// with Softplus, 0.83 on 512-1024-1024-640-256-128; the split-precision form: 13.5 ms per launch against 28.5 ms (the fused split
// kernel: 9.4 ms); where the rest goes: profiles/r06/generic_ablate.txt.
#include "pndf_device.h"

#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "pndf_generic.h"
#include "pndf_host.h"
#include "pndf_pack.h"

namespace {

constexpr int NTB = PNDF_GEN_NTB;
constexpr int TILE_F4 = 64;              // f32x4 elements of a weight tile (one per lane)
constexpr int SLOT_F4 = WG_THREADS;      // f32x4 elements of a scratch tile slot (one per thread of the workgroup)

// ---- the trunk's weights come through the SAME ring as in the fused kernels (pndf_device.h: Ring)
// One stream per step, in consumption order: encoder forward tiles (3 slots) | trunk forward passes, then the transposed matrices in
// the backward pass's order | encoder backward tiles (3 slots), followed by a replica of its first slots so that the fetch pointer
// never wraps inside a step.  Slots of 16 tiles, each wave DMAs a quarter of a slot, five buffers = a slot is fetched FOUR slots
// ahead, the counted vmcnt wait + one barrier sit in the middle of the slot being consumed.  Because the four waves SHARE a slot, a
// slot lasts 64 MFMAs per wave = 2,048 cycles and the look-ahead is 8,192 cycles for the ring's 80 KiB -- four times what a ring per
// wave buys with the same LDS (v5 - v7 of this file: L2 hit rate 78 %, 17 % of the wave cycles stalled: profiles/r06/generic_v7/), and
// the waves of a workgroup walk the stream in step, as the fused kernels' do.
// History of this loop on configs/amass.yaml (fraction of the fp32 MFMA peak; the fused exact-fp32 kernel: 0.89):
//   v1 0.22  one block of four output tiles per pass over the operand, weights and operand prefetched one k step through registers
//   v2 0.45  the accumulators of a pass (32 tiles) resident: the operand is read once per pass
//   v3 0.47  weight groups of eight tiles (1,024 cycles of look-ahead)         -> rocprofv3: 48 % of the wave cycles at a waitcnt, L2 hit 76 %
//   v4 0.41  a group shared by the four waves through LDS, ONE group ahead, a barrier per group (every wave gets the slowest wave's miss)
//   v5 0.60  a five-slot ring PRIVATE to each wave, fed by LDS-DMA four slots (2,048 cycles) ahead, no barrier in the trunk
//   v6 0.68  the slot laid out by hand: DMA pieces and the next slot's tile reads between the rounds of MFMAs
//   v7 0.68  one weight stream per step (no ring start-up per pass), operand read pipelined
//   v8 0.65  the fused kernels' shared ring, the non-MFMA items in four blocks between rounds of eight MFMAs (two register sets, copies)
//   v9 0.66  the shared ring, one item behind each MFMA, the next group's tiles read in ONE round straight into the registers they replace
//   v10 0.46 two register sets again, reads and copies spread over the group: hipcc answers with 3.7 register moves per MFMA
//   v11 0.66 MFMAs in tile pairs, every read straight into the registers its pair just released, spread evenly
//   v12 0.74 nothing decided at run time inside a k step (see gen_layer)
//   v13 0.78 this: the backward epilogue's derivative loads one group ahead of its stores (gen_backward), non-temporal stores (gen_store)
// The operand tile of a k step is the wave's own (its 16 poses): it comes by DMA as well, three k steps deep, into three 1-KiB buffers
// per wave in the chunk-mask rows (unused here), so that no compiler-visible vector-memory instruction sits in the k loop.
constexpr int GX_BUFS = 3;
constexpr int GX_WAVE_BYTES = GX_BUFS * TILE_BYTES;
static_assert(4 * GX_WAVE_BYTES <= MASK_ROWS * WG_THREADS, "the operand buffers live in the chunk-mask rows");
static_assert(NTB * 2 == SLOT_TILES, "a group of output tiles is half a ring slot: groups alternate between the two halves");

struct GenLds {
    uint32_t x_lds;        // uniform: LDS address of this wave's operand buffers
    const char* x_ptr;     // per lane: its 16 bytes of operand buffer 0
    uint32_t lane16;
};
__device__ __forceinline__ void gw_dma_tile(const char* base, uint32_t voff, uint32_t dst) {
    // the operand tiles are read ONCE: fetched non-temporally they do not push the weight stream out of L2 (arm 512 of PNDF_GEN_ABLATE:
    // plain; with the derivative loads below 15.2 -> 14.3 ms per launch on the split path, Softplus 17.5 -> 15.7; nothing on the fp32 path)
    if (PNDF_GEN_ABLATE & 512)
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(base), "s"(dst) : "memory", "m0");
    else
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt" : : "v"(voff), "s"(base), "s"(dst) : "memory", "m0");
}
// a derivative-factor tile of a backward epilogue: read once as well (arm 1024: plain)
__device__ __forceinline__ f32x4 gen_load_d(const f32x4* p) {
    if (PNDF_GEN_ABLATE & 1024) return *p;
    return __builtin_nontemporal_load(p);
}

// the tiles of the group being multiplied
struct GenW {
    f32x4 wc[NTB];
};
// the first group of the trunk (tile 0 of the slot behind the encoder's three)
__device__ __forceinline__ void gen_trunk_begin(Ring& ring, GenW& W) {
    ring_boundary(ring);
#pragma unroll
    for (int j = 0; j < NTB; ++j) W.wc[j] = ring_tile(ring, j);
}
// the end of the trunk's stream.  The last group has already run the events of the group behind it, which is padding (`half` = the half of
// its slot that one is): the second half of the last slot -- nothing left to do --, or the first half of a slot of padding, whose
// mid-slot events still have to happen
__device__ __forceinline__ void gen_trunk_end(Ring& ring, int half) {
    if (half == 0) {
        ring_midslot_sync(ring);
        DmaSrc src;
        uint32_t dst;
        ring_dma_begin(ring, src, dst);
        dst = __builtin_amdgcn_readfirstlane(dst);      // (see gen_layer)
#pragma unroll
        for (int j = 0; j < 4; ++j) ring_dma_piece(src, dst, j);
    }
}

// acc[t] += sum_k W(t, k) X[k] for `NG * NTB` output tiles (one pass over a layer) at once: the next nk * NG groups of the stream.
// `xbase` (uniform) = this wave's 1 KiB of operand tile 0 (tiles 4 KiB apart); H0 = the half of its slot the pass's first group is.
// Register indices must be compile-time, so the number of groups is a template parameter: the plan rounds a layer's output tiles up to
// whole groups (host side: gen_round_tiles) and the layer code is instantiated per group count of a pass.  (A first form kept 64
// accumulators under run-time guards `if (group < ng)`: hipcc answered with 700 spilled registers.)
// One wave per SIMD overlaps its own non-MFMA instructions with its own MFMAs only when they sit between them in program order, and
// an fp32 MFMA covers 32 cycles: at most ONE memory item behind each MFMA.  A group = 32 MFMAs over its 8 tiles in PAIRS:
// m = 8 pr + 2 s + h multiplies k sub-step s of tile 2 pr + h (two interleaved accumulator chains, as part B of the fused kernels), so
// a tile's registers are dead after 8 MFMAs and the next group's tile is read straight into them -- no second register set, no copies,
// and the eight reads of a group are spread evenly over it:
//   m = 2              (last group of a k step) the NEXT k step's operand tile: counted wait + LDS read
//   m = 4              the ring events of the NEXT group: slot boundary, or counted wait + barrier
//   m = 8 pr + 6 + h   tile 2 pr + h of the next group -> wc[2 pr + h] (its last MFMA is this one; needed again 26 MFMAs later)
//   m = 9, 11, 17, 19  one 1-KiB piece each of the slot fetch (groups whose successor is the second half of a slot)
// NOTHING in a k step is decided at run time: which half of a slot a group is is a template parameter (H0, alternating; the pass
// dispatcher tracks it), and every group has a successor in the stream -- the host pads the trunk's section so that the group behind the
// last one is padding (pndf_generic_create).  With run-time flags for both (v11) there was a branch behind every third MFMA
// (SQ_INSTS_BRANCH 4.9e8 against 1.77e9 MFMAs), and the reads, fetch pieces and epilogues that the fused kernels hide completely behind
// their MFMAs cost 8 ms of a 34 ms launch.
// Measured on the way here (profiles/r06/generic_ablate.txt; the bare MFMA loop of this kernel runs at 0.87 with the encoder): the four
// waves of a workgroup run in step behind the ring's barrier, so what one wave does in a round all four do.
//   v8  k-sub-step-major MFMAs, two register sets, the reads in two blocks and the 32 copies in one block between groups: 0.65
//   v9  the eight reads of a group in its LAST round, each straight into the registers its MFMA just released: 32 KiB for the LDS in 256
//       cycles -- exactly its bandwidth; the reads cost 7.5 % of the launch at the waits of the next round: 0.66
//   v10 two register sets, a read behind every other MFMA of rounds 0 - 1, copies behind round 3: hipcc keeps one set in AGPRs and
//       renames the accumulators (v_mfma with vDst != SrcC), 3.7 moves per MFMA (SQ_INSTS_VALU 9.0e9 against 2.3e9): 0.46
// vmcnt bookkeeping (loads retire in order; other operations of the wave in the queue only make a counted wait stricter): the ring's own
// wait is pndf_device.h's (at most two slot fetches of this wave in flight: vmcnt(8)).  Operand tile k + 1 is issued at the top of k step
// k - 1 and read at m = 2 of the last group of k step k: operand tile k + 2 and the slot fetches of at least NG - 1 groups are younger
// -> vmcnt(4 (NG - 1) + 1).  Operand tiles 0 and 1 of a pass are the youngest operations when they are needed: vmcnt(0), once per pass.
template <int NG, int H0>
__device__ __forceinline__ void gen_layer(Ring& ring, GenW& W, const char* xbase, int nk, f32x4 (&acc)[NG * NTB], const GenLds& L) {
    constexpr int XWAIT = 4 * (NG - 1) + 1;
    gw_dma_tile(xbase, L.lane16, L.x_lds);
    gw_dma_tile(xbase, (uint32_t)(nk > 1 ? 1 : 0) * (SLOT_F4 * 16u) + L.lane16, L.x_lds + TILE_BYTES);
    if (!(PNDF_GEN_ABLATE & 32)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    f32x4 xc = *(const f32x4*)L.x_ptr, xn = xc;
    uint32_t xb = 0;                                               // operand buffer of k step k: k mod 3
    // one k step: NG groups, the first one in half HS of its slot
    auto kstep = [&](auto hs, int k) {
        constexpr int HS = decltype(hs)::value;
        const uint32_t xb1 = (xb == GX_BUFS - 1) ? 0u : xb + 1, xb2 = (xb1 == GX_BUFS - 1) ? 0u : xb1 + 1;
        // operand tile k + 2 -> the buffer tile k - 1 was read from (during k step k - 2)
        if (!(PNDF_GEN_ABLATE & 4)) gw_dma_tile(xbase, (uint32_t)((k + 2 < nk) ? k + 2 : nk - 1) * (SLOT_F4 * 16u) + L.lane16, L.x_lds + xb2 * TILE_BYTES);
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            const int nh = ((HS + gi) & 1) ^ 1;                   // the half of its slot the NEXT group of the stream is
            DmaSrc src{nullptr, 0u};
            uint32_t dst = 0;
#pragma unroll
            for (int m = 0; m < 4 * NTB; ++m) {
                const int pr = m / 8, sI = (m % 8) / 2, j = 2 * pr + (m & 1);
                acc[gi * NTB + j] = mfma4(W.wc[j][sI], xc[sI], acc[gi * NTB + j]);      // two interleaved chains per pair of tiles
                __builtin_amdgcn_sched_barrier(0);
                if (m == 2 && gi == NG - 1) {
                    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(XWAIT) : "memory");
                    xn = *(const f32x4*)(L.x_ptr + xb1 * TILE_BYTES);
                }
                if (m == 4 && !(PNDF_GEN_ABLATE & 128)) {
                    if (nh == 0) ring_boundary(ring);
                    else ring_midslot_sync(ring);
                }
                if (m % 8 >= 6 && !(PNDF_GEN_ABLATE & 64)) W.wc[j] = ring_tile(ring, nh * NTB + j);
                if ((m == 9 || m == 11 || m == 17 || m == 19) && nh == 1) {
                    // (under SGPR pressure hipcc keeps ring state in VGPR lanes and hands one to `s_mov_b32 m0`: readfirstlane is free
                    // when the value is in an SGPR already)
                    if (m == 9) {
                        ring_dma_begin(ring, src, dst);
                        dst = __builtin_amdgcn_readfirstlane(dst);
                    }
                    ring_dma_piece(src, dst, m == 9 ? 0 : m == 11 ? 1 : m == 17 ? 2 : 3);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        xc = xn;
        xb = xb1;
    };
    if constexpr (NG % 2 == 0) {
        for (int k = 0; k < nk; ++k) kstep(std::integral_constant<int, H0>{}, k);
    } else {                      // an odd group count flips the half from one k step to the next: two k steps per iteration
        int k = 0;
        for (; k + 1 < nk; k += 2) {
            kstep(std::integral_constant<int, H0>{}, k);
            kstep(std::integral_constant<int, H0 ^ 1>{}, k + 1);
        }
        if (k < nk) kstep(std::integral_constant<int, H0>{}, k);
    }
    __builtin_amdgcn_sched_barrier(0);
}

// a tile store of an epilogue: non-temporal -- 2 MB per workgroup-step that nobody reads before 64 KiB .. 1 MB more have been written
// (28.8 against 29.6 ms per launch with plain stores, arm 256 of PNDF_GEN_ABLATE: every counted vmcnt wait behind an epilogue waits
// for its stores as well, gfx950 has one counter for loads and stores)
__device__ __forceinline__ void gen_store(f32x4* p, const f32x4& v) {
    if (PNDF_GEN_ABLATE & 256) *p = v;
    else __builtin_nontemporal_store(v, p);
}

// hidden activation of one D tile (reference net_modules.py:30-41,64-65) and its derivative factor
template <bool SP>
__device__ __forceinline__ void gen_act(f32x4& z, f32x4& dfac, float slope, const SpK& k) {
    if constexpr (SP) {
        act_softplus4<PNDF_SP_FORM, false>(z, k, dfac);
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool pos = z[r] > 0.0f;             // PyTorch: relu'(0) = 0, lrelu'(0) = slope
            dfac[r] = pos ? 1.0f : slope;
            z[r] = pos ? z[r] : z[r] * slope;
        }
    }
}

// relu family: the derivative of a hidden unit is one BIT (z > 0: PyTorch's convention at 0), 32 per lane for a group of eight tiles --
// a dword per lane and group in the scratch instead of eight fp32 tiles (the scratch's traffic is what bounds the split path: HBM)
__device__ __forceinline__ void gen_act_bits(f32x4& z, float slope, uint32_t& bits, int j) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const bool pos = z[r] > 0.0f;
        bits |= pos ? (1u << (4 * j + r)) : 0u;
        z[r] = pos ? z[r] : z[r] * slope;
    }
}
__device__ __forceinline__ f32x4 gen_dfac_bits(uint32_t bits, int j, float slope) {
    f32x4 d;
#pragma unroll
    for (int r = 0; r < 4; ++r) d[r] = ((bits >> (4 * j + r)) & 1u) ? 1.0f : slope;
    return d;
}

// one forward layer with NG groups of output tiles: bias -> accumulate -> (hidden layers) activation, derivative factor
template <int NG, int H0, bool SP>
__device__ __forceinline__ void gen_forward(Ring& ring, GenW& W, const float* bias, const char* xin, f32x4* xout, f32x4* dl, uint32_t* mk, int nk,
                                            bool last, float slope, const SpK& k, int g, f32x4& zlast, const GenLds& L) {
    f32x4 acc[NG * NTB];
#pragma unroll
    for (int t = 0; t < NG * NTB; ++t) acc[t] = (PNDF_GEN_ABLATE & 8) ? f32x4{0.f, 0.f, 0.f, 0.f} : *(const f32x4*)(bias + 16 * t + 4 * g);
    // the bias loads are consumed HERE: left pending, hipcc waits for them with `s_waitcnt vmcnt(0)` at the head of the k loop --
    // in every iteration, which drains the weight ring's look-ahead once per k step
#pragma unroll
    for (int t = 0; t < NG * NTB; ++t) asm volatile("" : "+v"(acc[t]));
    gen_layer<NG, H0>(ring, W, xin, nk, acc, L);
    if (last) {                    // the output layer: one unit, row 0 of tile 0; its activation is the caller's
        zlast = acc[0];
        return;
    }
    uint32_t bits = 0;
#pragma unroll
    for (int t = 0; t < NG * NTB; ++t) {
        f32x4 df = acc[t];
        if constexpr (SP) {
            if (!(PNDF_GEN_ABLATE & 16)) gen_act<SP>(acc[t], df, slope, k);
        } else {
            if (!(PNDF_GEN_ABLATE & 16)) gen_act_bits(acc[t], slope, bits, t % NTB);
        }
        if (!(PNDF_GEN_ABLATE & 1) || t == 0) {
            gen_store(xout + (size_t)t * SLOT_F4, acc[t]);
            if constexpr (SP) gen_store(dl + (size_t)t * SLOT_F4, df);
        } else {
            asm volatile("" : : "v"(acc[t]), "v"(df));      // (the arm keeps the arithmetic)
        }
        if (t % NTB == NTB - 1) {
            if constexpr (!SP) {
                __builtin_nontemporal_store(bits, mk + (size_t)(t / NTB) * WG_THREADS);
                bits = 0;
            }
            __builtin_amdgcn_sched_barrier(0);      // a group at a time: the accumulators fill up to half the register file
        }
    }
}

// one backward layer: G_in = W^T G_out, times the derivative factors of the layer below (l > 0) or into the pose's feature row
template <int NG, int H0, bool SP>
__device__ __forceinline__ void gen_backward(Ring& ring, GenW& W, const char* gin, f32x4* gout, const f32x4* dprev, const uint32_t* mkprev, float slope,
                                             float* my_f, int nk, int g, int t0, const GenLds& L) {
    f32x4 acc[NG * NTB];
#pragma unroll
    for (int t = 0; t < NG * NTB; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    gen_layer<NG, H0>(ring, W, gin, nk, acc, L);
    if (!SP && dprev) {
        // relu family: one dword of derivative bits per lane and group
        uint32_t mb[NG];
#pragma unroll
        for (int gq = 0; gq < NG; ++gq) mb[gq] = (PNDF_GEN_ABLATE & 2) ? 0xffffffffu : __builtin_nontemporal_load(mkprev + (size_t)gq * WG_THREADS);
#pragma unroll
        for (int t = 0; t < NG * NTB; ++t) {
            const f32x4 go = acc[t] * gen_dfac_bits(mb[t / NTB], t % NTB, slope);
            if (!(PNDF_GEN_ABLATE & 1) || t == 0) gen_store(gout + (size_t)t * SLOT_F4, go);
            else asm volatile("" : : "v"(go));
            if (t % NTB == NTB - 1) __builtin_amdgcn_sched_barrier(0);
        }
    } else if (dprev) {
        // x act'(z_{l-1}), a group of tiles at a time with the NEXT group's derivative factors already on their way: in program order
        // L0 | L1 S0 | L2 S1 | ..., so the counted wait hipcc puts in front of a group's multiplies (vmcnt counts loads and stores alike,
        // in order) never includes a store.  With load - multiply - store group after group, every group's loads queued behind the
        // previous group's stores: 3 ms of a 30 ms launch (arms 1, 2 and 3 of PNDF_GEN_ABLATE, profiles/r06/generic_ablate.txt).
        f32x4 dp[2][NTB];
#pragma unroll
        for (int j = 0; j < NTB; ++j) dp[0][j] = (PNDF_GEN_ABLATE & 2) ? f32x4{1.f, 1.f, 1.f, 1.f} : gen_load_d(dprev + (size_t)j * SLOT_F4);
#pragma unroll
        for (int gq = 0; gq < NG; ++gq) {
            if (gq + 1 < NG) {
#pragma unroll
                for (int j = 0; j < NTB; ++j)
                    dp[(gq + 1) & 1][j] = (PNDF_GEN_ABLATE & 2) ? f32x4{1.f, 1.f, 1.f, 1.f} : gen_load_d(dprev + (size_t)((gq + 1) * NTB + j) * SLOT_F4);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NTB; ++j) {
                const int t = gq * NTB + j;
                const f32x4 go = acc[t] * dp[gq & 1][j];
                if (!(PNDF_GEN_ABLATE & 1) || t == 0) gen_store(gout + (size_t)t * SLOT_F4, go);
                else asm volatile("" : : "v"(go));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {                       // d z_out / d x0 into the pose's feature row (t0 = 0: one pass)
#pragma unroll
        for (int t = 0; t < NG * NTB; ++t)
            if (t0 + t < 8) *(f32x4*)(my_f + 16 * (t0 + t) + 4 * g) = acc[t];
    }
}

// ------------------------------------------------------------------ the same trunk on split-precision fp16 MFMAs (precision f16x3)
// Every fp32 operand travels as fp16 hi + fp16 lo and a product block is three v_mfma_f32_16x16x32_f16 (hi hi + lo hi + hi lo, fp32
// accumulate; the dropped lo lo term is 2^-22 relative) -- pndf_kernel_split.hip's arithmetic, with the layer-by-layer structure of this
// file.  One MFMA contracts 32 k = TWO operand tiles, whose registers -- scaled, converted and packed -- are the B operand as they stand
// (pndf_layout.h "split-precision stream": block(M, nt, kb)); a weight block is a PAIR of 1-KiB tiles (hi, lo), a group = 8 output tiles
// x one k block = 8 pairs = ONE ring slot = 24 MFMAs, so every group runs the same ring events (no slot parity here).
// Operand scaling (exact: every factor is a power of two), simpler than the fused kernels' because a layer is complete before the
// next one starts: the stream carries s_l W (s_l: the largest |weight| of the layer in [2^12, 2^13), chosen by the packer), and every
// pose scales the operand tensor of a layer by its own sigma(p) with max_i |x_i(p)| sigma in [2^13, 2^14) -- the bound is MEASURED for
// every layer (running maximum over the epilogues of the producing layer + two cross-lane steps); the hi halves cannot overflow for
// any finite weights and poses, and the lo half of every value down to 2^-10 of the pose's largest stays a NORMAL fp16.  The fp32
// accumulators hold s_l sigma x the true value (they start from b s_l sigma); one multiply per value in the epilogue (`to_true`)
// brings it back, and the scratch holds TRUE fp32 values exactly as on the fp32 path.
typedef _Float16 gf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 gf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned gu32x4 __attribute__((ext_vector_type(4)));
struct GBlk {          // one k block (32 k) of the operand as B operand
    gf16x8 h, l;
};
struct GenWS {         // the eight weight pairs of the group being multiplied (tiles 2 j = hi, 2 j + 1 = lo of its slot)
    gf16x8 h[NTB], l[NTB];
};
__device__ __forceinline__ f32x4 gmf16(gf16x8 a, gf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
// (a, b) -> packed fp16 pairs: hi = rtz (one v_cvt_pkrtz), lo = rne(a - hi_a, b - hi_b) -- the remainders are exact in fp32, and a lo
// half rounded to NEAREST keeps the residual unbiased (pndf_kernel_split.hip split2: the same three instructions)
__device__ __forceinline__ void gsplit2(float a, float b, unsigned& hi, unsigned& lo) {
    const unsigned hp = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(a, b));
    hi = hp;
    unsigned l;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=&v"(l) : "v"(hp), "v"(a), "v"(b));
    lo = l;
}
// two fp32 C/D tiles (already scaled) -> the B operand of the k block they form
__device__ __forceinline__ void gpack_blk(const f32x4& t0, const f32x4& t1, GBlk& o) {
    unsigned h0, h1, h2, h3, l0, l1, l2, l3;
    gsplit2(t0[0], t0[1], h0, l0);
    gsplit2(t0[2], t0[3], h1, l1);
    gsplit2(t1[0], t1[1], h2, l2);
    gsplit2(t1[2], t1[3], h3, l3);
    o.h = __builtin_bit_cast(gf16x8, gu32x4{h0, h1, h2, h3});
    o.l = __builtin_bit_cast(gf16x8, gu32x4{l0, l1, l2, l3});
}
// |x| <= bound -> the power of two sigma with bound sigma in [2^13, 2^14), clamped to 2^-40 .. 2^40 (pndf_kernel_split.hip pose_scale)
__device__ __forceinline__ float gen_pose_scale(float bound) {
    uint32_t e = (__builtin_bit_cast(uint32_t, bound) >> 23) & 0xffu;
    e = e < 100u ? 100u : (e > 180u ? 180u : e);
    return __builtin_bit_cast(float, (267u - e) << 23);
}
// The same for a GRADIENT tensor, whose magnitude has no floor: behind a one-unit Softplus layer in its saturated branch the whole
// gradient of a pose is e^(beta z) ~ 1e-20 -- in fp32's range, and 2^-40 of clamp away from a normal fp16 (found by tools/sweep_generic.py:
// 100 % error on such poses).  No bias enters the backward accumulators, so the scale may be as large as fp32 carries the reciprocal of
// s_l sigma: bounds down to 2^-80 keep their place in the fp16 range.
__device__ __forceinline__ float gen_pose_scale_grad(float bound) {
    uint32_t e = (__builtin_bit_cast(uint32_t, bound) >> 23) & 0xffu;
    e = e < 47u ? 47u : (e > 180u ? 180u : e);
    return __builtin_bit_cast(float, (267u - e) << 23);
}
__device__ __forceinline__ float gen_pow2_rcp(float p) { return __builtin_bit_cast(float, 0x7F000000u - __builtin_bit_cast(uint32_t, p)); }
// a pose's rows live in the four lane groups (lane = 16 g + p): maximum over them
__device__ __forceinline__ float gen_pose_max(float m) {
    m = fmaxf(m, __shfl_xor(m, 16));
    return fmaxf(m, __shfl_xor(m, 32));
}
__device__ __forceinline__ float gen_absmax4(float m, const f32x4& v) {
    return fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
}

// the operand buffers of the split path: three k blocks of 2 KiB per wave, in the wave's OWN feature rows (LDS_F: free between the copy
// of x0 into the scratch and the last backward epilogue, which writes them when the wave's k loop is over)
constexpr int GXS_BLK_BYTES = 2 * TILE_BYTES;
static_assert(GX_BUFS * GXS_BLK_BYTES <= 16 * FSTRIDE * 4, "the split path's operand buffers fit the wave's feature rows");
static_assert((16 * FSTRIDE * 4) % 16 == 0, "16-byte aligned per wave");

__device__ __forceinline__ void gen_trunk_begin_split(Ring& ring, GenWS& W) {
    ring_boundary(ring);
#pragma unroll
    for (int j = 0; j < NTB / 2; ++j) {
        W.h[j] = __builtin_bit_cast(gf16x8, ring_tile(ring, 2 * j));
        W.l[j] = __builtin_bit_cast(gf16x8, ring_tile(ring, 2 * j + 1));
    }
    ring_midslot_sync(ring);
    {
        DmaSrc src;
        uint32_t dst;
        ring_dma_begin(ring, src, dst);
        dst = __builtin_amdgcn_readfirstlane(dst);
#pragma unroll
        for (int j = 0; j < 4; ++j) ring_dma_piece(src, dst, j);
    }
#pragma unroll
    for (int j = NTB / 2; j < NTB; ++j) {
        W.h[j] = __builtin_bit_cast(gf16x8, ring_tile(ring, 2 * j));
        W.l[j] = __builtin_bit_cast(gf16x8, ring_tile(ring, 2 * j + 1));
    }
}

// acc[t] += sum_kb W(t, kb) X[kb] for NG * NTB output tiles: the next nkb * NG slots of the stream.  `sigma` = this lane's pose's scale
// of the operand tensor.  A group = 24 MFMAs, m = 6 pr + 2 term + h over the tile pairs pr = 0 .. 3 (two interleaved accumulator
// chains; term 0: Wh Xh, 1: Wl Xh, 2: Wh Xl), and the NEXT slot's tiles are read straight into the registers their last MFMA released:
//   m = 0                 slot boundary                      m = 6 pr + 2 + h   lo tile of pair 2 pr + h of the next slot
//   m = 1  (last group of a k step) counted wait + the two raw operand tiles of the next k block
//   m = 6, 7  (same)      scale, split and pack them         m = 6 pr + 4 + h   hi tile of pair 2 pr + h of the next slot
//   m = 12                counted wait + barrier + the first piece of the slot fetch; m = 13, 18, 19: its other pieces
// (the reads of the next slot's second half -- pairs 4 .. 7, from m = 14 -- come behind the mid-slot events, pndf_device.h's protocol).
// The phase is LDS-read bound like the fused split kernels (2 KiB of weight pair per 3 MFMAs per wave).
// vmcnt: operand block kb + 2 is issued (two tiles) at the top of k step kb and read at m = 1 of the last group of k step kb + 1; younger
// by then: the two tiles of block kb + 3 and the fetch pieces of 2 NG - 1 groups -> vmcnt(8 NG - 2).
template <int NG>
__device__ __forceinline__ void gen_layer_split(Ring& ring, GenWS& W, const char* xbase, int nkb, float sigma, f32x4 (&acc)[NG * NTB], const GenLds& L) {
    constexpr int XWAIT = 8 * NG - 2;
    auto dma_blk = [&](int kb, uint32_t buf) {
        gw_dma_tile(xbase, (uint32_t)(2 * kb) * (SLOT_F4 * 16u) + L.lane16, L.x_lds + buf * GXS_BLK_BYTES);
        gw_dma_tile(xbase, (uint32_t)(2 * kb + 1) * (SLOT_F4 * 16u) + L.lane16, L.x_lds + buf * GXS_BLK_BYTES + TILE_BYTES);
    };
    dma_blk(0, 0);
    dma_blk(nkb > 1 ? 1 : 0, 1);
    if (!(PNDF_GEN_ABLATE & 32)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    GBlk xc, xn;
    {
        const f32x4 ra = *(const f32x4*)L.x_ptr, rb = *(const f32x4*)(L.x_ptr + TILE_BYTES);
        gpack_blk(ra * sigma, rb * sigma, xc);
        xn = xc;
    }
    uint32_t xb = 0;
    for (int kb = 0; kb < nkb; ++kb) {
        const uint32_t xb1 = (xb == GX_BUFS - 1) ? 0u : xb + 1, xb2 = (xb1 == GX_BUFS - 1) ? 0u : xb1 + 1;
        if (!(PNDF_GEN_ABLATE & 4)) dma_blk((kb + 2 < nkb) ? kb + 2 : nkb - 1, xb2);
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            DmaSrc src{nullptr, 0u};
            uint32_t dst = 0;
            f32x4 ra = f32x4{0.f, 0.f, 0.f, 0.f}, rb = ra;
#pragma unroll
            for (int m = 0; m < 3 * NTB; ++m) {
                const int pr = m / 6, term = (m % 6) / 2, j = 2 * pr + (m & 1);
                acc[gi * NTB + j] = gmf16(term == 1 ? W.l[j] : W.h[j], term == 2 ? xc.l : xc.h, acc[gi * NTB + j]);
                __builtin_amdgcn_sched_barrier(0);
                if (m == 0) ring_boundary(ring);
                if (m == 1 && gi == NG - 1) {
                    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(XWAIT) : "memory");
                    ra = *(const f32x4*)(L.x_ptr + xb1 * GXS_BLK_BYTES);
                    rb = *(const f32x4*)(L.x_ptr + xb1 * GXS_BLK_BYTES + TILE_BYTES);
                }
                if (m % 6 == 2 || m % 6 == 3) W.l[j] = __builtin_bit_cast(gf16x8, ring_tile(ring, 2 * j + 1));
                if (m % 6 == 4 || m % 6 == 5) W.h[j] = __builtin_bit_cast(gf16x8, ring_tile(ring, 2 * j));
                if (m == 6 && gi == NG - 1) {
                    ra = ra * sigma;
                    rb = rb * sigma;
                }
                if (m == 7 && gi == NG - 1) gpack_blk(ra, rb, xn);
                if (m == 12) {
                    ring_midslot_sync(ring);
                    ring_dma_begin(ring, src, dst);
                    dst = __builtin_amdgcn_readfirstlane(dst);
                    ring_dma_piece(src, dst, 0);
                }
                if (m == 13) ring_dma_piece(src, dst, 1);
                if (m == 18) ring_dma_piece(src, dst, 2);
                if (m == 19) ring_dma_piece(src, dst, 3);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        xc = xn;
        xb = xb1;
    }
    __builtin_amdgcn_sched_barrier(0);
}

template <int NG, bool SP>
__device__ __forceinline__ void gen_forward_split(Ring& ring, GenWS& W, const float* bias, const char* xin, f32x4* xout, f32x4* dl, uint32_t* mk, int nkb,
                                                  bool last, float slope, const SpK& k, int g, f32x4& zlast, const GenLds& L, float sigma,
                                                  float to_true, float& amax) {
    const float bscale = gen_pow2_rcp(to_true);
    f32x4 acc[NG * NTB];
#pragma unroll
    for (int t = 0; t < NG * NTB; ++t) acc[t] = (PNDF_GEN_ABLATE & 8) ? f32x4{0.f, 0.f, 0.f, 0.f} : *(const f32x4*)(bias + 16 * t + 4 * g) * bscale;
#pragma unroll
    for (int t = 0; t < NG * NTB; ++t) asm volatile("" : "+v"(acc[t]));      // (see gen_forward)
    gen_layer_split<NG>(ring, W, xin, nkb, sigma, acc, L);
    if (last) {
        zlast = acc[0] * to_true;
        return;
    }
    uint32_t bits = 0;
#pragma unroll
    for (int t = 0; t < NG * NTB; ++t) {
        f32x4 z = acc[t] * to_true, df = z;
        if constexpr (SP) gen_act<SP>(z, df, slope, k);
        else gen_act_bits(z, slope, bits, t % NTB);
        amax = gen_absmax4(amax, z);
        if (!(PNDF_GEN_ABLATE & 1) || t == 0) {
            gen_store(xout + (size_t)t * SLOT_F4, z);
            if constexpr (SP) gen_store(dl + (size_t)t * SLOT_F4, df);
        } else {
            asm volatile("" : : "v"(z), "v"(df));
        }
        if (t % NTB == NTB - 1) {
            if constexpr (!SP) {
                __builtin_nontemporal_store(bits, mk + (size_t)(t / NTB) * WG_THREADS);
                bits = 0;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

template <int NG, bool SP>
__device__ __forceinline__ void gen_backward_split(Ring& ring, GenWS& W, const char* gin, f32x4* gout, const f32x4* dprev, const uint32_t* mkprev, float slope,
                                                   float* my_f, int nkb, int g, int t0, const GenLds& L, float sigma, float to_true, float& amax) {
    f32x4 acc[NG * NTB];
#pragma unroll
    for (int t = 0; t < NG * NTB; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    gen_layer_split<NG>(ring, W, gin, nkb, sigma, acc, L);
    if (!SP && dprev) {            // relu family: derivative bits (see gen_backward)
        uint32_t mb[NG];
#pragma unroll
        for (int gq = 0; gq < NG; ++gq) mb[gq] = (PNDF_GEN_ABLATE & 2) ? 0xffffffffu : __builtin_nontemporal_load(mkprev + (size_t)gq * WG_THREADS);
#pragma unroll
        for (int t = 0; t < NG * NTB; ++t) {
            const f32x4 go = (acc[t] * to_true) * gen_dfac_bits(mb[t / NTB], t % NTB, slope);
            amax = gen_absmax4(amax, go);
            if (!(PNDF_GEN_ABLATE & 1) || t == 0) gen_store(gout + (size_t)t * SLOT_F4, go);
            else asm volatile("" : : "v"(go));
            if (t % NTB == NTB - 1) __builtin_amdgcn_sched_barrier(0);
        }
    } else if (dprev) {            // (the loads one group ahead of the stores: see gen_backward)
        f32x4 dp[2][NTB];
#pragma unroll
        for (int j = 0; j < NTB; ++j) dp[0][j] = (PNDF_GEN_ABLATE & 2) ? f32x4{1.f, 1.f, 1.f, 1.f} : gen_load_d(dprev + (size_t)j * SLOT_F4);
#pragma unroll
        for (int gq = 0; gq < NG; ++gq) {
            if (gq + 1 < NG) {
#pragma unroll
                for (int j = 0; j < NTB; ++j)
                    dp[(gq + 1) & 1][j] = (PNDF_GEN_ABLATE & 2) ? f32x4{1.f, 1.f, 1.f, 1.f} : gen_load_d(dprev + (size_t)((gq + 1) * NTB + j) * SLOT_F4);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NTB; ++j) {
                const int t = gq * NTB + j;
                const f32x4 go = (acc[t] * to_true) * dp[gq & 1][j];
                amax = gen_absmax4(amax, go);
                if (!(PNDF_GEN_ABLATE & 1) || t == 0) gen_store(gout + (size_t)t * SLOT_F4, go);
                else asm volatile("" : : "v"(go));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        // the operand buffers ARE this wave's feature rows, and the last k steps re-issue their final block (the counted waits want a
        // fixed number of operations per k step): such a fetch may still be in flight -- it must not land on top of the rows written here
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int t = 0; t < NG * NTB; ++t)
            if (t0 + t < 8) *(f32x4*)(my_f + 16 * (t0 + t) + 4 * g) = acc[t] * to_true;
    }
}

// the group counts the layer code is instantiated for (the plan rounds up to the next one: at most a third of a layer is padding)
// A pass keeps at most 4 groups = 32 tiles = 128 accumulator registers (a whole 1024-wide layer at once -- 256 -- left hipcc 150 - 220
// spilled registers: everything that is not an MFMA accumulator has to fit the 256 architectural VGPRs); wider layers take two
// passes, each of which reads the operand tiles once.
#define PNDF_GEN_GROUP_CASES(X) X(1, 0) X(1, 1) X(2, 0) X(2, 1) X(3, 0) X(3, 1) X(4, 0) X(4, 1)
constexpr int GEN_PASS_GROUPS = 4;

// SP: the trunk's activation is Softplus (else relu / lrelu); ESP: the encoder's.  Every config of the reference has ESP == SP;
// net_modules.py:128 reads model.StrEnc.act on its own, so the other two combinations exist as well.
// SPLIT: the trunk on split-precision fp16 MFMAs (precision f16x3 / f16), else exact fp32.
template <bool SP, bool ESP = SP, bool SPLIT = false>
__device__ __forceinline__ void pndf_generic_body(const PndfGenericArgs& args) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4;
    const int p = lane & 15;
    const int wp = wave * 16 + p;
    const int L = args.nlayers;

    // this workgroup's scratch: [0, 64) activations ping, [64, 128) pong, then the derivative factors (args.d_off)
    f32x4* const wg = (f32x4*)args.scratch + (size_t)blockIdx.x * args.wg_tiles * SLOT_F4 + tid;
    f32x4* const xbuf[2] = {wg, wg + (size_t)PNDF_GEN_XTILES * SLOT_F4};
    // the same two buffers as the LDS-DMA sees them: uniform address of this WAVE's 1 KiB of tile 0 (the lanes add 16 bytes each)
    const char* const xwave = (const char*)((f32x4*)args.scratch + (size_t)blockIdx.x * args.wg_tiles * SLOT_F4) + wave * TILE_BYTES;
    const char* const xuni[2] = {xwave, xwave + (size_t)PNDF_GEN_XTILES * SLOT_F4 * 16};
    GenLds gl;
    if constexpr (SPLIT) {           // (three k blocks of 2 KiB in the wave's own feature rows: see GXS_BLK_BYTES)
        gl.x_lds = (uint32_t)(size_t)(PNDF_LDS char*)(smem + LDS_F) + wave * (16 * FSTRIDE * 4);
        gl.x_ptr = smem + LDS_F + wave * (16 * FSTRIDE * 4) + lane * 16;
    } else {
        gl.x_lds = (uint32_t)(size_t)(PNDF_LDS char*)(smem + LDS_MASK) + wave * GX_WAVE_BYTES;
        gl.x_ptr = smem + LDS_MASK + wave * GX_WAVE_BYTES + lane * 16;
    }
    gl.lane16 = (uint32_t)lane * 16u;

    ActP ap;                         // the TRUNK's activation parameters ...
    ap.slope = args.slope;
    ap.k = sp_consts(args.beta);
    ap.sp = SpRef{nullptr, 0u};
    ap.stage = nullptr;
    ap.lane = lane;
    ActP ape = ap;                   // ... and the encoder's
    ape.slope = args.enc_slope;
    ape.k = sp_consts(args.enc_beta);
    // the encoder parks its 42 derivative tiles at slots SP_SLOT_ENC + i of `ape.sp` (pndf_device.h): point slot SP_SLOT_ENC at enc_d_off
    ape.sp = SpRef{ESP ? (const char*)((f32x4*)args.scratch + ((size_t)blockIdx.x * args.wg_tiles + args.enc_d_off) * SLOT_F4)
                           - (size_t)SP_SLOT_ENC * WG_THREADS * SP_LANE_BYTES
                     : nullptr,
                   (uint32_t)tid * SP_LANE_BYTES};

    float* const lds_bias = (float*)(smem + LDS_BIAS);
    float* const lds_q = (float*)(smem + LDS_Q);
    float* const my_q = lds_q + wp * NQ;
    float* const my_f = (float*)(smem + LDS_F) + wp * FSTRIDE;
    float* const my_gn = (float*)(smem + LDS_GN) + wp * FSTRIDE;   // aliases the feature row of the pose

    Ring ring;
    ring.gstream = args.enc_stream;      // the step's stream: encoder forward | trunk | encoder backward (+ replica of its first slots)
    ring.smem = smem;
    ring.lane = lane;
    ring.st_wait = ring.st_bar = 0;
    ring.st_n = 0;
    for (int i = tid; i < BIAS_FLOATS / 4; i += WG_THREADS) ((f32x4*)lds_bias)[i] = ((const f32x4*)args.bias)[i];

    const long long nblocks = (args.B + WG_POSES - 1) / WG_POSES;
    for (long long blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
        const long long pose0 = blk * WG_POSES;
        ring_start(ring, wave);      // slots 0..3 in flight; every block restarts the ring (the forward-only mode leaves it mid-stream)
        {
            long long nvalid = args.B - pose0;
            if (nvalid > WG_POSES) nvalid = WG_POSES;
            const f32x4* src = (const f32x4*)(args.q_in + pose0 * NQ);
            const int nvec = (int)nvalid * (NQ / 4);
            for (int i = tid; i < WG_POSES * (NQ / 4); i += WG_THREADS) {
                const int src_i = i < nvec ? i : (nvec - (NQ / 4) + (i % (NQ / 4)));
                ((f32x4*)lds_q)[i] = src[src_i];
            }
        }
        ring_wait_dma();
        __syncthreads();

        const int nsteps = (args.mode == MODE_PROJECT) ? args.steps : 1;
        float dval = 0.f;
        for (int step = 0; step < nsteps; ++step) {
            if (step) ring.fetch_off -= (uint32_t)args.w_slots * SLOT_BYTES;      // the fetch pointer has run one step's length (ring_next_step)
            // ---------------- encoder (or the normalised pose itself) -> x0, 128 rows in the pose's feature row
            uint32_t eb[6] = {0, 0, 0, 0, 0, 0};
            float poison = 0.f;
            if (args.noenc) {
                poison = noenc_forward<SP>(my_q, my_f, g);
                ring_skip_encoder_section(ring);
            } else {
                if constexpr (SP && !ESP) {      // a Softplus trunk behind a relu-family encoder: the NaN / inf poison of the pose (joint_axis_norms)
                    float ss[4];
                    poison = joint_axis_norms<true>(my_q, ss);
                }
                const float pe = encoder_forward<ESP>(my_q, my_f, lds_bias + ENCB_OFF, eb, ring, ape, g);
                if constexpr (ESP) poison = pe;
            }
            float amax = 0.f;              // (split path) largest |value| of this lane's rows of the tensor being produced
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const f32x4 v = *(const f32x4*)(my_f + 16 * t + 4 * g);
                xbuf[0][(size_t)t * SLOT_F4] = v;
                if constexpr (SPLIT) amax = gen_absmax4(amax, v);
            }

            // ---------------- trunk forward, layer by layer (net_modules.py:51-69)
            GenW wcur;
            GenWS wsp;
            int half = 0;                  // which half of its slot the next group of the stream is (uniform; fp32 path)
            f32x4 zlast = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (SPLIT) {
                // (the x0 tiles this lane just stored come back through the operand DMA: the pass start's vmcnt(0) covers them)
                gen_trunk_begin_split(ring, wsp);
                for (int l = 0; l < L; ++l) {
                    const int nkb = args.kb[l], ng = args.ntp[l] / NTB;
                    const float* bias = args.lbias + args.b_off[l];
                    f32x4* dl = wg + (size_t)args.d_off[l] * SLOT_F4;
                    uint32_t* mk = (uint32_t*)(wg - tid + (size_t)args.d_off[l] * SLOT_F4) + tid;      // relu family: a dword per lane and group
                    const float sigma = gen_pose_scale(gen_pose_max(amax));
                    const float to_true = args.w_inv[l] * gen_pow2_rcp(sigma);
                    amax = 0.f;
                    for (int g0 = 0; g0 < ng; g0 += GEN_PASS_GROUPS) {
                        const int n = (ng - g0 < GEN_PASS_GROUPS) ? ng - g0 : GEN_PASS_GROUPS, t0 = g0 * NTB;
                        switch (n) {
#define PNDF_GEN_FWDS(N) case N: gen_forward_split<N, SP>(ring, wsp, bias + 16 * t0, xuni[l & 1], xbuf[(l + 1) & 1] + (size_t)t0 * SLOT_F4, dl + (size_t)t0 * SLOT_F4, mk + (size_t)g0 * WG_THREADS, nkb, l == L - 1, args.slope, ap.k, g, zlast, gl, sigma, to_true, amax); break;
                            PNDF_GEN_FWDS(1) PNDF_GEN_FWDS(2) PNDF_GEN_FWDS(3) PNDF_GEN_FWDS(4)
#undef PNDF_GEN_FWDS
                            default: break;
                        }
                    }
                }
            } else {
            gen_trunk_begin(ring, wcur);
            for (int l = 0; l < L; ++l) {
                const int nk = args.kt[l], ng = args.ntp[l] / NTB;
                const float* bias = args.lbias + args.b_off[l];
                f32x4* dl = wg + (size_t)args.d_off[l] * SLOT_F4;
                uint32_t* mk = (uint32_t*)(wg - tid + (size_t)args.d_off[l] * SLOT_F4) + tid;          // relu family: a dword per lane and group
                for (int g0 = 0; g0 < ng; g0 += GEN_PASS_GROUPS) {      // passes of at most 4 groups of output tiles
                    const int n = (ng - g0 < GEN_PASS_GROUPS) ? ng - g0 : GEN_PASS_GROUPS, t0 = g0 * NTB;
                    switch (2 * n + half) {      // (stream order: [pass][k tile][tile of the pass])
#define PNDF_GEN_FWD(N, H) case 2 * N + H: gen_forward<N, H, SP>(ring, wcur, bias + 16 * t0, xuni[l & 1], xbuf[(l + 1) & 1] + (size_t)t0 * SLOT_F4, dl + (size_t)t0 * SLOT_F4, mk + (size_t)g0 * WG_THREADS, nk, l == L - 1, args.slope, ap.k, g, zlast, gl); break;
                        PNDF_GEN_GROUP_CASES(PNDF_GEN_FWD)
#undef PNDF_GEN_FWD
                        default: break;      // (pndf_generic_create plans no other group count)
                    }
                    half ^= (n * nk) & 1;
                }
            }
            }
            // row 0 of the output tile lives in register 0 of lane group 0: every lane of the pose reads it from there
            const float z7 = __shfl(zlast[0], p);
            float gz7;
            if constexpr (SP) {
                dval = act_softplus<PNDF_SP_FORM_OUT>(z7, ap.k, gz7) + poison;      // output Softplus, net_modules.py:39-41,69
                gz7 += poison;
            } else {
                dval = (z7 != z7) ? z7 : fmaxf(z7, 0.f);      // output ReLU for relu AND lrelu, net_modules.py:30-37 (relu(NaN) = NaN)
                gz7 = (z7 > 0.f) ? 1.f : 0.f;
                if constexpr (ESP) {                         // (a Softplus encoder swallows a NaN pose: v_min / v_max; see joint_axis_norms)
                    dval += poison;
                    gz7 += poison;
                }
            }
            if (args.mode == MODE_FORWARD) break;      // (the ring stays mid-stream: drained at the end of the block)
            float gscale = gz7;
            if (args.mode == MODE_FORWARD_GRAD && args.grad_out) {
                long long pidx = pose0 + wp;
                if (pidx >= args.B) pidx = args.B - 1;
                gscale = gz7 * args.grad_out[pidx];
            }

            // ---------------- trunk backward: the seed is d z_out / d z_out = 1 in row 0 (the output activation's derivative
            // and grad_outputs scale the result, as in the fused kernels)
            int cur = 0;
            xbuf[0][0] = (g == 0) ? f32x4{1.f, 0.f, 0.f, 0.f} : f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (SPLIT) {
                xbuf[0][SLOT_F4] = f32x4{0.f, 0.f, 0.f, 0.f};      // the other tile of the seed's k block
                amax = 1.0f;
                for (int l = L - 1; l >= 0; --l) {
                    const int nkb = args.nb[l], ng = args.ktp[l] / NTB;
                    const f32x4* dprev = (l > 0) ? wg + (size_t)args.d_off[l - 1] * SLOT_F4 : nullptr;
                    const uint32_t* mkprev = (const uint32_t*)(wg - tid + (size_t)args.d_off[l > 0 ? l - 1 : 0] * SLOT_F4) + tid;
                    const float sigma = gen_pose_scale_grad(gen_pose_max(amax));
                    const float to_true = args.w_inv[l] * gen_pow2_rcp(sigma);
                    amax = 0.f;
                    for (int g0 = 0; g0 < ng; g0 += GEN_PASS_GROUPS) {
                        const int n = (ng - g0 < GEN_PASS_GROUPS) ? ng - g0 : GEN_PASS_GROUPS, t0 = g0 * NTB;
                        switch (n) {
#define PNDF_GEN_BWDS(N) case N: gen_backward_split<N, SP>(ring, wsp, xuni[cur], xbuf[cur ^ 1] + (size_t)t0 * SLOT_F4, dprev ? dprev + (size_t)t0 * SLOT_F4 : nullptr, mkprev + (size_t)g0 * WG_THREADS, args.slope, my_f, nkb, g, t0, gl, sigma, to_true, amax); break;
                            PNDF_GEN_BWDS(1) PNDF_GEN_BWDS(2) PNDF_GEN_BWDS(3) PNDF_GEN_BWDS(4)
#undef PNDF_GEN_BWDS
                            default: break;
                        }
                    }
                    cur ^= 1;
                }
                // (every group has run the whole slot behind it: the last one the slot of padding the host appends -- nothing left to do)
            } else {
            for (int l = L - 1; l >= 0; --l) {
                const int nk = args.nt[l], ng = args.ktp[l] / NTB;
                const f32x4* dprev = (l > 0) ? wg + (size_t)args.d_off[l - 1] * SLOT_F4 : nullptr;
                const uint32_t* mkprev = (const uint32_t*)(wg - tid + (size_t)args.d_off[l > 0 ? l - 1 : 0] * SLOT_F4) + tid;
                for (int g0 = 0; g0 < ng; g0 += GEN_PASS_GROUPS) {
                    const int n = (ng - g0 < GEN_PASS_GROUPS) ? ng - g0 : GEN_PASS_GROUPS, t0 = g0 * NTB;
                    switch (2 * n + half) {
#define PNDF_GEN_BWD(N, H) case 2 * N + H: gen_backward<N, H, SP>(ring, wcur, xuni[cur], xbuf[cur ^ 1] + (size_t)t0 * SLOT_F4, dprev ? dprev + (size_t)t0 * SLOT_F4 : nullptr, mkprev + (size_t)g0 * WG_THREADS, args.slope, my_f, nk, g, t0, gl); break;
                        PNDF_GEN_GROUP_CASES(PNDF_GEN_BWD)
#undef PNDF_GEN_BWD
                        default: break;
                    }
                    half ^= (n * nk) & 1;
                }
                cur ^= 1;
            }
            gen_trunk_end(ring, half);
            }
            __syncthreads();

            // ---------------- encoder backward + normalise backward + update (fp32, as the fused kernels)
            if (args.noenc) ring_skip_encoder_section(ring);
            else encoder_backward<ESP>(my_f, my_gn, eb, ring, ape, g);
            {
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
                for (int j = g; j < NJ; j += 4) {
                    const f32x4 qv = *(const f32x4*)(my_q + 4 * j);
                    const f32x4 gv = *(const f32x4*)(my_gn + 4 * j);
                    f32x4 o;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float dq = gv[c] / denom[c] - qv[c] * kk[c];
                        const float dqs = dq * gscale;
                        o[c] = (args.mode == MODE_PROJECT) ? project_update(qv[c], dval, dqs) : dqs;
                    }
                    *(f32x4*)(my_q + 4 * j) = o;
                }
            }
            __syncthreads();
        }

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
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
}

}  // namespace

extern "C" __global__ void __launch_bounds__(WG_THREADS, 1) pndf_generic_relu_kernel(PndfGenericArgs args) {
    pndf_generic_body<false>(args);
}
extern "C" __global__ void __launch_bounds__(WG_THREADS, 1) pndf_generic_softplus_kernel(PndfGenericArgs args) {
    pndf_generic_body<true>(args);
}
// model.StrEnc.act and model.DFNet.act of different families (net_modules.py:128 vs :30)
extern "C" __global__ void __launch_bounds__(WG_THREADS, 1) pndf_generic_relu_spenc_kernel(PndfGenericArgs args) {
    pndf_generic_body<false, true>(args);
}
extern "C" __global__ void __launch_bounds__(WG_THREADS, 1) pndf_generic_softplus_reluenc_kernel(PndfGenericArgs args) {
    pndf_generic_body<true, false>(args);
}

// precision f16x3 (and f16) on run-time widths: the same four family pairs with the trunk on split-precision fp16 MFMAs
extern "C" __global__ void __launch_bounds__(WG_THREADS, 1) pndf_generic_split_relu_kernel(PndfGenericArgs args) {
    pndf_generic_body<false, false, true>(args);
}
extern "C" __global__ void __launch_bounds__(WG_THREADS, 1) pndf_generic_split_softplus_kernel(PndfGenericArgs args) {
    pndf_generic_body<true, true, true>(args);
}
extern "C" __global__ void __launch_bounds__(WG_THREADS, 1) pndf_generic_split_relu_spenc_kernel(PndfGenericArgs args) {
    pndf_generic_body<false, true, true>(args);
}
extern "C" __global__ void __launch_bounds__(WG_THREADS, 1) pndf_generic_split_softplus_reluenc_kernel(PndfGenericArgs args) {
    pndf_generic_body<true, false, true>(args);
}

// ------------------------------------------------------------------------------------------ host side
using namespace pndf;

struct PndfGeneric {
    pndf_config cfg;
    int L = 0;
    bool enc = true;
    int resident = 0;
    PndfGenericArgs plan;            // tile counts and offsets (pointers filled per launch)
    size_t wf_tiles = 0, wb_tiles = 0, lbias_floats = 0;
    int trunk_tiles = 0;             // wf_tiles + wb_tiles rounded up to whole ring slots
    bool split = false;              // the trunk on split-precision fp16 MFMAs (precision f16x3 / f16) -- decided at pndf_generic_load
    size_t enc_alloc_tiles = 0;      // tiles allocated for the step's stream (the larger of the two plans)
    char* d_enc = nullptr;
    float *d_bias = nullptr, *d_wf = nullptr, *d_wb = nullptr, *d_lb = nullptr, *d_scratch = nullptr;
    bool have_weights = false;
    // the scratch is shared by all launches of the handle: a launch on another stream than the previous one first waits
    // (on the device) for that one's completion event, as the softplus engines do
    hipEvent_t done = nullptr;
    void* last_stream = nullptr;
    bool pending = false;
};

bool pndf_generic_needed(const pndf_config& cfg) {
    // PNDF_FORCE_GENERIC=1: also configs/amass.yaml itself takes this path (tools/bench_generic.py: what the runtime plan costs
    // against the compile-time one, same network, same box)
    const char* force = getenv("PNDF_FORCE_GENERIC");
    if (force && force[0] == '1') return true;
    if (cfg.dims[0] == DIMS[0] && cfg.enc_act != -1 &&
        (cfg.enc_act != cfg.act || (cfg.act == PNDF_ACT_SOFTPLUS && cfg.enc_beta > 0.f && cfg.enc_beta != cfg.beta)))
        return true;      // StrEnc.act / beta differ from DFNet's (net_modules.py:128 vs :30): the fused kernels have one activation family
    if (cfg.n_dims != NLIN + 1) return true;
    for (int i = 1; i < NLIN; ++i)
        if (cfg.dims[i] > DIMS[i]) return true;
    // Softplus networks with a layer of a few units, split precision: behind such a layer a pose's WHOLE gradient is e^(beta z) of a
    // saturated unit (1e-20 and below), and the fused split kernels' per-pose scale stops at 2^40 -- they lose it (measured:
    // 126-256-2-1024-512-2-64-1, 49 of 200 poses beyond 1 % in d d/d q, tools/r6/fused_narrow.py); the runtime-planned split kernels' gradient
    // scale reaches 2^-80 (gen_pose_scale_grad).  With eight units or fewer that can happen; with a whole tile of units it takes every one
    // of them saturated at once.  (precision fp32: the fused exact kernel has fp32's own range.)
    if (cfg.act == PNDF_ACT_SOFTPLUS && cfg.precision != PNDF_PREC_FP32)
        for (int i = 1; i < NLIN; ++i)
            if (cfg.dims[i] <= 8) return true;
    return false;
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int round_up_int(int a, int b) { return ceil_div(a, b) * b; }
constexpr int GEN_STREAM_PAD_SLOTS = 5;      // >= RING_SLOTS - 1: the replica of the stream's first slots behind it (the ring never wraps)
static inline int gen_enc_act(const pndf_config& cfg) { return cfg.enc_act == -1 ? cfg.act : cfg.enc_act; }
static inline float gen_enc_beta(const pndf_config& cfg) { return cfg.enc_beta > 0.f ? cfg.enc_beta : cfg.beta; }
typedef void (*gen_kernel_t)(PndfGenericArgs);
static gen_kernel_t gen_kernel(const pndf_config& cfg, bool split, const char** name) {
    const bool sp = cfg.act == PNDF_ACT_SOFTPLUS, esp = (cfg.dims[0] == DIMS[0]) ? gen_enc_act(cfg) == PNDF_ACT_SOFTPLUS : sp;
    if (split) {
        if (sp && esp) { *name = "pndf_generic_split_softplus_kernel"; return pndf_generic_split_softplus_kernel; }
        if (!sp && !esp) { *name = "pndf_generic_split_relu_kernel"; return pndf_generic_split_relu_kernel; }
        if (sp) { *name = "pndf_generic_split_softplus_reluenc_kernel"; return pndf_generic_split_softplus_reluenc_kernel; }
        *name = "pndf_generic_split_relu_spenc_kernel";
        return pndf_generic_split_relu_spenc_kernel;
    }
    if (sp && esp) { *name = "pndf_generic_softplus_kernel"; return pndf_generic_softplus_kernel; }
    if (!sp && !esp) { *name = "pndf_generic_relu_kernel"; return pndf_generic_relu_kernel; }
    if (sp) { *name = "pndf_generic_softplus_reluenc_kernel"; return pndf_generic_softplus_reluenc_kernel; }
    *name = "pndf_generic_relu_spenc_kernel";
    return pndf_generic_relu_spenc_kernel;
}
// tiles -> tiles rounded up to a group count the layer code exists for (PNDF_GEN_GROUP_CASES)
static inline int gen_round_tiles(int tiles) {
    const int groups = ceil_div(tiles, NTB);
    const int full = (groups - 1) / GEN_PASS_GROUPS * GEN_PASS_GROUPS, rest = groups - full;      // whole passes of 4 groups + a last one
    return (full + rest) * NTB;      // (every 1 .. GEN_PASS_GROUPS groups of a last pass have their instantiation: PNDF_GEN_GROUP_CASES)
}

// tile counts and offsets of the step's stream and of the workgroup's scratch, for the fp32 or the split-precision trunk
static void gen_plan(PndfGeneric* g, bool split) {
    const pndf_config& cfg = g->cfg;
    const int L = g->L;
    PndfGenericArgs& P = g->plan;
    memset(&P, 0, sizeof(P));
    P.nlayers = L;
    P.noenc = g->enc ? 0 : 1;
    int bo = 0, slot = 2 * PNDF_GEN_XTILES;
    for (int l = 0; l < L; ++l) {
        const int in = (l == 0) ? 128 : cfg.dims[l];      // x0 is the pose's 128-row feature buffer (126 | 84 rows used, the rest zero)
        const int outw = cfg.dims[l + 1];
        P.kt[l] = ceil_div(in, 16);
        P.nt[l] = ceil_div(outw, 16);
        P.kb[l] = ceil_div(P.kt[l], 2);
        P.nb[l] = ceil_div(P.nt[l], 2);
        P.ktp[l] = gen_round_tiles(P.kt[l]);
        P.ntp[l] = gen_round_tiles(P.nt[l]);
        P.w_inv[l] = 1.0f;
        P.b_off[l] = bo;
        P.d_off[l] = slot;
        bo += 16 * P.ntp[l];
        if (l < L - 1) slot += P.ntp[l];
    }
    // stream tiles of a layer: fp32 one tile per (output tile, k tile); split one PAIR (hi, lo) per (output tile, k block of 32)
    int wf = 0, wb = 0;
    for (int l = 0; l < L; ++l) {
        P.wf_off[l] = wf;
        wf += split ? 2 * P.ntp[l] * P.kb[l] : P.ntp[l] * P.kt[l];
    }
    g->wf_tiles = wf;                 // the backward matrices follow the forward ones in ONE stream, in the order the backward pass walks them
    for (int l = L - 1; l >= 0; --l) {
        P.wb_off[l] = wf + wb;
        wb += split ? 2 * P.ktp[l] * P.nb[l] : P.ktp[l] * P.nt[l];
    }
    g->wb_tiles = wb;
    P.enc_d_off = slot;
    if (gen_enc_act(cfg) == PNDF_ACT_SOFTPLUS && g->enc) slot += 2 * NJ;
    P.wg_tiles = slot;
    // the step's stream: encoder forward (3 slots) | trunk (whole groups of NTB tiles, padded to whole slots) | encoder backward (3 slots)
    // Every group of the trunk runs the ring events and the tile reads of the group BEHIND it (gen_layer has no "last group" case), so
    // the last one needs padding behind it: fp32 the other half of its slot, or -- when the trunk fills its last slot -- a slot of its
    // own; split (a group = a slot) always a slot.
    g->trunk_tiles = split ? (wf + wb) + SLOT_TILES : round_up_int((wf + wb) + NTB, SLOT_TILES);
    P.w_slots = (2 * ENC_TILES_PADDED + g->trunk_tiles) / SLOT_TILES;
    g->lbias_floats = bo;
    g->split = split;
}

int pndf_generic_create(PndfGeneric** out, const pndf_config& cfg, int resident_wgs, std::string& err) {
    *out = nullptr;
    const int L = cfg.n_dims - 1;
    if (L < 2 || L > PNDF_GEN_MAXLIN) { err = "DFNet depth: 2 .. 8 linear layers (n_dims 3 .. 9)"; return PNDF_ERR_UNSUPPORTED; }
    PndfGeneric* g = new PndfGeneric();
    g->cfg = cfg;
    g->L = L;
    g->enc = cfg.dims[0] == DIMS[0];
    g->resident = resident_wgs;
    // the stream is allocated for the larger of the two plans: pndf_generic_load falls back to the fp32 trunk when a layer cannot be
    // scaled into the fp16 range (scratch and biases have the same layout in both)
    gen_plan(g, false);
    size_t alloc_slots = (size_t)g->plan.w_slots;
    if (cfg.precision != PNDF_PREC_FP32) {
        gen_plan(g, true);
        if ((size_t)g->plan.w_slots > alloc_slots) alloc_slots = (size_t)g->plan.w_slots;
    }
    g->enc_alloc_tiles = (alloc_slots + GEN_STREAM_PAD_SLOTS) * SLOT_TILES;
    PndfGenericArgs& P = g->plan;
    hipError_t e = hipMalloc((void**)&g->d_bias, BIAS_FLOATS * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void**)&g->d_enc, g->enc_alloc_tiles * TILE_BYTES);
    g->d_wf = g->d_wb = nullptr;      // (one stream: encoder and trunk tiles live in d_enc)
    if (e == hipSuccess) e = hipMalloc((void**)&g->d_lb, g->lbias_floats * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void**)&g->d_scratch, (size_t)resident_wgs * P.wg_tiles * SLOT_F4 * sizeof(f32x4));
    if (e == hipSuccess) e = hipEventCreateWithFlags(&g->done, hipEventDisableTiming);
    for (const void* kfn : {(const void*)pndf_generic_relu_kernel, (const void*)pndf_generic_softplus_kernel,
                            (const void*)pndf_generic_relu_spenc_kernel, (const void*)pndf_generic_softplus_reluenc_kernel,
                            (const void*)pndf_generic_split_relu_kernel, (const void*)pndf_generic_split_softplus_kernel,
                            (const void*)pndf_generic_split_relu_spenc_kernel, (const void*)pndf_generic_split_softplus_reluenc_kernel})
        if (e == hipSuccess) e = hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
    if (e != hipSuccess) {
        err = std::string("pndf_create (runtime-planned DFNet): ") + hipGetErrorString(e);
        pndf_generic_destroy(g);
        return PNDF_ERR_HIP;
    }
    *out = g;
    return PNDF_OK;
}

void pndf_generic_destroy(PndfGeneric* g) {
    if (!g) return;
    if (g->done) (void)hipEventDestroy(g->done);
    for (void* p : {(void*)g->d_enc, (void*)g->d_bias, (void*)g->d_wf, (void*)g->d_wb, (void*)g->d_lb, (void*)g->d_scratch})
        if (p) (void)hipFree(p);
    delete g;
}

int pndf_generic_load(PndfGeneric* g, const float* const* tensors, const int64_t* numel, int n_tensors, std::string& err) {
    const int L = g->L;
    const int want = (g->enc ? 4 * NJ : 0) + 2 * L;
    if (!tensors || !numel || n_tensors != want) {
        err = "expected " + std::to_string(want) + " tensors in state-dict order (encoder: 84, then weight and bias of every dfnet.lin)";
        return PNDF_ERR_BAD_SHAPE;
    }
    int t = 0;
    for (int j = 0; g->enc && j < NJ; ++j) {
        const int64_t sz[4] = {HID * enc_in(j), HID, FEAT * HID, FEAT};
        for (int k = 0; k < 4; ++k, ++t)
            if (!tensors[t] || numel[t] != sz[k]) { err = "encoder tensor missing or of the wrong size"; return PNDF_ERR_BAD_SHAPE; }
    }
    const float* const* lin = tensors + t;
    const int64_t* ln = numel + t;
    for (int l = 0; l < L; ++l) {
        const int64_t in = g->cfg.dims[l], outw = g->cfg.dims[l + 1];
        if (!lin[2 * l] || !lin[2 * l + 1] || ln[2 * l] != in * outw || ln[2 * l + 1] != outw) {
            err = "dfnet.lin" + std::to_string(l) + " does not match the configured dims";
            return PNDF_ERR_BAD_SHAPE;
        }
    }
    // precision f16x3 / f16: the trunk on split-precision fp16 MFMAs -- the weights of layer l travel as s_l W, s_l = the power of two
    // that brings the layer's largest |weight| into [2^12, 2^13) (pndf_capi.hip "split-precision stream").  A layer without a finite
    // non-zero weight cannot be scaled: such a network runs the exact fp32 kernels, whatever precision was asked for.
    float wscale[PNDF_GEN_MAXLIN];
    bool split = g->cfg.precision != PNDF_PREC_FP32;
    for (int l = 0; l < L && split; ++l) {
        float mx = 0.f;
        bool nan = false;
        const int64_t n = (int64_t)g->cfg.dims[l] * g->cfg.dims[l + 1];
        for (int64_t i = 0; i < n; ++i) {
            const float a = fabsf(lin[2 * l][i]);
            nan |= (a != a);
            if (a > mx) mx = a;
        }
        if (nan || !(mx > 0x1p-100f && mx < 0x1p100f)) { split = false; break; }
        int e;
        (void)frexpf(mx, &e);                            // mx = f * 2^e, f in [0.5, 1)
        wscale[l] = ldexpf(1.0f, 13 - e);                // s_l * mx in [2^12, 2^13)
    }
    gen_plan(g, split);
    PndfGenericArgs& P = g->plan;
    const size_t step_tiles = (size_t)P.w_slots * SLOT_TILES;
    if (step_tiles + (size_t)GEN_STREAM_PAD_SLOTS * SLOT_TILES > g->enc_alloc_tiles) { err = "internal: stream larger than its allocation"; return PNDF_ERR_BAD_SHAPE; }
    std::vector<float> stream((step_tiles + (size_t)GEN_STREAM_PAD_SLOTS * SLOT_TILES) * TILE_FLOATS, 0.f), lb(g->lbias_floats, 0.f), bias(BIAS_FLOATS, 0.f);
    float* const trunk = stream.data() + (size_t)ENC_TILES_PADDED * TILE_FLOATS;      // wf_off / wb_off count tiles from here
    const bool sp_trunk = g->cfg.act == PNDF_ACT_SOFTPLUS;
    for (int l = 0; l < L; ++l) {
        const int in = g->cfg.dims[l], outw = g->cfg.dims[l + 1];
        const pndf_pack::Mat F{lin[2 * l], outw, in, false}, T{lin[2 * l], outw, in, true};
        float* dst = trunk + (size_t)P.wf_off[l] * TILE_FLOATS;
        const int PT = GEN_PASS_GROUPS * NTB;                  // consumption order: [pass of <= 32 output tiles][k tile][tile of the pass]
        if (split) {                                           // ... [pass][k block][tile of the pass] pairs (hi tile, lo tile)
            P.w_inv[l] = 1.0f / wscale[l];
            for (int t0 = 0; t0 < P.ntp[l]; t0 += PT)
                for (int k = 0; k < P.kb[l]; ++k)
                    for (int t = t0; t < P.ntp[l] && t < t0 + PT; ++t, dst += 2 * TILE_FLOATS) pndf_pack::emit_pair_f16(F, t, k, wscale[l], dst);
            dst = trunk + (size_t)P.wb_off[l] * TILE_FLOATS;
            for (int t0 = 0; t0 < P.ktp[l]; t0 += PT)
                for (int k = 0; k < P.nb[l]; ++k)
                    for (int t = t0; t < P.ktp[l] && t < t0 + PT; ++t, dst += 2 * TILE_FLOATS) pndf_pack::emit_pair_f16(T, t, k, wscale[l], dst);
        } else {
            for (int t0 = 0; t0 < P.ntp[l]; t0 += PT)
                for (int k = 0; k < P.kt[l]; ++k)
                    for (int t = t0; t < P.ntp[l] && t < t0 + PT; ++t, dst += TILE_FLOATS) pndf_pack::emit_tile(F, t, k, dst);
            dst = trunk + (size_t)P.wb_off[l] * TILE_FLOATS;
            for (int t0 = 0; t0 < P.ktp[l]; t0 += PT)
                for (int k = 0; k < P.nt[l]; ++k)
                    for (int t = t0; t < P.ktp[l] && t < t0 + PT; ++t, dst += TILE_FLOATS) pndf_pack::emit_tile(T, t, k, dst);
        }
        memcpy(lb.data() + P.b_off[l], lin[2 * l + 1], sizeof(float) * outw);
        // split path, Softplus: the zero-padded units of a layer would enter the per-pose operand bound it measures (softplus(0) =
        // ln 2 / beta, derivative 1 / 2); a bias of -1e6 makes both exactly zero (pndf_capi.hip does the same for narrower networks)
        if (split && sp_trunk && l < L - 1)
            for (int j = outw; j < 16 * P.ntp[l]; ++j) lb[P.b_off[l] + j] = -1.0e6f;
    }
    for (int l = 0; l < 8; ++l) bias[SCALE_OFF + l] = 1.0f;
    if (g->enc) {
        for (int j = 0; j < NJ; ++j) {
            memcpy(bias.data() + ENCB_OFF + 32 * j, tensors[4 * j + 1], sizeof(float) * HID);
            memcpy(bias.data() + ENCB_OFF + 32 * j + 16 + ENC_FEAT_ROW, tensors[4 * j + 3], sizeof(float) * FEAT);
        }
        pndf_pack::emit_encoder_sections(tensors, stream.data(), stream.data() + ((size_t)ENC_TILES_PADDED + g->trunk_tiles) * TILE_FLOATS);
    }
    // replica of the first slots behind the stream: the ring's fetch offset never wraps inside a step
    memcpy(stream.data() + step_tiles * TILE_FLOATS, stream.data(), (size_t)GEN_STREAM_PAD_SLOTS * SLOT_TILES * TILE_FLOATS * sizeof(float));
    hipError_t e = hipDeviceSynchronize();      // no launch may still be reading the old weights
    if (e == hipSuccess) e = hipMemcpy(g->d_lb, lb.data(), lb.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(g->d_bias, bias.data(), bias.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(g->d_enc, stream.data(), stream.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) { err = std::string("pndf_load_weights (runtime-planned DFNet): ") + hipGetErrorString(e); return PNDF_ERR_HIP; }
    g->have_weights = true;
    return PNDF_OK;
}

const char* pndf_generic_kernel_name(const PndfGeneric* g) {
    const char* name = "";
    (void)gen_kernel(g->cfg, g->split, &name);
    return name;
}

int pndf_generic_launch(PndfGeneric* g, int mode, const float* q, const float* gout, float* qo, float* d, int64_t B, int steps,
                        void* stream, std::string& err) {
    if (!g->have_weights) { err = "pndf_load_weights has not been called"; return PNDF_ERR_NO_WEIGHTS; }
    PndfGenericArgs a = g->plan;
    a.q_in = q; a.q_out = qo; a.d_out = d; a.grad_out = gout;
    a.enc_stream = g->d_enc; a.bias = g->d_bias; a.wfwd = nullptr; a.wbwd = nullptr; a.lbias = g->d_lb; a.scratch = g->d_scratch;
    a.B = B; a.steps = steps; a.mode = mode;
    a.slope = (g->cfg.act == PNDF_ACT_LRELU) ? 0.01f : 0.0f;      // nn.LeakyReLU() default slope, net_modules.py:31
    a.beta = g->cfg.beta;
    a.enc_slope = (gen_enc_act(g->cfg) == PNDF_ACT_LRELU) ? 0.01f : 0.0f;
    a.enc_beta = gen_enc_beta(g->cfg);
    const int64_t nblocks = (B + WG_POSES - 1) / WG_POSES;
    const dim3 grid((unsigned)(nblocks < g->resident ? nblocks : g->resident)), block(WG_THREADS);
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing((hipStream_t)stream, &cap);
    const bool capturing = cap != hipStreamCaptureStatusNone;
    hipError_t e = hipSuccess;
    if (g->pending && g->last_stream != stream && !capturing) e = hipStreamWaitEvent((hipStream_t)stream, g->done, 0);
    if (e == hipSuccess) {
        const char* name = "";
        hipLaunchKernelGGL(gen_kernel(g->cfg, g->split, &name), grid, block, LDS_TOTAL, (hipStream_t)stream, a);
        e = hipGetLastError();
    }
    if (e == hipSuccess && !capturing) {
        e = hipEventRecord(g->done, (hipStream_t)stream);
        g->last_stream = stream;
        g->pending = true;
    }
    if (e != hipSuccess) { err = std::string("runtime-planned DFNet launch: ") + hipGetErrorString(e); return PNDF_ERR_HIP; }
    return PNDF_OK;
}

PNDF_EXPORT_EXPERIMENT_WORD(generic)
