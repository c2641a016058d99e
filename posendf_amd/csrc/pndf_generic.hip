// Runtime-planned Pose-NDF kernel for gfx950: any DFNet the reference can build.
//
// reference model/network/net_modules.py:14-28 takes the hidden widths of DFNet as a free list (`dims`), and the README points
// users at checkpoints of configs other than configs/amass.yaml.  The fused kernels of pndf_kernel*.hip are compile-time plans of
// that one architecture (pair-fused phases, register-resident activations); everything else -- 2 .. 8 linear layers, hidden widths
// 1 .. 1024 -- runs here, with the same ownership (a workgroup = 4 waves = 64 poses for the whole call, one persistent launch for
// all projection steps), the same MFMA encoder, normalisation and update code (pndf_device.h), and the trunk LAYER BY LAYER from a
// plan that travels in the kernel arguments:
//   * arithmetic: exact fp32 (v_mfma_f32_16x16x4_f32, fp32 accumulate from the bias), the same operation order as
//     pndf_fused_relu_kernel -- whatever `precision` the handle asked for (a request for speed, not for less accuracy);
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
//   * weights: fp32 tiles `tile(M, nt, kt)` in consumption order [pass][k tile][tile of the pass], streamed global -> LDS by DMA into a
//     five-slot ring PRIVATE to each wave, four slots (2,048 cycles of matrix pipe) ahead, counted vmcnt waits, no barrier in the trunk
//     (see "the trunk's weight ring" below for the forms that were measured before it).
// Roofline: fp32 MFMA (157.3 TFLOP/s); algorithmic work per pose-step 4 x sum_l in_l out_l FLOP.  Not the benchmark path
// (BASELINE.json names amass.yaml).  Measured (tools/bench_generic.py, B = 65,536 x 10 steps, profiles/r06/generic_arch*.jsonl), as a
// fraction of the fp32 MFMA peak on configs/amass.yaml itself (PNDF_FORCE_GENERIC=1; the fused exact-fp32 kernel: 0.89):
//   v1 0.22  one block of four output tiles per pass over the operand, weights and operand prefetched one k step through registers
//   v2 0.45  the accumulators of a pass (32 tiles) resident: the operand is read once per pass
//   v3 0.47  weight groups of eight tiles (1,024 cycles of look-ahead)         -> rocprofv3: 48 % of the wave cycles at a waitcnt, L2 hit 76 %
//   v4 0.41  a group shared by the four waves through LDS, one barrier per group (the barrier hands every wave the slowest wave's miss)
//   v5 0.60  a five-slot weight ring PRIVATE to each wave, fed by LDS-DMA four slots ahead, no barrier in the trunk
//   v6 0.68  the slot laid out by hand: DMA pieces and the next slot's tile reads between the rounds of MFMAs (this file; 0.72 on a
//            wider network, 0.63 with Softplus)
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

// ---- the trunk's weight ring: PRIVATE to each wave, fed by LDS-DMA
// The rocprofv3 counters of the register-prefetch form (profiles/r06/generic/): 47.6 % MFMA busy, 47.7 % of the wave cycles at an
// s_waitcnt, L2 hit rate 76 % -- 22 MB of fp32 weights per step do not fit 4 MB of L2, workgroups that wait drift apart, and with one
// group (1,024 cycles) of look-ahead every miss is a stall.  Sharing a group among the four waves through LDS with a barrier per group
// is slower still (0.41: the barrier hands every wave the slowest wave's miss).  What the fused kernels have is DISTANCE without
// registers: global_load_lds_dwordx4 moves a tile global -> LDS with no destination register, so any number of slots can be in flight;
// here every wave runs its own five-slot ring (4 tiles = 16 MFMAs per slot, fetched FOUR slots = 2,048 cycles ahead, counted vmcnt
// waits, no barrier anywhere: the waves of a workgroup share nothing in the trunk), and the operand tile of the next k step comes the
// same way.  The encoder's ring buffers (80 KiB) and the chunk-mask rows (unused here) are idle during the trunk.
constexpr int GW_SLOT_TILES = 4;
constexpr int GW_SLOT_BYTES = GW_SLOT_TILES * TILE_BYTES;          // 4 KiB
constexpr int GW_RING = 5;
constexpr int GW_AHEAD = GW_RING - 1;
constexpr int GW_WAVE_BYTES = GW_RING * GW_SLOT_BYTES;             // 20 KiB per wave
constexpr int GX_WAVE_BYTES = 2 * TILE_BYTES;                      // operand tile, double buffered
static_assert(4 * GW_WAVE_BYTES <= RING_SLOTS * SLOT_BYTES, "four private rings live in the encoder's ring buffers");
static_assert(4 * GX_WAVE_BYTES <= MASK_ROWS * WG_THREADS, "the operand buffers live in the chunk-mask rows");
static_assert(NTB % GW_SLOT_TILES == 0, "a group of output tiles is whole slots");

struct GenLds {
    uint32_t w_lds;        // uniform: LDS address of this wave's ring
    uint32_t x_lds;        // uniform: LDS address of this wave's two operand tiles
    const char* w_ptr;     // per lane: its 16 bytes of tile 0 of ring slot 0
    const char* x_ptr;     // per lane: its 16 bytes of operand buffer 0
    uint32_t lane16;
};

// one ring slot (4 tiles): M0 = LDS destination, the instruction offset moves both addresses (pndf_device.h: ring_dma_piece)
__device__ __forceinline__ void gw_dma_slot(const char* base, uint32_t voff, uint32_t dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %0, %1\n\t"
                 "global_load_lds_dwordx4 %0, %1 offset:1024\n\t"
                 "global_load_lds_dwordx4 %0, %1 offset:2048\n\t"
                 "global_load_lds_dwordx4 %0, %1 offset:3072"
                 : : "v"(voff), "s"(base), "s"(dst) : "memory", "m0");
}
// the same slot piece by piece, one piece behind each round of four MFMAs (pieces 1 .. 3 rely on M0 as piece 0 left it: nothing
// else in the trunk writes M0 between them -- the operand tile's DMA is issued at the top of a k step, outside a slot)
template <int PIECE>
__device__ __forceinline__ void gw_dma_piece(const char* base, uint32_t voff, uint32_t dst) {
    if constexpr (PIECE == 0)
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(base), "s"(dst) : "memory", "m0");
    else if constexpr (PIECE == 1)
        asm volatile("global_load_lds_dwordx4 %0, %1 offset:1024" : : "v"(voff), "s"(base) : "memory");
    else if constexpr (PIECE == 2)
        asm volatile("global_load_lds_dwordx4 %0, %1 offset:2048" : : "v"(voff), "s"(base) : "memory");
    else
        asm volatile("global_load_lds_dwordx4 %0, %1 offset:3072" : : "v"(voff), "s"(base) : "memory");
}
__device__ __forceinline__ void gw_dma_tile(const char* base, uint32_t voff, uint32_t dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(base), "s"(dst) : "memory", "m0");
}

// The weight stream of a step is ONE sequence of slots in consumption order -- forward passes layer by layer, then the transposed
// matrices in the backward pass's order -- so the ring is started once per step and runs through every pass: a pass ends with the
// first slots of the next one already in flight (with a ring per pass the 16 passes of a step each began on an empty ring).
struct GenRing {
    uint32_t fetch, last;        // per lane: byte offset of the next slot to fetch / of the stream's last slot (+ lane * 16)
    uint32_t fbuf, rbuf;         // uniform: ring buffer the next fetch goes to / the next read comes from
    f32x4 wt[2][GW_SLOT_TILES];  // the tiles of the current slot (set 0 at every pass boundary: a pass is an even number of slots)
};
__device__ __forceinline__ void gen_ring_start(GenRing& R, const char* wbase, int total_slots, const GenLds& L) {
    R.last = (uint32_t)(total_slots - 1) * GW_SLOT_BYTES + L.lane16;      // (fetches past the end re-read the last slot: no branch)
    R.fetch = L.lane16;
    R.fbuf = 0;
#pragma unroll
    for (int i = 0; i < GW_AHEAD; ++i) {
        gw_dma_slot(wbase, R.fetch < R.last ? R.fetch : R.last, L.w_lds + R.fbuf);
        R.fetch += GW_SLOT_BYTES;
        R.fbuf += GW_SLOT_BYTES;
    }
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(4 * (GW_AHEAD - 1)) : "memory");
#pragma unroll
    for (int j = 0; j < GW_SLOT_TILES; ++j) R.wt[0][j] = *(const f32x4*)(L.w_ptr + j * TILE_BYTES);
    R.rbuf = GW_SLOT_BYTES;
}

// acc[t] += sum_k W(t, k) X[k] for `NG * NTB` output tiles (one pass over a layer) at once: the next nk * SPK slots of the ring.
// `wbase` (uniform) = the stream's first tile; `xbase` (uniform) = this wave's 1 KiB of operand tile 0 (tiles 4 KiB apart).  Register
// indices must be compile-time, so the number of groups is a template parameter: the plan rounds a layer's output tiles up to whole
// groups (host side: gen_round_tiles) and the layer code is instantiated per group count of a pass.  (A first form kept 64
// accumulators under run-time guards `if (group < ng)`: hipcc answered with 700 spilled registers.)
// The tiles of a slot are read from LDS one slot ahead of their MFMAs, into one of two register sets that alternate by slot parity
// (no copies).  One wave per SIMD overlaps its own non-MFMA instructions with its own MFMAs only when they sit between them in
// program order (DESIGN.md section 2 "epilogue in MFMA slots"), so a slot is laid out by hand and pinned:
//   M M M M  P0 | M M M M  P1 | M M M M  P2 P3  wait(slot t + 1)  R R R R  [X] | M M M M
// P = one 1-KiB piece of the fetch of slot t + 4, R = the four tile reads of slot t + 1, X (last slot of a k step) = the read of the
// NEXT k step's operand tile: their latency hides behind the last round.
// vmcnt bookkeeping (loads retire in order; other operations of the wave in the queue only make a counted wait stricter):
//   slot t + 1 is read after DMA(t + 4) has been issued: three younger slots x 4 pieces -> vmcnt(12).  Operand tile k + 1 is issued
//   at the top of k step k and read in its last slot: the SPK slots issued in between are younger -> vmcnt(min(4 SPK, 12)).  Operand
//   tile 0 of a pass is the youngest operation when it is needed: vmcnt(0), once per pass (the ring's slots land meanwhile).
template <int NG>
__device__ __forceinline__ void gen_layer(GenRing& R, const char* wbase, const char* xbase, int nk, f32x4 (&acc)[NG * NTB], const GenLds& L) {
    constexpr int SPK = NG * NTB / GW_SLOT_TILES;                  // slots per k step: even, so the two register sets alternate by si
    static_assert(SPK % 2 == 0, "the tile registers of consecutive slots alternate by slot parity");
    constexpr int XWAIT = 4 * SPK < 4 * (GW_AHEAD - 1) ? 4 * SPK : 4 * (GW_AHEAD - 1);
    gw_dma_tile(xbase, L.lane16, L.x_lds);
    if (nk > 1) gw_dma_tile(xbase, SLOT_F4 * 16u + L.lane16, L.x_lds + TILE_BYTES);
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(0) : "memory");
    f32x4 xc = *(const f32x4*)L.x_ptr, xn = xc;
    for (int k = 0; k < nk; ++k) {
        // operand tile k + 2 -> the buffer tile k was read from (k >= 1: at the end of k step k - 1; k = 0: just now)
        if (k >= 1) gw_dma_tile(xbase, (uint32_t)((k + 1 < nk) ? k + 1 : k) * (SLOT_F4 * 16u) + L.lane16, L.x_lds + ((k + 1) & 1) * TILE_BYTES);
#pragma unroll
        for (int si = 0; si < SPK; ++si) {
            const f32x4 (&wc)[GW_SLOT_TILES] = R.wt[si & 1];
            f32x4 (&wn)[GW_SLOT_TILES] = R.wt[(si + 1) & 1];
            const uint32_t voff = R.fetch < R.last ? R.fetch : R.last;
            const uint32_t dst = L.w_lds + R.fbuf;                 // the buffer slot t - 1 was read from, a slot ago
            R.fetch += GW_SLOT_BYTES;
            R.fbuf = (R.fbuf == (GW_RING - 1) * GW_SLOT_BYTES) ? 0u : R.fbuf + GW_SLOT_BYTES;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < GW_SLOT_TILES; ++j)
                    acc[si * GW_SLOT_TILES + j] = mfma4(wc[j][s], xc[s], acc[si * GW_SLOT_TILES + j]);      // four independent chains
                __builtin_amdgcn_sched_barrier(0);
                if (s == 0) gw_dma_piece<0>(wbase, voff, dst);
                if (s == 1) gw_dma_piece<1>(wbase, voff, dst);
                if (s == 2) {
                    gw_dma_piece<2>(wbase, voff, dst);
                    gw_dma_piece<3>(wbase, voff, dst);
                    if (si == SPK - 1) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(XWAIT) : "memory");                 // slot t + 1 AND operand tile k + 1
                    else asm volatile("s_waitcnt vmcnt(%0)" : : "n"(4 * (GW_AHEAD - 1)) : "memory");                 // slot t + 1 has landed
#pragma unroll
                    for (int j = 0; j < GW_SLOT_TILES; ++j) wn[j] = *(const f32x4*)(L.w_ptr + R.rbuf + j * TILE_BYTES);
                    R.rbuf = (R.rbuf == (GW_RING - 1) * GW_SLOT_BYTES) ? 0u : R.rbuf + GW_SLOT_BYTES;
                    if (si == SPK - 1) xn = *(const f32x4*)(L.x_ptr + ((k + 1) & 1) * TILE_BYTES);
                }
            }
        }
        xc = xn;
    }
    __builtin_amdgcn_sched_barrier(0);
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

// one forward layer with NG groups of output tiles: bias -> accumulate -> (hidden layers) activation, derivative factor
template <int NG, bool SP>
__device__ __forceinline__ void gen_forward(const char* w, const float* bias, const char* xin, f32x4* xout, f32x4* dl, int nk, bool last,
                                            float slope, const SpK& k, int g, f32x4& zlast, const GenLds& L, GenRing& R, const char* wbase) {
    f32x4 acc[NG * NTB];
#pragma unroll
    for (int t = 0; t < NG * NTB; ++t) acc[t] = *(const f32x4*)(bias + 16 * t + 4 * g);
    // the bias loads are consumed HERE: left pending, hipcc waits for them with `s_waitcnt vmcnt(0)` at the head of the k loop --
    // in every iteration, which drains the weight ring's look-ahead once per k step
#pragma unroll
    for (int t = 0; t < NG * NTB; ++t) asm volatile("" : "+v"(acc[t]));
    (void)w;
    gen_layer<NG>(R, wbase, xin, nk, acc, L);
    if (last) {                    // the output layer: one unit, row 0 of tile 0; its activation is the caller's
        zlast = acc[0];
        return;
    }
#pragma unroll
    for (int t = 0; t < NG * NTB; ++t) {
        f32x4 df;
        gen_act<SP>(acc[t], df, slope, k);
        xout[(size_t)t * SLOT_F4] = acc[t];
        dl[(size_t)t * SLOT_F4] = df;
        if (t % NTB == NTB - 1) __builtin_amdgcn_sched_barrier(0);      // a group at a time: the accumulators fill up to half the register file
    }
}

// one backward layer: G_in = W^T G_out, times the derivative factors of the layer below (l > 0) or into the pose's feature row
template <int NG>
__device__ __forceinline__ void gen_backward(const char* w, const char* gin, f32x4* gout, const f32x4* dprev, float* my_f, int nk, int g, int t0,
                                             const GenLds& L, GenRing& R, const char* wbase) {
    (void)w;
    f32x4 acc[NG * NTB];
#pragma unroll
    for (int t = 0; t < NG * NTB; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    gen_layer<NG>(R, wbase, gin, nk, acc, L);
#pragma unroll
    for (int t = 0; t < NG * NTB; ++t) {
        if (dprev) gout[(size_t)t * SLOT_F4] = acc[t] * dprev[(size_t)t * SLOT_F4];      // x act'(z_{l-1})
        else if (t0 + t < 8) *(f32x4*)(my_f + 16 * (t0 + t) + 4 * g) = acc[t];           // d z_out / d x0 (t0 = 0: one pass)
        // (without it hipcc hoists all derivative loads above the first multiply: up to 256 more live registers)
        if (t % NTB == NTB - 1) __builtin_amdgcn_sched_barrier(0);
    }
}

// the group counts the layer code is instantiated for (the plan rounds up to the next one: at most a third of a layer is padding)
// A pass keeps at most 4 groups = 32 tiles = 128 accumulator registers (a whole 1024-wide layer at once -- 256 -- left hipcc 150 - 220
// spilled registers: everything that is not an MFMA accumulator has to fit the 256 architectural VGPRs); wider layers take two
// passes, each of which reads the operand tiles once.
#define PNDF_GEN_GROUP_CASES(X) X(1) X(2) X(3) X(4)
constexpr int GEN_PASS_GROUPS = 4;

// SP: the trunk's activation is Softplus (else relu / lrelu); ESP: the encoder's.  Every config of the reference has ESP == SP;
// net_modules.py:128 reads model.StrEnc.act on its own, so the other two combinations exist as well.
template <bool SP, bool ESP = SP>
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
    gl.w_lds = (uint32_t)(size_t)(PNDF_LDS char*)(smem + LDS_RING) + wave * GW_WAVE_BYTES;
    gl.x_lds = (uint32_t)(size_t)(PNDF_LDS char*)(smem + LDS_MASK) + wave * GX_WAVE_BYTES;
    gl.w_ptr = smem + LDS_RING + wave * GW_WAVE_BYTES + lane * 16;
    gl.x_ptr = smem + LDS_MASK + wave * GX_WAVE_BYTES + lane * 16;
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
    ring.smem = smem;
    ring.lane = lane;
    ring.st_wait = ring.st_bar = 0;
    ring.st_n = 0;
    for (int i = tid; i < BIAS_FLOATS / 4; i += WG_THREADS) ((f32x4*)lds_bias)[i] = ((const f32x4*)args.bias)[i];

    const long long nblocks = (args.B + WG_POSES - 1) / WG_POSES;
    for (long long blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
        const long long pose0 = blk * WG_POSES;
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
        __syncthreads();

        const int nsteps = (args.mode == MODE_PROJECT) ? args.steps : 1;
        float dval = 0.f;
        for (int step = 0; step < nsteps; ++step) {
            // ---------------- encoder (or the normalised pose itself) -> x0, 128 rows in the pose's feature row
            uint32_t eb[6] = {0, 0, 0, 0, 0, 0};
            float poison = 0.f;
            if (args.noenc) {
                poison = noenc_forward<SP>(my_q, my_f, g);
            } else {
                if constexpr (SP && !ESP) {      // a Softplus trunk behind a relu-family encoder: the NaN / inf poison of the pose (joint_axis_norms)
                    float ss[4];
                    poison = joint_axis_norms<true>(my_q, ss);
                }
                // the encoder's tiles come through the weight ring like in the fused kernels: forward section of the stream
                ring.gstream = args.enc_stream;
                ring_start(ring, wave);
                ring_wait_dma();
                __syncthreads();
                const float pe = encoder_forward<ESP>(my_q, my_f, lds_bias + ENCB_OFF, eb, ring, ape, g);
                if constexpr (ESP) poison = pe;
                ring_wait_dma();      // (the ring fetched ahead into the section's padding: drain before the next restart)
                __syncthreads();      // ... by EVERY wave: the trunk's private weight rings reuse the ring's buffers
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) xbuf[0][(size_t)t * SLOT_F4] = *(const f32x4*)(my_f + 16 * t + 4 * g);

            // ---------------- trunk forward, layer by layer (net_modules.py:51-69); the weight ring starts here and runs to the end of the
            // backward pass (forward-only calls leave it after the forward half: drained below)
            GenRing wring;
            gen_ring_start(wring, (const char*)args.wfwd, args.w_slots, gl);
            f32x4 zlast = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int l = 0; l < L; ++l) {
                const int nk = args.kt[l], ng = args.ntp[l] / NTB;
                const char* w = (const char*)args.wfwd + (size_t)args.wf_off[l] * TILE_BYTES;
                const float* bias = args.lbias + args.b_off[l];
                f32x4* dl = wg + (size_t)args.d_off[l] * SLOT_F4;
                for (int g0 = 0; g0 < ng; g0 += GEN_PASS_GROUPS) {      // passes of at most 8 groups of output tiles
                    const int n = (ng - g0 < GEN_PASS_GROUPS) ? ng - g0 : GEN_PASS_GROUPS, t0 = g0 * NTB;
                    const char* wp = w + (size_t)g0 * nk * NTB * TILE_BYTES;      // stream order: [pass][k tile][tile of the pass]
                    switch (n) {
#define PNDF_GEN_FWD(N) case N: gen_forward<N, SP>(wp, bias + 16 * t0, xuni[l & 1], xbuf[(l + 1) & 1] + (size_t)t0 * SLOT_F4, dl + (size_t)t0 * SLOT_F4, nk, l == L - 1, args.slope, ap.k, g, zlast, gl, wring, (const char*)args.wfwd); break;
                        PNDF_GEN_GROUP_CASES(PNDF_GEN_FWD)
#undef PNDF_GEN_FWD
                        default: break;      // (pndf_generic_create plans no other group count)
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
            if (args.mode == MODE_FORWARD) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the ring's look-ahead into the backward half)
                break;
            }
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
            for (int l = L - 1; l >= 0; --l) {
                const int nk = args.nt[l], ng = args.ktp[l] / NTB;
                const char* w = (const char*)args.wbwd + (size_t)args.wb_off[l] * TILE_BYTES;
                const f32x4* dprev = (l > 0) ? wg + (size_t)args.d_off[l - 1] * SLOT_F4 : nullptr;
                for (int g0 = 0; g0 < ng; g0 += GEN_PASS_GROUPS) {
                    const int n = (ng - g0 < GEN_PASS_GROUPS) ? ng - g0 : GEN_PASS_GROUPS, t0 = g0 * NTB;
                    const char* wp = w + (size_t)g0 * nk * NTB * TILE_BYTES;
                    switch (n) {
#define PNDF_GEN_BWD(N) case N: gen_backward<N>(wp, xuni[cur], xbuf[cur ^ 1] + (size_t)t0 * SLOT_F4, dprev ? dprev + (size_t)t0 * SLOT_F4 : nullptr, my_f, nk, g, t0, gl, wring, (const char*)args.wfwd); break;
                        PNDF_GEN_GROUP_CASES(PNDF_GEN_BWD)
#undef PNDF_GEN_BWD
                        default: break;
                    }
                }
                cur ^= 1;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the ring's re-reads of the stream's last slot
            __syncthreads();

            // ---------------- encoder backward + normalise backward + update (fp32, as the fused kernels)
            if (!args.noenc) {
                ring.gstream = args.enc_stream + (size_t)PNDF_GEN_ENC_SECTION_TILES * TILE_BYTES;
                ring_start(ring, wave);
                ring_wait_dma();
                __syncthreads();
                encoder_backward<ESP>(my_f, my_gn, eb, ring, ape, g);
                ring_wait_dma();
            }
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

// ------------------------------------------------------------------------------------------ host side
using namespace pndf;

struct PndfGeneric {
    pndf_config cfg;
    int L = 0;
    bool enc = true;
    int resident = 0;
    PndfGenericArgs plan;            // tile counts and offsets (pointers filled per launch)
    size_t wf_tiles = 0, wb_tiles = 0, lbias_floats = 0;
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
    return false;
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int gen_enc_act(const pndf_config& cfg) { return cfg.enc_act == -1 ? cfg.act : cfg.enc_act; }
static inline float gen_enc_beta(const pndf_config& cfg) { return cfg.enc_beta > 0.f ? cfg.enc_beta : cfg.beta; }
typedef void (*gen_kernel_t)(PndfGenericArgs);
static gen_kernel_t gen_kernel(const pndf_config& cfg, const char** name) {
    const bool sp = cfg.act == PNDF_ACT_SOFTPLUS, esp = (cfg.dims[0] == DIMS[0]) ? gen_enc_act(cfg) == PNDF_ACT_SOFTPLUS : sp;
    if (sp && esp) { *name = "pndf_generic_softplus_kernel"; return pndf_generic_softplus_kernel; }
    if (!sp && !esp) { *name = "pndf_generic_relu_kernel"; return pndf_generic_relu_kernel; }
    if (sp) { *name = "pndf_generic_softplus_reluenc_kernel"; return pndf_generic_softplus_reluenc_kernel; }
    *name = "pndf_generic_relu_spenc_kernel";
    return pndf_generic_relu_spenc_kernel;
}
// tiles -> tiles rounded up to a group count the layer code exists for (PNDF_GEN_GROUP_CASES)
static inline int gen_round_tiles(int tiles) {
    const int groups = ceil_div(tiles, NTB);
    const int full = (groups - 1) / GEN_PASS_GROUPS * GEN_PASS_GROUPS, rest = groups - full;      // whole passes of 8 groups + a last one
    return (full + rest) * NTB;      // (every 1 .. GEN_PASS_GROUPS groups of a last pass have their instantiation: PNDF_GEN_GROUP_CASES)
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
    PndfGenericArgs& P = g->plan;
    memset(&P, 0, sizeof(P));
    P.nlayers = L;
    P.noenc = g->enc ? 0 : 1;
    int wf = 0, wb = 0, bo = 0, slot = 2 * PNDF_GEN_XTILES;
    for (int l = 0; l < L; ++l) wf += gen_round_tiles(ceil_div(cfg.dims[l + 1], 16)) * ceil_div((l == 0) ? 128 : cfg.dims[l], 16);
    g->wf_tiles = wf;                 // the backward matrices follow the forward ones in ONE stream, in the order the backward pass walks them
    wf = 0;
    for (int l = L - 1; l >= 0; --l) {
        P.wb_off[l] = (int)g->wf_tiles + wb;
        wb += gen_round_tiles(ceil_div((l == 0) ? 128 : cfg.dims[l], 16)) * ceil_div(cfg.dims[l + 1], 16);
    }
    g->wb_tiles = wb;
    for (int l = 0; l < L; ++l) {
        const int in = (l == 0) ? 128 : cfg.dims[l];      // x0 is the pose's 128-row feature buffer (126 | 84 rows used, the rest zero)
        const int outw = cfg.dims[l + 1];
        P.kt[l] = ceil_div(in, 16);
        P.nt[l] = ceil_div(outw, 16);
        P.ktp[l] = gen_round_tiles(P.kt[l]);
        P.ntp[l] = gen_round_tiles(P.nt[l]);
        P.wf_off[l] = wf;
        P.b_off[l] = bo;
        P.d_off[l] = slot;
        wf += P.ntp[l] * P.kt[l];
        bo += 16 * P.ntp[l];
        if (l < L - 1) slot += P.ntp[l];
    }
    P.enc_d_off = slot;
    if (gen_enc_act(cfg) == PNDF_ACT_SOFTPLUS && g->enc) slot += 2 * NJ;
    P.wg_tiles = slot;
    P.w_slots = (int)((g->wf_tiles + g->wb_tiles) / GW_SLOT_TILES);      // (every pass is whole groups of NTB = 2 slots)
    g->lbias_floats = bo;
    hipError_t e = hipMalloc((void**)&g->d_bias, BIAS_FLOATS * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void**)&g->d_enc, (size_t)2 * PNDF_GEN_ENC_SECTION_TILES * TILE_BYTES);
    if (e == hipSuccess) e = hipMalloc((void**)&g->d_wf, (g->wf_tiles + g->wb_tiles) * TILE_BYTES);
    g->d_wb = nullptr;                // (one stream: the backward half lives behind the forward half in d_wf)
    if (e == hipSuccess) e = hipMalloc((void**)&g->d_lb, g->lbias_floats * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void**)&g->d_scratch, (size_t)resident_wgs * P.wg_tiles * SLOT_F4 * sizeof(f32x4));
    if (e == hipSuccess) e = hipEventCreateWithFlags(&g->done, hipEventDisableTiming);
    for (const void* kfn : {(const void*)pndf_generic_relu_kernel, (const void*)pndf_generic_softplus_kernel,
                            (const void*)pndf_generic_relu_spenc_kernel, (const void*)pndf_generic_softplus_reluenc_kernel})
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
    const PndfGenericArgs& P = g->plan;
    std::vector<float> wf(g->wf_tiles * TILE_FLOATS), wb(g->wb_tiles * TILE_FLOATS), lb(g->lbias_floats, 0.f), bias(BIAS_FLOATS, 0.f),
        enc((size_t)2 * PNDF_GEN_ENC_SECTION_TILES * TILE_FLOATS, 0.f);
    for (int l = 0; l < L; ++l) {
        const int in = g->cfg.dims[l], outw = g->cfg.dims[l + 1];
        const pndf_pack::Mat F{lin[2 * l], outw, in, false}, T{lin[2 * l], outw, in, true};
        float* dst = wf.data() + (size_t)P.wf_off[l] * TILE_FLOATS;
        const int PT = GEN_PASS_GROUPS * NTB;                  // consumption order: [pass of <= 32 output tiles][k tile][tile of the pass]
        for (int t0 = 0; t0 < P.ntp[l]; t0 += PT)
            for (int k = 0; k < P.kt[l]; ++k)
                for (int t = t0; t < P.ntp[l] && t < t0 + PT; ++t, dst += TILE_FLOATS) pndf_pack::emit_tile(F, t, k, dst);
        dst = wb.data() + (size_t)(P.wb_off[l] - (int)g->wf_tiles) * TILE_FLOATS;
        for (int t0 = 0; t0 < P.ktp[l]; t0 += PT)
            for (int k = 0; k < P.nt[l]; ++k)
                for (int t = t0; t < P.ktp[l] && t < t0 + PT; ++t, dst += TILE_FLOATS) pndf_pack::emit_tile(T, t, k, dst);
        memcpy(lb.data() + P.b_off[l], lin[2 * l + 1], sizeof(float) * outw);
    }
    for (int l = 0; l < 8; ++l) bias[SCALE_OFF + l] = 1.0f;
    if (g->enc) {
        for (int j = 0; j < NJ; ++j) {
            memcpy(bias.data() + ENCB_OFF + 32 * j, tensors[4 * j + 1], sizeof(float) * HID);
            memcpy(bias.data() + ENCB_OFF + 32 * j + 16 + ENC_FEAT_ROW, tensors[4 * j + 3], sizeof(float) * FEAT);
        }
        pndf_pack::emit_encoder_sections(tensors, enc.data(), enc.data() + (size_t)PNDF_GEN_ENC_SECTION_TILES * TILE_FLOATS);
    }
    hipError_t e = hipDeviceSynchronize();      // no launch may still be reading the old weights
    if (e == hipSuccess) e = hipMemcpy(g->d_wf, wf.data(), wf.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(g->d_wf + g->wf_tiles * TILE_FLOATS, wb.data(), wb.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(g->d_lb, lb.data(), lb.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(g->d_bias, bias.data(), bias.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(g->d_enc, enc.data(), enc.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) { err = std::string("pndf_load_weights (runtime-planned DFNet): ") + hipGetErrorString(e); return PNDF_ERR_HIP; }
    g->have_weights = true;
    return PNDF_OK;
}

const char* pndf_generic_kernel_name(const PndfGeneric* g) {
    const char* name = "";
    (void)gen_kernel(g->cfg, &name);
    return name;
}

int pndf_generic_launch(PndfGeneric* g, int mode, const float* q, const float* gout, float* qo, float* d, int64_t B, int steps,
                        void* stream, std::string& err) {
    if (!g->have_weights) { err = "pndf_load_weights has not been called"; return PNDF_ERR_NO_WEIGHTS; }
    PndfGenericArgs a = g->plan;
    a.q_in = q; a.q_out = qo; a.d_out = d; a.grad_out = gout;
    a.enc_stream = g->d_enc; a.bias = g->d_bias; a.wfwd = g->d_wf; a.wbwd = g->d_wf; a.lbias = g->d_lb; a.scratch = g->d_scratch;
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
        hipLaunchKernelGGL(gen_kernel(g->cfg, &name), grid, block, LDS_TOTAL, (hipStream_t)stream, a);
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
