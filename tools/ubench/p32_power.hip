// Micro-benchmark: what would the pair-32 trunk (tests/test_p32_layout.py) buy on the dominant loop,
// measured the way that matters on this chip -- wall time of the whole GPU under its power cap, random operand data?
//
// Both variants run the (512 -> 1024 -> 512)-like chunk loop of the split kernel on the PRODUCT's weight ring
// (pndf_device.h: 5 x 16 KiB slots, LDS-DMA four slots ahead, one barrier per slot, one tile read per MFMA step) with
// random weights, three MFMAs per product block (hh, hl, lh), 64 poses per workgroup, 8 slots per chunk:
//   V16  today's ownership: a wave = 16 poses, reads all 16 tiles of a slot, 24 x v_mfma_f32_16x16x32_f16 per slot;
//        part A: chains on two chunk accumulators, part B: a chain of three per output tile (32 of them)
//   V32  a wave PAIR = 32 poses, each wave reads 8 of the 16 tiles, 12 x v_mfma_f32_32x32x16_f16 per slot (the same MFMA
//        time); part A: ONE chain over the wave's half of the contraction, part B: chains of six on 8 output tiles;
//        per chunk 2 + 2 ds_write_b128 / ds_read_b128 of exchange with the partner wave
// No epilogue arithmetic, no encoder: the loop that is 65 % of a step.  Results are not checked (finite, random).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-inline-asm -I posendf_amd/csrc tools/ubench/p32_power.hip -o gpurun_ab/p32_power
#include "pndf_device.h"
#include <stdio.h>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {
__device__ __forceinline__ f32x4 mf16(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 mf32(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ unsigned rnd(unsigned& s) { s = s * 1664525u + 1013904223u; return s; }
__device__ __forceinline__ f16x8 rand_operand(unsigned& s, float scale) {
    f16x8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (_Float16)(((int)(rnd(s) >> 16) % 2001 - 1000) * 1e-3f * scale);
    return v;
}
#define SB() __builtin_amdgcn_sched_barrier(0)

// ring events + the two DMA pieces of a half slot (group of 8 tiles starting at tile TN of its slot)
template <int TN>
__device__ __forceinline__ void group_events(Ring& ring, DmaSrc& src, uint32_t& dst) {
    if constexpr (TN == 0) ring_boundary(ring);
    if constexpr (TN == SLOT_TILES / 2) {
        ring_midslot_sync(ring);
        ring_dma_begin(ring, src, dst);
    }
}
template <int TN, int HALF>
__device__ __forceinline__ void dma_two(const DmaSrc& src, uint32_t dst) {
    // pieces 0,1 in the group that ran the mid-slot events, pieces 2,3 in the next one
    constexpr int p0 = (TN == SLOT_TILES / 2) ? 0 : 2;
    ring_dma_piece(src, dst, p0 + HALF);
}
}  // namespace

// ------------------------------------------------------------------ V16
// DUPDATA (round 3): all 8 tile reads are issued, but pairs 2, 3 of the group read the LDS addresses of pairs 0, 1 -- the
// operand DATA of the HALFREADS arm (every second pair repeats a weight tile) at V16's read volume: separates what halving
// the reads saves from what feeding the MFMA's A operand the same bits twice in a row saves.
template <int TN, int XB, int BASE, bool PART_A, bool HALFREADS = false, bool CHAIN6 = false, bool DUPDATA = false>
__device__ __forceinline__ void v16_group(Ring& ring, DmaSrc& src, uint32_t& dst, f16x8 (&cur)[8], const f16x8 (&xh)[16], const f16x8 (&xl)[16],
                                          f32x4 (&acc)[32]) {
    f16x8 nxt[8];
    // tiles of the NEXT group are read one per MFMA step; this group's are in `cur` (hi, lo, hi, lo, ...)
    constexpr int TNEXT = (TN + 8) % SLOT_TILES;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    SB();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        // CHAIN6 (diagnostic): part B visits every accumulator for two pairs in a row, as chunks of four tiles would
        const int a = PART_A ? (i & 1) : CHAIN6 ? (((BASE + i) >> 1) & 31) : ((BASE + i) & 31);
        const f16x8 wh = cur[2 * i], wl = cur[2 * i + 1];
        acc[a] = mf16(wh, xh[XB + i], acc[a]);
        SB();
        if (i == 0) group_events<TNEXT>(ring, src, dst);
        if (DUPDATA) nxt[2 * i] = __builtin_bit_cast(f16x8, ring_tile(ring, TNEXT + 2 * (i & 1)));
        else if (!HALFREADS || i < 2) nxt[2 * i] = __builtin_bit_cast(f16x8, ring_tile(ring, TNEXT + 2 * i));
        else nxt[2 * i] = nxt[2 * i - 4];                 // (diagnostic arm: V16's MFMA structure with pair-32's read volume)
        SB();
        acc[a] = mf16(wh, xl[XB + i], acc[a]);
        SB();
        if (DUPDATA) nxt[2 * i + 1] = __builtin_bit_cast(f16x8, ring_tile(ring, TNEXT + 2 * (i & 1) + 1));
        else if (!HALFREADS || i < 2) nxt[2 * i + 1] = __builtin_bit_cast(f16x8, ring_tile(ring, TNEXT + 2 * i + 1));
        else nxt[2 * i + 1] = nxt[2 * i - 3];
        SB();
        acc[a] = mf16(wl, xh[XB + i], acc[a]);
        SB();
        if (i == 1) dma_two<TNEXT, 0>(src, dst);
        if (i == 3) dma_two<TNEXT, 1>(src, dst);
        SB();
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) cur[i] = nxt[i];
}

template <int SL, bool HR, bool C6 = false, bool DUP = false>
__device__ __forceinline__ void v16_slots(Ring& ring, DmaSrc& src, uint32_t& dst, f16x8 (&cur)[8], const f16x8 (&xh)[16], const f16x8 (&xl)[16],
                                          f32x4 (&acc)[32], int& slot) {
    if constexpr (SL < 8) {
        v16_group<0, (SL & 3) * 4, SL * 8, (SL < 4), HR, C6, DUP>(ring, src, dst, cur, xh, xl, acc);
        v16_group<8, (SL & 3) * 4, SL * 8 + 4, (SL < 4), HR, C6, DUP>(ring, src, dst, cur, xh, xl, acc);
        if (++slot == STEP_SLOTS) { slot = 0; ring_next_step(ring); }
        v16_slots<SL + 1, HR, C6, DUP>(ring, src, dst, cur, xh, xl, acc, slot);
    }
}

template <bool HR, bool C6 = false, bool DUP = false>
__global__ void __launch_bounds__(256, 1) k16(const char* stream, float* out, int nchunks, unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 97u + 1u;
    Ring ring; ring.gstream = stream; ring.smem = smem; ring.lane = lane;
    ring_start(ring, wave);
    ring_wait_dma();
    __syncthreads();
    f16x8 xh[16], xl[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { xh[i] = rand_operand(s, 1.0f); xl[i] = rand_operand(s, 4e-4f); }
    f32x4 acc[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = f32x4{0, 0, 0, 0};
    f16x8 cur[8];
    ring_boundary(ring);
    for (int i = 0; i < 8; ++i) cur[i] = __builtin_bit_cast(f16x8, ring_tile(ring, i));
    DmaSrc src{nullptr, 0u};
    uint32_t dst = 0;
    int slot = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int c = 0; c < nchunks; ++c) v16_slots<0, HR, C6, DUP>(ring, src, dst, cur, xh, xl, acc, slot);
    if (threadIdx.x == 0) cyc[blockIdx.x] = __builtin_amdgcn_s_memtime() - t0;
    float r = 0;
    for (int i = 0; i < 32; ++i) r += acc[i][0] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = r;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ------------------------------------------------------------------ V32
template <int TN, int XB, int NX, bool INTERLEAVE>
__device__ __forceinline__ void v32_group(Ring& ring, DmaSrc& src, uint32_t& dst, f16x8 (&cur)[4], const f16x8 (&xh)[NX], const f16x8 (&xl)[NX],
                                          f32x16& acc0, f32x16& acc1, int half) {
    f16x8 nxt[4];
    constexpr int TNEXT = (TN + 8) % SLOT_TILES;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    SB();
    if constexpr (INTERLEAVE) {
        // the two pairs alternate, each on its own accumulator: no MFMA depends on the one issued just before it
        const f16x8 wh0 = cur[0], wl0 = cur[1], wh1 = cur[2], wl1 = cur[3];
        acc0 = mf32(wh0, xh[XB], acc0);
        SB();
        group_events<TNEXT>(ring, src, dst);
        nxt[0] = __builtin_bit_cast(f16x8, ring_tile(ring, TNEXT));
        SB();
        acc1 = mf32(wh1, xh[XB + 1], acc1);
        SB();
        nxt[1] = __builtin_bit_cast(f16x8, ring_tile(ring, TNEXT + 1));
        SB();
        acc0 = mf32(wh0, xl[XB], acc0);
        SB();
        nxt[2] = __builtin_bit_cast(f16x8, ring_tile(ring, TNEXT + 2));
        SB();
        acc1 = mf32(wh1, xl[XB + 1], acc1);
        SB();
        nxt[3] = __builtin_bit_cast(f16x8, ring_tile(ring, TNEXT + 3));
        SB();
        acc0 = mf32(wl0, xh[XB], acc0);
        SB();
        dma_two<TNEXT, 0>(src, dst);
        SB();
        acc1 = mf32(wl1, xh[XB + 1], acc1);
        SB();
        dma_two<TNEXT, 1>(src, dst);
        SB();
    } else {
#pragma unroll
    for (int i = 0; i < 2; ++i) {                       // this wave's two pairs of the group: tiles 4 half + 2 i, + 1
        f32x16& a = i ? acc1 : acc0;
        const f16x8 wh = cur[2 * i], wl = cur[2 * i + 1];
        a = mf32(wh, xh[XB + i], a);
        SB();
        if (i == 0) group_events<TNEXT>(ring, src, dst);
        nxt[2 * i] = __builtin_bit_cast(f16x8, ring_tile(ring, TNEXT + 2 * i));
        SB();
        a = mf32(wh, xl[XB + i], a);
        SB();
        nxt[2 * i + 1] = __builtin_bit_cast(f16x8, ring_tile(ring, TNEXT + 2 * i + 1));
        SB();
        a = mf32(wl, xh[XB + i], a);
        SB();
        if (i == 0) dma_two<TNEXT, 0>(src, dst);        // the two DMA pieces of this half slot
        else dma_two<TNEXT, 1>(src, dst);
        SB();
    }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) cur[i] = nxt[i];
}
template <int SL, bool IL>
__device__ __forceinline__ void v32_part_a(Ring& ring, DmaSrc& src, uint32_t& dst, f16x8 (&cur)[4], const f16x8 (&xh)[16], const f16x8 (&xl)[16],
                                           f32x16& part, f32x16& part2, int half, int& slot) {
    if constexpr (SL < 4) {
        v32_group<0, SL * 4, 16, IL>(ring, src, dst, cur, xh, xl, part, IL ? part2 : part, half);
        v32_group<8, SL * 4 + 2, 16, IL>(ring, src, dst, cur, xh, xl, part, IL ? part2 : part, half);
        if (++slot == STEP_SLOTS) { slot = 0; ring_next_step(ring); }
        v32_part_a<SL + 1, IL>(ring, src, dst, cur, xh, xl, part, part2, half, slot);
    }
}
template <int SL, bool IL>
__device__ __forceinline__ void v32_part_b(Ring& ring, DmaSrc& src, uint32_t& dst, f16x8 (&cur)[4], f16x8 (&yh)[2], const f16x8 (&yl)[2],
                                           f32x16 (&acc)[8], f32x16& part, char* mine, const char* theirs, int half, int& slot) {
    if constexpr (SL < 4) {
        if constexpr (SL == 1) {                          // partner's partial sums, epilogue stand-in, activated half out
            const f32x4 o0 = *(const f32x4*)(theirs), o1 = *(const f32x4*)(theirs + 16);
            f32x4 z0 = f32x4{part[0], part[1], part[2], part[3]} + o0, z1 = f32x4{part[4], part[5], part[6], part[7]} + o1;
            f16x8 nh;
            for (int j = 0; j < 4; ++j) { nh[j] = (_Float16)(z0[j] * 1e-3f); nh[4 + j] = (_Float16)(z1[j] * 1e-3f); }
            *(f16x8*)(mine + 2048) = nh;
            *(f16x8*)(mine + 2048 + 16) = nh;
            for (int j = 0; j < 16; ++j) part[j] = 0.f;
        }
        if constexpr (SL == 3) {                          // the partner's activated half of the NEXT chunk's operand
            const f16x8 t = *(const f16x8*)(theirs + 2048);
            yh[1] = t;
        }
        // interleaved: the two pairs of a group feed two DIFFERENT output tiles (each tile gets its two k-blocks from the two groups)
        v32_group<0, 0, 2, IL>(ring, src, dst, cur, yh, yl, acc[SL * 2], acc[IL ? SL * 2 + 1 : SL * 2], half);
        v32_group<8, 0, 2, IL>(ring, src, dst, cur, yh, yl, acc[IL ? SL * 2 : SL * 2 + 1], acc[SL * 2 + 1], half);
        if (++slot == STEP_SLOTS) { slot = 0; ring_next_step(ring); }
        v32_part_b<SL + 1, IL>(ring, src, dst, cur, yh, yl, acc, part, mine, theirs, half, slot);
    }
}

template <bool IL>
__global__ void __launch_bounds__(256, 1) k32(const char* stream, float* out, int nchunks, unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = wave & 1;
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 97u + 1u;
    Ring ring; ring.gstream = stream; ring.smem = smem; ring.lane = lane;
    ring_start(ring, wave);
    ring_wait_dma();
    __syncthreads();
    ring.lane = lane + 256 * half;      // tile reads: this wave's four tiles of a half slot start at tile 4 * half (4 KiB = 256 x 16 B)
    f16x8 xh[16], xl[16];                              // this wave's half of the contraction: 16 k-blocks of 16, 32 poses
#pragma unroll
    for (int i = 0; i < 16; ++i) { xh[i] = rand_operand(s, 1.0f); xl[i] = rand_operand(s, 4e-4f); }
    f32x16 acc[8], part, part2;
    for (int j = 0; j < 16; ++j) part2[j] = 0.f;
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    for (int j = 0; j < 16; ++j) part[j] = 0.f;
    f16x8 yh[2] = {rand_operand(s, 1.0f), rand_operand(s, 1.0f)}, yl[2] = {rand_operand(s, 4e-4f), rand_operand(s, 4e-4f)};
    f16x8 cur[4];
    ring_boundary(ring);
    for (int i = 0; i < 4; ++i) cur[i] = __builtin_bit_cast(f16x8, ring_tile(ring, i));
    DmaSrc src{nullptr, 0u};
    uint32_t dst = 0;
    char* xch = smem + LDS_F;                            // exchange window: [wave][lane][32 B] x 2
    char* mine = xch + wave * 4096 + lane * 32;
    const char* theirs = xch + (wave ^ 1) * 4096 + lane * 32;
    int slot = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int c = 0; c < nchunks; ++c) {
        // ---- part A: 4 slots, ONE accumulator chain over this wave's 16 k-blocks
        v32_part_a<0, IL>(ring, src, dst, cur, xh, xl, part, part2, half, slot);
        if constexpr (IL) { part = part + part2; for (int j = 0; j < 16; ++j) part2[j] = 0.f; }
        // ---- exchange 1: the partner finalises the other 16 rows: 8 registers out, 8 in (the barrier of the next slot orders it)
        *(f32x4*)(mine) = f32x4{part[8], part[9], part[10], part[11]};
        *(f32x4*)(mine + 16) = f32x4{part[12], part[13], part[14], part[15]};
        // ---- part B: 4 slots, chains of six on this wave's 8 output tiles
        v32_part_b<0, IL>(ring, src, dst, cur, yh, yl, acc, part, mine, theirs, half, slot);
    }
    if (threadIdx.x == 0) cyc[blockIdx.x] = __builtin_amdgcn_s_memtime() - t0;
    float r = (float)yh[1][0];
    for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][15];
    out[blockIdx.x * 256 + threadIdx.x] = r;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ------------------------------------------------------------------ V32b: pair-32 on 16x16x32 MFMAs
// A wave pair shares 32 poses as TWO 16-pose operand sets; a wave reads 8 of the 16 tiles of a slot and uses every weight
// tile for both pose halves: per pair of tiles hh_p0 hh_p1 hl_p0 hl_p1 (A = Wh kept for FOUR MFMAs) lh_p0 lh_p1, on two
// alternating accumulators.  Same MFMA count and time per slot as V16 (24 x 16 cycles), half the LDS reads.
template <int TN, int XB, int NX, int NACC, bool CHAIN = false, int DIAG = 0>
__device__ __forceinline__ void v32b_group(Ring& ring, DmaSrc& src, uint32_t& dst, f16x8 (&cur)[4], const f16x8 (&xh)[NX][2], const f16x8 (&xl)[NX][2],
                                           f32x4 (&acc)[NACC][2], int a0, int a1, int half) {
    f16x8 nxt[4];
    constexpr int TNEXT = (TN + 8) % SLOT_TILES;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    SB();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int a = i ? a1 : a0;
        const f16x8 wh = cur[2 * i], wl = cur[2 * i + 1];
        if constexpr (CHAIN) {        // one pose half after the other: chains of three per accumulator, A = Wh kept for two only;
                                      // all four tile reads of the next group behind the first four MFMAs of this one
            acc[a][0] = mf16(wh, xh[XB + i][0], acc[a][0]);
            SB();
            if (i == 0) {
                group_events<TNEXT>(ring, src, dst);
                nxt[0] = __builtin_bit_cast(f16x8, ring_tile(ring, TNEXT));
            }
            SB();
            acc[a][0] = mf16(wh, xl[XB + i][0], acc[a][0]);
            SB();
            if (i == 0) nxt[1] = __builtin_bit_cast(f16x8, ring_tile(ring, TNEXT + 1));
            SB();
            acc[a][0] = mf16(wl, xh[XB + i][0], acc[a][0]);
            SB();
            if (i == 0) nxt[2] = __builtin_bit_cast(f16x8, ring_tile(ring, TNEXT + 2));
            SB();
            constexpr int P1 = (DIAG == 1) ? 0 : 1;       // (diag 1: the second pose half reuses the first one's operands)
            constexpr int A1 = (DIAG == 2) ? 0 : 1;       // (diag 2: both pose halves accumulate onto one accumulator)
            acc[a][A1] = mf16(wh, xh[XB + i][P1], acc[a][A1]);
            SB();
            if (i == 0) nxt[3] = __builtin_bit_cast(f16x8, ring_tile(ring, TNEXT + 3));
            if (i == 0) dma_two<TNEXT, 0>(src, dst);
            else dma_two<TNEXT, 1>(src, dst);
            SB();
            acc[a][A1] = mf16(wh, xl[XB + i][P1], acc[a][A1]);
            acc[a][A1] = mf16(wl, xh[XB + i][P1], acc[a][A1]);
            SB();
            continue;
        }
        acc[a][0] = mf16(wh, xh[XB + i][0], acc[a][0]);
        SB();
        if (i == 0) group_events<TNEXT>(ring, src, dst);
        nxt[2 * i] = __builtin_bit_cast(f16x8, ring_tile(ring, TNEXT + 2 * i));
        SB();
        acc[a][1] = mf16(wh, xh[XB + i][1], acc[a][1]);
        SB();
        nxt[2 * i + 1] = __builtin_bit_cast(f16x8, ring_tile(ring, TNEXT + 2 * i + 1));
        SB();
        acc[a][0] = mf16(wh, xl[XB + i][0], acc[a][0]);
        acc[a][1] = mf16(wh, xl[XB + i][1], acc[a][1]);
        SB();
        if (i == 0) dma_two<TNEXT, 0>(src, dst);
        else dma_two<TNEXT, 1>(src, dst);
        SB();
        acc[a][0] = mf16(wl, xh[XB + i][0], acc[a][0]);
        acc[a][1] = mf16(wl, xh[XB + i][1], acc[a][1]);
        SB();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) cur[i] = nxt[i];
}
template <int SL, bool CHAIN, int DIAG>
__device__ __forceinline__ void v32b_slots(Ring& ring, DmaSrc& src, uint32_t& dst, f16x8 (&cur)[4], const f16x8 (&xh)[8][2], const f16x8 (&xl)[8][2],
                                           const f16x8 (&yh)[1][2], const f16x8 (&yl)[1][2], f32x4 (&ch)[2][2], f32x4 (&acc)[16][2],
                                           char* mine, const char* theirs, int half, int& slot) {
    if constexpr (SL < 8) {
        if constexpr (SL < 4) {       // part A: this wave's 8 k-blocks, two chunk tiles x two pose halves
            v32b_group<0, (SL & 3) * 2, 8, 2, CHAIN, DIAG>(ring, src, dst, cur, xh, xl, ch, 0, 1, half);
            v32b_group<8, (SL & 3) * 2, 8, 2, CHAIN, DIAG>(ring, src, dst, cur, xh, xl, ch, 0, 1, half);
        } else {                      // part B: this wave's 16 output tiles, one chunk k-block
            if constexpr (SL == 4) {  // exchange: partial chunk sums out, partner's in (stand-in for the epilogue)
                *(f32x4*)(mine) = ch[1][0];
                *(f32x4*)(mine + 16) = ch[1][1];
            }
            if constexpr (SL == 5) {
                ch[0][0] = ch[0][0] + *(const f32x4*)(theirs);
                ch[0][1] = ch[0][1] + *(const f32x4*)(theirs + 16);
                *(f32x4*)(mine + 2048) = ch[0][0];
                *(f32x4*)(mine + 2048 + 16) = ch[0][1];
            }
            if constexpr (SL == 7) {
                const f32x4 t = *(const f32x4*)(theirs + 2048);
                ch[1][0] = t * 1e-6f;
                ch[1][1] = t * 1e-6f;
            }
            v32b_group<0, 0, 1, 16, CHAIN, DIAG>(ring, src, dst, cur, yh, yl, acc, (SL - 4) * 4, (SL - 4) * 4 + 1, half);
            v32b_group<8, 0, 1, 16, CHAIN, DIAG>(ring, src, dst, cur, yh, yl, acc, (SL - 4) * 4 + 2, (SL - 4) * 4 + 3, half);
        }
        if (++slot == STEP_SLOTS) { slot = 0; ring_next_step(ring); }
        v32b_slots<SL + 1, CHAIN, DIAG>(ring, src, dst, cur, xh, xl, yh, yl, ch, acc, mine, theirs, half, slot);
    }
}
template <bool CHAIN, int DIAG = 0>
__global__ void __launch_bounds__(256, 1) k32b(const char* stream, float* out, int nchunks, unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = wave & 1;
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 97u + 1u;
    Ring ring; ring.gstream = stream; ring.smem = smem; ring.lane = lane;
    ring_start(ring, wave);
    ring_wait_dma();
    __syncthreads();
    ring.lane = lane + 256 * half;      // tile reads: this wave's four tiles of a half slot start at tile 4 * half (4 KiB = 256 x 16 B)
    f16x8 xh[8][2], xl[8][2], yh[1][2], yl[1][2];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int p = 0; p < 2; ++p) { xh[i][p] = rand_operand(s, 1.0f); xl[i][p] = rand_operand(s, 4e-4f); }
#pragma unroll
    for (int p = 0; p < 2; ++p) { yh[0][p] = rand_operand(s, 1.0f); yl[0][p] = rand_operand(s, 4e-4f); }
    f32x4 ch[2][2], acc[16][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) for (int p = 0; p < 2; ++p) ch[i][p] = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 16; ++i) for (int p = 0; p < 2; ++p) acc[i][p] = f32x4{0, 0, 0, 0};
    f16x8 cur[4];
    ring_boundary(ring);
#pragma unroll
    for (int i = 0; i < 4; ++i) cur[i] = __builtin_bit_cast(f16x8, ring_tile(ring, i));
    DmaSrc src{nullptr, 0u};
    uint32_t dst = 0;
    char* xch = smem + LDS_F;
    char* mine = xch + wave * 4096 + lane * 32;
    const char* theirs = xch + (wave ^ 1) * 4096 + lane * 32;
    int slot = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int c = 0; c < nchunks; ++c) v32b_slots<0, CHAIN, DIAG>(ring, src, dst, cur, xh, xl, yh, yl, ch, acc, mine, theirs, half, slot);
    if (threadIdx.x == 0) cyc[blockIdx.x] = __builtin_amdgcn_s_memtime() - t0;
    float r = ch[0][0][0] + ch[1][1][1];
#pragma unroll
    for (int i = 0; i < 16; ++i) r += acc[i][0][0] + acc[i][1][3];
    out[blockIdx.x * 256 + threadIdx.x] = r;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}


// ------------------------------------------------------------------ VROLE: producer / consumer wave pairs (round 3)
// A wave pair shares 32 poses as two 16-pose operand sets, but instead of splitting the contraction (V32 / V32b: partial
// sums and activated halves cross the pair twice per chunk) the two waves take the two PARTS of the fused layer pair:
//   A-wave  holds the phase input of BOTH pose sets (256 registers, read-only B operands), computes the chunk rows
//           (part A) for both: each weight pair read ONCE -> 6 MFMAs; hands the activated chunk over through LDS (4 KiB)
//   B-wave  holds the output accumulators of BOTH pose sets (256 registers), runs part B: again 6 MFMAs per pair read
// (lin2, lin3) has as many part-A as part-B pairs per chunk (32 + 32), so the two roles are balanced; the accumulator
// layer of one phase is the input layer of the next, so the roles simply swap from phase to phase with no exchange.
// Every slot carries 4 part-A pairs (tiles 0..7) and 4 part-B pairs (tiles 8..15): a wave reads 8 of the 16 tiles and
// issues 24 MFMAs per slot, as in V16.  ORDER: 0 = set-major (hh0 hl0 lh0 hh1 hl1 lh1: chains of three, Wh kept for
// two), 1 = term-major (hh0 hh1 hl0 hl1 lh0 lh1: Wh kept for four, two alternating accumulators).
// hl term with the lo operand held in an AccVGPR: hipcc never feeds an MFMA's A / B operand from the AGPR file by itself
// (it copies the 4 registers to VGPRs first: v_accvgpr_read x4 + s_nop in front of the MFMA), the ISA allows it.  The
// chain order hh (builtin), hl (this), lh (builtin) keeps a compiler-visible MFMA last on every accumulator, so the
// hazard recognizer still covers every later VALU read of the accumulator.
__device__ __forceinline__ f32x4 mf16_hl_agpr(f16x8 w, f16x8 x_lo, f32x4 c) {
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(w), "a"(x_lo));
    return c;
}
template <int TN, bool ROLE_B, int ORDER>
__device__ __forceinline__ void vrole_group(Ring& ring, DmaSrc& src, uint32_t& dst, f16x8 (&cur)[4], const f16x8 (&oh)[2][2],
                                            const f16x8 (&ol)[2][2], f32x4 (&acc0)[2], f32x4 (&acc1)[2]) {
    // this wave's two pairs of the half slot: tiles TN' .. TN'+3 of its own half (ring.lane carries the half offset)
    f16x8 nxt[4];
    constexpr int TNEXT = (TN + 8) % SLOT_TILES;      // event bookkeeping only: the wave visits both halves' events
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    SB();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        f32x4 (&a)[2] = i ? acc1 : acc0;
        const f16x8 wh = cur[2 * i], wl = cur[2 * i + 1];
        if constexpr (ORDER == 0) {
            a[0] = mf16(wh, oh[i][0], a[0]);
            SB();
            if (i == 0) group_events<TNEXT>(ring, src, dst);
            nxt[2 * i] = __builtin_bit_cast(f16x8, ring_tile(ring, (TNEXT / 2) + 2 * i));
            SB();
            a[0] = ROLE_B ? mf16(wh, ol[i][0], a[0]) : mf16_hl_agpr(wh, ol[i][0], a[0]);
            SB();
            nxt[2 * i + 1] = __builtin_bit_cast(f16x8, ring_tile(ring, (TNEXT / 2) + 2 * i + 1));
            SB();
            a[0] = mf16(wl, oh[i][0], a[0]);
            SB();
            if (i == 0) dma_two<TNEXT, 0>(src, dst);
            else dma_two<TNEXT, 1>(src, dst);
            SB();
            a[1] = mf16(wh, oh[i][1], a[1]);
            a[1] = ROLE_B ? mf16(wh, ol[i][1], a[1]) : mf16_hl_agpr(wh, ol[i][1], a[1]);
            a[1] = mf16(wl, oh[i][1], a[1]);
            SB();
        } else {
            a[0] = mf16(wh, oh[i][0], a[0]);
            SB();
            if (i == 0) group_events<TNEXT>(ring, src, dst);
            nxt[2 * i] = __builtin_bit_cast(f16x8, ring_tile(ring, (TNEXT / 2) + 2 * i));
            SB();
            a[1] = mf16(wh, oh[i][1], a[1]);
            SB();
            nxt[2 * i + 1] = __builtin_bit_cast(f16x8, ring_tile(ring, (TNEXT / 2) + 2 * i + 1));
            SB();
            a[0] = mf16(wh, ol[i][0], a[0]);
            a[1] = mf16(wh, ol[i][1], a[1]);
            SB();
            if (i == 0) dma_two<TNEXT, 0>(src, dst);
            else dma_two<TNEXT, 1>(src, dst);
            SB();
            a[0] = mf16(wl, oh[i][0], a[0]);
            a[1] = mf16(wl, oh[i][1], a[1]);
            SB();
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) cur[i] = nxt[i];
}

// A-role: 8 slots per chunk, 4 pairs per slot = k-blocks 2 SL, 2 SL + 1 x chunk tiles 0, 1
template <int SL, int ORDER>
__device__ __forceinline__ void vrole_a_slots(Ring& ring, DmaSrc& src, uint32_t& dst, f16x8 (&cur)[4], const f16x8 (&xh)[16][2],
                                              const f16x8 (&xl)[16][2], f32x4 (&ch)[2][2], int& slot) {
    if constexpr (SL < 8) {
        // group 0: (kb = 2 SL, ci = 0), (kb = 2 SL, ci = 1); group 1: the same for kb = 2 SL + 1
        const f16x8 oh0[2][2] = {{xh[2 * SL][0], xh[2 * SL][1]}, {xh[2 * SL][0], xh[2 * SL][1]}};
        const f16x8 ol0[2][2] = {{xl[2 * SL][0], xl[2 * SL][1]}, {xl[2 * SL][0], xl[2 * SL][1]}};
        vrole_group<0, false, ORDER>(ring, src, dst, cur, oh0, ol0, ch[0], ch[1]);
        const f16x8 oh1[2][2] = {{xh[2 * SL + 1][0], xh[2 * SL + 1][1]}, {xh[2 * SL + 1][0], xh[2 * SL + 1][1]}};
        const f16x8 ol1[2][2] = {{xl[2 * SL + 1][0], xl[2 * SL + 1][1]}, {xl[2 * SL + 1][0], xl[2 * SL + 1][1]}};
        vrole_group<8, false, ORDER>(ring, src, dst, cur, oh1, ol1, ch[0], ch[1]);
        if (++slot == STEP_SLOTS) { slot = 0; ring_next_step(ring); }
        vrole_a_slots<SL + 1, ORDER>(ring, src, dst, cur, xh, xl, ch, slot);
    }
}
// B-role: 4 pairs per slot = output tiles 4 SL .. 4 SL + 3, one chunk k-block
template <int SL, int ORDER>
__device__ __forceinline__ void vrole_b_slots(Ring& ring, DmaSrc& src, uint32_t& dst, f16x8 (&cur)[4], const f16x8 (&yh)[2],
                                              const f16x8 (&yl)[2], f32x4 (&acc)[32][2], int& slot) {
    if constexpr (SL < 8) {
        const f16x8 oh[2][2] = {{yh[0], yh[1]}, {yh[0], yh[1]}};
        const f16x8 ol[2][2] = {{yl[0], yl[1]}, {yl[0], yl[1]}};
        vrole_group<0, true, ORDER>(ring, src, dst, cur, oh, ol, acc[4 * SL], acc[4 * SL + 1]);
        vrole_group<8, true, ORDER>(ring, src, dst, cur, oh, ol, acc[4 * SL + 2], acc[4 * SL + 3]);
        if (++slot == STEP_SLOTS) { slot = 0; ring_next_step(ring); }
        vrole_b_slots<SL + 1, ORDER>(ring, src, dst, cur, yh, yl, acc, slot);
    }
}

template <int ORDER>
__global__ void __launch_bounds__(256, 1) krole(const char* stream, float* out, int nchunks, unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int role_b = wave & 1;
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 97u + 1u;
    Ring ring; ring.gstream = stream; ring.smem = smem; ring.lane = lane;
    ring_start(ring, wave);
    ring_wait_dma();
    __syncthreads();
    ring.lane = lane + 512 * role_b;     // tile reads: A-waves read tiles 0..7 of a slot, B-waves tiles 8..15 (8 KiB = 512 x 16 B)
    char* xch = smem + LDS_F;            // hand-over window per pair: [buffer][pose set][hi | lo][lane] x 16 B = 8 KiB
    char* win = xch + (wave >> 1) * 8192;
    DmaSrc src{nullptr, 0u};
    uint32_t dst = 0;
    int slot = 0;
    f16x8 cur[4];
    ring_boundary(ring);
#pragma unroll
    for (int i = 0; i < 4; ++i) cur[i] = __builtin_bit_cast(f16x8, ring_tile(ring, i));
    float r = 0;
    unsigned long long t0;
    if (!role_b) {
        f16x8 xh[16][2], xl[16][2];
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int p = 0; p < 2; ++p) { xh[i][p] = rand_operand(s, 1.0f); xl[i][p] = rand_operand(s, 4e-4f); }
        // the lo operands are BORN in the accumulation register file (v_accvgpr_write with an "=a" result) so that the
        // "a"-constrained hl MFMAs below read them in place: values that are defined in VGPRs get copied in front of every use
        if constexpr (ORDER == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i)
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
                    const u32x4v v = __builtin_bit_cast(u32x4v, xl[i][p]);
                    u32x4v r;
                    asm volatile("v_accvgpr_write_b32 %0, %4\n\tv_accvgpr_write_b32 %1, %5\n\tv_accvgpr_write_b32 %2, %6\n\tv_accvgpr_write_b32 %3, %7"
                                 : "=a"(r[0]), "=a"(r[1]), "=a"(r[2]), "=a"(r[3]) : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
                    xl[i][p] = __builtin_bit_cast(f16x8, r);
                }
        }
        f32x4 ch[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i) for (int p = 0; p < 2; ++p) ch[i][p] = f32x4{0, 0, 0, 0};
        t0 = __builtin_amdgcn_s_memtime();
        for (int c = 0; c < nchunks; ++c) {
            vrole_a_slots<0, ORDER>(ring, src, dst, cur, xh, xl, ch, slot);
            // epilogue stand-in + hand-over of the activated chunk (both pose sets, hi and lo): 4 x 16 B per lane
            char* w = win + (c & 1) * 4096 + lane * 16;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                f16x8 h;
                for (int j = 0; j < 4; ++j) { h[j] = (_Float16)(ch[0][p][j] * 1e-3f); h[4 + j] = (_Float16)(ch[1][p][j] * 1e-3f); }
                *(f16x8*)(w + p * 2048) = h;
                *(f16x8*)(w + p * 2048 + 1024) = h;
                ch[0][p] = f32x4{0, 0, 0, 0};
                ch[1][p] = f32x4{0, 0, 0, 0};
            }
        }
        r = ch[0][0][0];
    } else {
        f32x4 acc[32][2];
#pragma unroll
        for (int i = 0; i < 32; ++i) for (int p = 0; p < 2; ++p) acc[i][p] = f32x4{0, 0, 0, 0};
        f16x8 yh[2] = {rand_operand(s, 1.0f), rand_operand(s, 1.0f)}, yl[2] = {rand_operand(s, 4e-4f), rand_operand(s, 4e-4f)};
        t0 = __builtin_amdgcn_s_memtime();
        for (int c = 0; c < nchunks; ++c) {
            vrole_b_slots<0, ORDER>(ring, src, dst, cur, yh, yl, acc, slot);
            // the chunk the A-wave handed over one period ago (the slot barriers in between order the window)
            const char* w = win + ((c + 1) & 1) * 4096 + lane * 16;
            if (c > 0) {
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const f16x8 h = *(const f16x8*)(w + p * 2048);
                    const f16x8 l = *(const f16x8*)(w + p * 2048 + 1024);
                    yh[p] = h;
                    for (int j = 0; j < 8; ++j) yl[p][j] = l[j] * (_Float16)4e-4f;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 32; ++i) r += acc[i][0][0] + acc[i][1][3];
    }
    if (threadIdx.x == 0) cyc[blockIdx.x] = __builtin_amdgcn_s_memtime() - t0;
    out[blockIdx.x * 256 + threadIdx.x] = r;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <class K>
static float run(K kern, const char* name, const char* stream, float* out, int nchunks) {
    static unsigned long long* cyc = nullptr;
    if (!cyc) hipMalloc(&cyc, 1024 * 8);
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(1024), dim3(256), LDS_TOTAL, 0, stream, out, nchunks, cyc);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(1024), dim3(256), LDS_TOTAL, 0, stream, out, nchunks, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfma_flop = (double)1024 * 4 * nchunks * 192 * 2.0 * 16 * 16 * 32;     // 192 16x16x32-equivalents per wave and chunk
    static unsigned long long h[1024];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double mean = 0;
    for (int i = 0; i < 1024; ++i) mean += h[i];
    mean /= 1024;
    // s_memtime counts shader cycles (as in tools/gpu_region_timing.py); 4 workgroup rounds over 256 CUs
    printf("%-34s %8.2f ms  %8.1f TFLOP/s issued  %7.0f cycles per chunk (MFMA floor 3072)  clock %.2f GHz  (%s)\n", name, ms,
           mfma_flop / (ms * 1e-3) / 1e12, mean / nchunks, mean * 4 / (ms * 1e-3) / 1e9, hipGetErrorString(hipGetLastError()));
    return ms;
}

int main() {
    const size_t bytes = (size_t)(STEP_TILES + 4 * SLOT_TILES) * TILE_BYTES;
    char* stream; float* out;
    hipMalloc(&stream, bytes); hipMalloc(&out, 1024 * 256 * 4);
    _Float16* h = (_Float16*)malloc(bytes);
    unsigned s = 7u;
    for (size_t i = 0; i < bytes / 2; ++i) { s = s * 1664525u + 1013904223u; h[i] = (_Float16)(((int)(s >> 16) % 2001 - 1000) * 1e-3f); }
    hipMemcpy(stream, h, bytes, hipMemcpyHostToDevice);
    const int nchunks = 3200;                                 // = 100 steps of the 32-chunk phase
    for (int rep = 0; rep < 2; ++rep) {
        const float a = run(k16<false>, "V16 (16 poses per wave)", stream, out, nchunks);
        run(k16<true>, "V16 with half the tile reads (diag)", stream, out, nchunks);
        run(k16<false, false, true>, "V16, all reads, duplicated data (diag)", stream, out, nchunks);
        run(k16<false, true>, "V16, part B chains of six (diag)", stream, out, nchunks);
        const float b = run(k32<false>, "V32 pair-major chains", stream, out, nchunks);
        const float c = run(k32<true>, "V32 two accumulators interleaved", stream, out, nchunks);
        const float d = run(k32b<false>, "V32b 16x16x32, Wh kept x4", stream, out, nchunks);
        const float e = run(k32b<true>, "V32b 16x16x32, chains of three", stream, out, nchunks);
        run(k32b<true, 1>, "  diag: same B for both halves", stream, out, nchunks);
        run(k32b<true, 2>, "  diag: one accumulator per pair", stream, out, nchunks);
        const float f = run(krole<0>, "VROLE A-wave / B-wave, set-major", stream, out, nchunks);
        const float g = run(krole<1>, "VROLE A-wave / B-wave, term-major", stream, out, nchunks);
        printf("   time vs V16: VROLE set-major %.3f  term-major %.3f\n", f / a, g / a);
        printf("   time vs V16: V32 chains %.3f  V32 interleaved %.3f  V32b %.3f  V32b chains %.3f\n", b / a, c / a, d / a, e / a);
    }
    return 0;
}
