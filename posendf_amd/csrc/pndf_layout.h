// Shared host/device description of the packed weight stream consumed by the fused Pose-NDF kernel.
//
// The trunk (reference model/network/net_modules.py:46-72, dims configs/amass.yaml:26,30) is executed
// TRANSPOSED: every wave owns 16 poses and keeps their activations in registers in the MFMA C/D
// layout of v_mfma_f32_16x16x4_f32 (lane = (g, p): pose p = lane & 15, lane group g = lane >> 4;
// register r of output tile t holds row 16 t + 4 g + r).  That D layout is exactly the B-operand layout
// of the next layer's MFMA (k_local = g) provided the A operand (the weights) is fetched with the
// matching k permutation -- so activations never leave the register file and are never transposed.
// The only thing that streams is weights: one linear sequence of 1 KiB "tiles", pre-permuted on the
// host into consumption order, DMA'd global->LDS in 16-tile slots and read with one ds_read_b128 per
// lane per tile.
//
//   tile(M, nt, kt)[lane*4 + s] = M[16 nt + (lane & 15)][16 kt + 4 (lane >> 4) + s]      (M row-major)
//
// Layers are fused in pairs ("phases"): phase = (A: K_A -> R_A rows, chunked CT tiles at a time;
// B: R_A -> N_B rows accumulated in registers), so the wider intermediate (256 / 1024 / 256 wide) is
// never materialised beyond one chunk.  Backward phases use the transposed matrices.
#pragma once
#include "pndf_experiment.h"

namespace pndf {

constexpr int NJ = 21;            // joints, reference net_utils.py:46
constexpr int NQ = 84;            // 21 x 4 quaternion components
constexpr int NFEAT = 126;        // 21 x 6 encoder features, net_modules.py:116,125
constexpr int FEAT = 6;
constexpr int HID = 10;
constexpr int NLIN = 7;
constexpr int DIMS[NLIN + 1] = {126, 256, 512, 1024, 512, 256, 64, 1};
// Any other DFNet the reference can build (net_modules.py:14-28: `dims` is a free list of hidden widths) runs on the
// runtime-planned kernels of pndf_generic.hip: 2 .. 8 linear layers (n_dims 3 .. 9), hidden widths 1 .. 1024.
constexpr int MAXLIN = 8;
constexpr int MAX_WIDTH = 1024;

constexpr int TILE_FLOATS = 256;
constexpr int TILE_BYTES = 1024;
constexpr int SLOT_TILES = 16;    // tiles per LDS ring slot (16 KiB)
constexpr int WG_POSES = 64;      // 4 waves x 16 poses
constexpr int WG_THREADS = 256;

// phase description: A-part K tiles, chunk tiles, number of chunks, B-part output tiles
struct Phase {
    int KA, CT, NC, NB;
    int a_lin, b_lin;       // which dfnet.lin{l} supplies A / B
    bool transposed;        // backward phases use W^T
};

// forward: (lin0,lin1) (lin2,lin3) (lin4,lin5); backward: (lin5^T,lin4^T) (lin3^T,lin2^T) (lin1^T,lin0^T)
constexpr Phase PHASES[6] = {
    {8, 2, 8, 32, 0, 1, false},
    {32, 2, 32, 32, 2, 3, false},
    {32, 4, 4, 4, 4, 5, false},
    {4, 4, 4, 32, 5, 4, true},
    {32, 2, 32, 32, 3, 2, true},
    {32, 2, 8, 8, 1, 0, true},
};

constexpr int phase_chunk_tiles(const Phase& p) { return p.CT * p.KA + p.NB * p.CT; }
constexpr int phase_tiles(const Phase& p) { return p.NC * phase_chunk_tiles(p); }

// The encoder (21 BoneMLPs, reference net_modules.py:75-170) also runs on the MFMA pipe, batched over the 16
// poses of a wave: per joint two 16x16 weight tiles forward (W1: rows = 10 hidden units, k = 4|10 inputs;
// W2: rows 4..9 = the 6 features, k = hidden) and two transposed tiles backward.  Putting the features on
// rows 4..9 makes a joint's output tile (D layout) directly the B operand of its children (input vector =
// [quat(4) | parent feature(6)]: lane group 0 substitutes the child's own normalised quaternion).
// 42 tiles per direction, padded to 3 slots; forward tiles open the step's stream, backward tiles close it.
constexpr int ENC_TILES = 2 * 21;
constexpr int ENC_TILES_PADDED = 48;
constexpr int TRUNK_FWD_TILES = phase_tiles(PHASES[0]) + phase_tiles(PHASES[1]) + phase_tiles(PHASES[2]);
constexpr int TRUNK_BWD_TILES = phase_tiles(PHASES[3]) + phase_tiles(PHASES[4]) + phase_tiles(PHASES[5]);
constexpr int FWD_TILES = ENC_TILES_PADDED + TRUNK_FWD_TILES;
constexpr int BWD_TILES = TRUNK_BWD_TILES + ENC_TILES_PADDED;
constexpr int STEP_TILES = FWD_TILES + BWD_TILES;
constexpr int FWD_SLOTS = FWD_TILES / SLOT_TILES;
constexpr int STEP_SLOTS = STEP_TILES / SLOT_TILES;
static_assert(FWD_TILES % SLOT_TILES == 0 && BWD_TILES % SLOT_TILES == 0, "slot alignment");
static_assert(phase_chunk_tiles(PHASES[0]) % SLOT_TILES == 0, "chunk bodies are whole slots");
static_assert(phase_chunk_tiles(PHASES[1]) % SLOT_TILES == 0, "chunk bodies are whole slots");
static_assert(phase_chunk_tiles(PHASES[2]) % SLOT_TILES == 0, "chunk bodies are whole slots");
static_assert(TRUNK_FWD_TILES == 5312 && TRUNK_BWD_TILES == 5312 && STEP_TILES == 10720, "amass.yaml trunk");
static_assert(ENC_TILES_PADDED % SLOT_TILES == 0 && ENC_TILES <= ENC_TILES_PADDED - SLOT_TILES / 2 + 2,
              "every ring event of the encoder's slots must fall on a tile that is actually read");

// ---- split-precision stream (pndf_kernel_split.hip): every fp32 weight is carried as fp16 hi + fp16 lo and
// every product block is three v_mfma_f32_16x16x32_f16 (hi*hi + hi*lo + lo*hi, fp32 accumulate; the dropped
// lo*lo term is 2^-22 relative).  One MFMA contracts 32 k = TWO 16-row tiles of the previous layer, so the
// unit is a "pair": 1 KiB hi tile + 1 KiB lo tile of one (16 rows x 32 k) block,
//   block(M, nt, kb)[lane*8 + jj] = M[16 nt + (lane & 15)][16 (2 kb + (jj >> 2)) + 4 (lane >> 4) + (jj & 3)]
// (the k order inside an MFMA is free as long as A and B agree; this one makes the B operand = the packed
// registers of two consecutive C/D tiles).  Same tile count and slot structure as the fp32 stream; inside a
// phase the order is software-pipelined:  A(0) | A(1) B(0) | A(2) B(1) | ... | B(NC-1)
//   A(c): (kb, ci) pairs        B(c): (nb, b) pairs, b = k-block inside the chunk (CT / 2 of them)
// Tile parity (relied on by the two-term kernels, which fetch and read the hi tiles only): a pair occupies two consecutive stream
// tiles, hi first; every part of a chunk is an even number of tiles and starts on a slot boundary, so within a slot -- and within
// the four-tile window one wave fetches of it -- EVEN tiles are hi tiles and ODD tiles are lo tiles.
constexpr int PAIR_HI_TILE = 0, PAIR_LO_TILE = 1;
static_assert(SLOT_TILES % 2 == 0 && (SLOT_TILES / 4) % 2 == 0, "a wave's window of a slot holds whole pairs");
constexpr int phase_a_pairs(const Phase& p) { return (p.KA / 2) * p.CT; }
constexpr int phase_b_pairs(const Phase& p) { return p.NB * (p.CT / 2); }
static_assert(2 * (phase_a_pairs(PHASES[1]) + phase_b_pairs(PHASES[1])) == phase_chunk_tiles(PHASES[1]), "same tile count");
static_assert(2 * (phase_a_pairs(PHASES[2]) + phase_b_pairs(PHASES[2])) == phase_chunk_tiles(PHASES[2]), "same tile count");

// Experiment switch (split-precision stream and kernels only): chunk size of the two big phases (lin2,lin3) / (lin3^T,lin2^T).
// 2 = product; 4 = part B's accumulators get chains of six MFMAs per chunk instead of three (DESIGN.md Appendix C 7.1 b).
// (PNDF_BIG_CT: default in pndf_experiment.h)
enum Precision { PREC_FP32 = 0, PREC_F16X3 = 1, PREC_F16 = 2, PREC_BF16 = 3 };

// bias block (floats) copied to LDS: b0..b5, then w6 (64), then b6
constexpr int BIAS_OFF[NLIN] = {0, 256, 768, 1792, 2304, 2560, 2688};
constexpr int W6_OFF = 2624;
constexpr int NOENC_IN = 84;        // DFNet in_dim without the structure encoder (model.StrEnc.use = False): 21 x 4
constexpr int ENCB_OFF = 2692;      // encoder biases: per joint b1 padded to 16, then b2 on rows 4..9 of 16
constexpr int SCALE_OFF = ENCB_OFF + 21 * 32;   // split stream: 1 / (weight scale) of lin0..lin5 (8 floats; 1.0 in the fp32 block)
// split stream: norms for the a-priori operand bounds of the chunked layers (pndf_kernel_split.hip "operand scaling")
//   [0..2] ||W_l||_inf (largest absolute row sum), l = 0, 2, 4      [3..5] max |b_l|, l = 0, 2, 4
//   [6..8] ||W_l^T||_inf (largest absolute column sum), l = 5, 3, 1 [9..11] unused
constexpr int NORM_OFF = SCALE_OFF + 8;
constexpr int BIAS_FLOATS = NORM_OFF + 12;

constexpr int PARENT[NJ] = {-1, -1, -1, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19};
constexpr int enc_in(int j) { return PARENT[j] < 0 ? 4 : 10; }
constexpr int ENC_FEAT_ROW = 4;     // features live on rows 4..9 of the joint's output tile

// chunk-mask rows (one byte per lane per row) for the three chunked layers x1 (8 chunks), x3 (32), x5 (4 x 2 rows)
constexpr int MASK_BASE[3] = {0, 8, 40};

enum Act { ACT_RELU = 0, ACT_LRELU = 1, ACT_SOFTPLUS = 2 };

}  // namespace pndf
