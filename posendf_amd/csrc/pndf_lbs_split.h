// Split-precision form of the body-model kernels (pndf_lbs.hip): data layout and kernel arguments.  The fp32 layout and
// PndfLbsArgs live in pndf_args.h.
#pragma once
#include "pndf_args.h"

// One vertex group of the packed model for the split-precision kernels (bytes): every fp32 operand as fp16 hi + fp16 lo,
// scaled by a power of two.  Planes of halfs [row][16 v], v fastest -- ONE copy serves both contractions: the reverse pass
// contracts over vertices (plain 8-byte reads of a row), and ds_read_b64_tr_b16 turns 4 rows x 16 v into the forward
// pass's operand (one vertex per lane, four consecutive rows).
//   PH / PL [3 comps][224 k][16 v]   pose blend shapes x p_scale (rows 207..223 zero)      (tiles of 16 rows, see below)
//   WH / WL [32 joints][16 v]        skinning weights x w_scale (rows 24..31 zero)
//   VS [3 comps][16 v] fp32          shaped template        FL [16 v] int32   -2 padding, -1 ordinary, >= 0 extra-joint index
// Inside a tile of 16 rows (512 bytes) the 64 units of 8 bytes (row r, vertex quad q) sit at unit index
//   32 (r / 8) + 4 (r % 8) + (q ^ 2 (r / 8))
// i.e. rows 8 .. 15 carry their vertex quads swapped pairwise.  Both access patterns then touch 32 different 8-byte bank slots
// per 32-lane group (ds_read_b64 and ds_read_b64_tr_b16 are served in two 32-lane groups over 64 banks): the row reads (lane
// (g, p) -> row p, quad g: rows 0 .. 15 x quads {0, 1} | {2, 3}) and the transposed reads (lane m of lane group g -> row
// 4 g + m / 4, quad m % 4: rows 0 .. 7 | 8 .. 15 x all quads).  With the plain row-major order the row reads of rows r and
// r + 8 collide: 7 conflict cycles per LDS instruction measured (SQ_LDS_BANK_CONFLICT 4.3e9 -> 5e7 per launch).
__host__ __device__ constexpr int pndf_lbs_sb_unit(int r, int q) { return 32 * (r / 8) + 4 * (r % 8) + (q ^ (2 * (r / 8))); }
// byte offset of (row, vertex) inside a plane
__host__ __device__ constexpr int pndf_lbs_sb_at(int row, int v) {
    return (row / 16) * 512 + 8 * pndf_lbs_sb_unit(row % 16, v / 4) + 2 * (v % 4);
}
constexpr int PNDF_LBS_KP = 224;                                               // 7 k-blocks of 32
constexpr int PNDF_LBS_KB = PNDF_LBS_KP / 32;
constexpr int PNDF_LBS_SB_PLANE = PNDF_LBS_KP * PNDF_LBS_GV * 2;               // 7168 bytes per component
constexpr int PNDF_LBS_SB_PH = 0;
constexpr int PNDF_LBS_SB_PL = 3 * PNDF_LBS_SB_PLANE;                          // 21504
constexpr int PNDF_LBS_SB_WH = 2 * PNDF_LBS_SB_PL;                             // 43008
constexpr int PNDF_LBS_SB_WL = PNDF_LBS_SB_WH + 32 * PNDF_LBS_GV * 2;          // 44032
constexpr int PNDF_LBS_SB_VS = PNDF_LBS_SB_WL + 32 * PNDF_LBS_GV * 2;          // 45056
constexpr int PNDF_LBS_SB_FL = PNDF_LBS_SB_VS + 3 * PNDF_LBS_GV * 4;           // 45248
constexpr int PNDF_LBS_SB_BYTES = 46080;                                       // 45 KiB = 45 LDS-DMA pieces of 1 KiB
static_assert(PNDF_LBS_SB_FL + PNDF_LBS_GV * 4 <= PNDF_LBS_SB_BYTES && PNDF_LBS_SB_BYTES % 1024 == 0, "split blob layout");

// B operands per frame and lane group g, written by the pose kernel: element i of a k-block is contraction index
// 16 (i / 4) + 4 g + i % 4 -- the rows the two transposed reads of a k-block hand to lane group g.
constexpr int PNDF_LBS_PFS_HALFS = 4 * PNDF_LBS_KB * 2 * 8;                    // [g][k-block][hi, lo][8] = 448 halfs per frame
constexpr int PNDF_LBS_APS_HALFS = 4 * 12 * 2 * 8;                             // [g][entry][hi, lo][8]   = 768 halfs per frame
constexpr float PNDF_LBS_PF_SCALE = 4096.0f;                                   // |R - I| <= 2  ->  <= 2^13

struct PndfLbsSplitArgs {
    PndfLbsArgs base;
    const void* sblob;         // [NG][PNDF_LBS_SB_BYTES]
    void* pfs;                 // [S*T][PNDF_LBS_PFS_HALFS] halfs: pose feature x PNDF_LBS_PF_SCALE
    void* Aps;                 // [S*T][PNDF_LBS_APS_HALFS] halfs: relative joint transforms x a_scale
    float a_scale;             // power of two with |A| a_scale < 2^14
    float off_true;            // 1 / (p_scale PNDF_LBS_PF_SCALE): accumulator -> pose-blend offset
    float tm_true;             // 1 / (w_scale a_scale): accumulator -> skinning-transform entry
    float g_scale, x_scale;    // powers of two for the reverse operands d L / d v_posed and d L / d verts (x) [v_posed, 1]
    float gpf_true, gA_true;   // 1 / (p_scale g_scale), 1 / (w_scale x_scale)
    float reserved0;
};
static_assert(sizeof(PndfLbsSplitArgs) == sizeof(PndfLbsArgs) + 56, "PndfLbsSplitArgs layout");
