// The ONE definition of every kernel-argument struct of the library.  Each struct is filled by a C-ABI entry point
// (pndf_capi.hip, pndf_denoise.hip, pndf_quatdist.hip, pndf_lbs.hip) and read by a __global__ function that may live in
// another translation unit: both sides include this header, and the static_asserts pin the layout the kernels were
// written against (a silent mismatch between two copies would corrupt every launch).
#pragma once
#include "pndf_experiment.h"
#include <stddef.h>
#include <stdint.h>

enum { MODE_FORWARD = 0, MODE_FORWARD_GRAD = 1, MODE_PROJECT = 2 };

struct PndfKernelArgs {
    const float* q_in;      // [B,84]
    float* q_out;           // [B,84]  projected poses (PROJECT) or dd/dq * grad_out (FORWARD_GRAD)
    float* d_out;           // [B]
    const float* grad_out;  // [B] or null (FORWARD_GRAD only)
    const char* stream;     // packed trunk weights, STEP_TILES KiB (+ a replica of the first ring slots)
    const float* bias;      // BIAS_FLOATS
    float* dbg;             // null, or DBG_TOTAL*256 floats written by workgroup 0 (first step) / timing stamps
    long long B;
    int steps;
    int mode;               // MODE_FORWARD / MODE_FORWARD_GRAD / MODE_PROJECT (pndf_device.h)
    float slope;            // 0 = relu, 0.01 = lrelu
    float beta;             // softplus beta
    float* scratch;         // softplus: gridDim.x * SP_WG_FLOATS floats of derivative scratch, else null
    int reserved0;
    int noenc;              // 1 = model.StrEnc.use False: the trunk sees the normalised quaternions (in_dim 84)
};
static_assert(sizeof(PndfKernelArgs) == 96, "PndfKernelArgs layout");
static_assert(offsetof(PndfKernelArgs, stream) == 32 && offsetof(PndfKernelArgs, B) == 56 &&
              offsetof(PndfKernelArgs, steps) == 64 && offsetof(PndfKernelArgs, slope) == 72 &&
              offsetof(PndfKernelArgs, scratch) == 80 && offsetof(PndfKernelArgs, noenc) == 92, "PndfKernelArgs layout");

// ---- runtime-planned DFNet (pndf_generic.hip): any depth / width the reference's `dims` list can describe
constexpr int PNDF_GEN_MAXLIN = 8;         // linear layers (n_dims 3 .. 9)
constexpr int PNDF_GEN_NTB = 8;            // output tiles of one group: its 32 MFMAs hide the fetch of the next group's weights
constexpr int PNDF_GEN_XTILES = 64;        // tiles of the widest activation (1024 rows)
struct PndfGenericArgs {
    const float* q_in;      // [B,84]
    float* q_out;           // [B,84]
    float* d_out;           // [B]
    const float* grad_out;  // [B] or null
    const char* enc_stream; // the step's weight stream (1-KiB tiles): encoder forward (48) | trunk forward passes, then the transposed matrices in the
                            // backward pass's order, padded to whole 16-tile slots | encoder backward (48), + a replica of its first slots
    const float* bias;      // BIAS_FLOATS block (the encoder's biases)
    const float* wfwd;      // (unused: the trunk's tiles are part of enc_stream; wf_off / wb_off count tiles from its tile 48)
    const float* wbwd;
    const float* lbias;     // biases, each layer padded to whole blocks of tiles
    float* scratch;         // gridDim.x * wg_tiles tile slots of 4 KiB: activations (ping, pong), derivative factors per layer
    long long B;
    int steps, mode;
    float slope, beta;           // trunk: 0 | 0.01 (relu | lrelu), Softplus beta
    float enc_slope, enc_beta;   // the encoder's own (model.StrEnc.act / beta; equal to the trunk's in every config of the reference)
    int noenc, nlayers;
    int kt[PNDF_GEN_MAXLIN];     // contraction tiles of the forward pass = ceil(in / 16)
    int nt[PNDF_GEN_MAXLIN];     // contraction tiles of the backward pass = ceil(out / 16)
    int ktp[PNDF_GEN_MAXLIN];    // kt, nt rounded up to whole blocks: output tiles of the backward / forward pass
    int ntp[PNDF_GEN_MAXLIN];
    int wf_off[PNDF_GEN_MAXLIN]; // first tile of the layer in wfwd / wbwd
    int wb_off[PNDF_GEN_MAXLIN];
    int b_off[PNDF_GEN_MAXLIN];  // first float of the layer in lbias
    int d_off[PNDF_GEN_MAXLIN];  // first tile slot of the layer's derivative factors in the workgroup's scratch
    int enc_d_off;               // softplus: first tile slot of the encoder's 42 derivative tiles
    int wg_tiles;                // tile slots per workgroup
    int w_slots;                 // 16-tile ring slots of the step's weight stream
    // split-precision trunk (precision f16x3): k blocks of 32 = two operand tiles, weight pairs (hi tile, lo tile), exact scales
    int kb[PNDF_GEN_MAXLIN];     // contraction blocks of the forward pass = ceil(kt / 2)
    int nb[PNDF_GEN_MAXLIN];     // contraction blocks of the backward pass = ceil(nt / 2)
    float w_inv[PNDF_GEN_MAXLIN];// 1 / s_l: the stream carries s_l W, s_l = the power of two with max |W_l| s_l in [2^12, 2^13)
};
constexpr int PNDF_GEN_ENC_SECTION_TILES = 48 + 4 * 16;      // the encoder's 3 slots + what the ring fetches ahead (4 slots)

struct PndfDenoiseArgs {
    const float* theta_in; // [S,T,69] current poses (read: a frame's neighbours belong to other threads / workgroups)
    float* theta_out;      // [S,T,69] updated poses (the caller swaps the two buffers every step)
    const float* theta0;   // [S,T,69] the noisy input (data term of the pose-space surrogate)
    const float* d;        // [S*T] engine distances of the current theta
    const float* dq;       // [S*T,84] engine d d / d q (unit grad_outputs)
    float* m;              // Adam first moment  [S,T,69]
    float* v;              // Adam second moment [S,T,69]
    float* q_next;         // [S*T,84] quaternions of the UPDATED theta
    const float* g_extra;  // null, or [S,T,69]: gradient of the body-model terms (pndf_lbs_terms_grad), already weighted;
                           //   when given, the pose-space surrogate terms are NOT added
    int S, T, adam_step, prior_power;
    float lr, beta1, beta2, eps;
    // loss weights of this outer iteration, already evaluated (include/posendf_amd.h pndf_denoise_weights):
    //   pose prior  prior_coef c^prior_power  (c = mean_t d),   temporal  temp_coef mean ||..||,   data  data_coef mean ||..||
    float prior_coef, temp_coef, data_coef;
    int reserved0;
};
static_assert(sizeof(PndfDenoiseArgs) == 120 && offsetof(PndfDenoiseArgs, g_extra) == 64 &&
              offsetof(PndfDenoiseArgs, S) == 72 && offsetof(PndfDenoiseArgs, lr) == 88 &&
              offsetof(PndfDenoiseArgs, prior_coef) == 104, "PndfDenoiseArgs layout");

struct PndfQuatDistArgs {
    const float* noise;    // [B,21,4]
    const float* valid;    // [B,K,21,4]
    float* vals;           // [B,k]
    long long* idx;        // [B,k]
    int K, k, metric;      // metric 0 = geo, 1 = euc
    float w[21];           // joint weights (1/21 each when unweighted)
};
static_assert(sizeof(PndfQuatDistArgs) == 128 && offsetof(PndfQuatDistArgs, K) == 32 &&
              offsetof(PndfQuatDistArgs, w) == 44, "PndfQuatDistArgs layout");

// ---- linear-blend skinning (pndf_lbs.hip): SMPL-shaped body model, smplx lbs() restated
constexpr int PNDF_LBS_J = 24;             // joints of the kinematic tree (SMPL)
constexpr int PNDF_LBS_PF = 208;           // pose feature: 23 x 9 = 207 rotation-matrix entries, padded to 52 k-steps of 4
constexpr int PNDF_LBS_GV = 16;            // vertices per group = rows of one MFMA tile
constexpr int PNDF_LBS_MAX_EXTRA = 32;     // joints picked from vertices (SMPL: 21)
// one vertex group in the packed model ("blob", floats; lane-linear tiles, see pndf_lbs.hip):
//   P  [3 comps][208 k][16 v]   pose blend shapes      W [32 joints][16 v]  skinning weights (rows 24..31 zero)
//   VS [3 comps][16 v]          shaped template        FL [16 v] int32      -2 padding, -1 ordinary, >= 0 extra-joint index
constexpr int PNDF_LBS_BLOB_P = 0;
constexpr int PNDF_LBS_BLOB_W = 3 * PNDF_LBS_PF * PNDF_LBS_GV;                 // 9984
constexpr int PNDF_LBS_BLOB_VS = PNDF_LBS_BLOB_W + 32 * PNDF_LBS_GV;           // 10496
constexpr int PNDF_LBS_BLOB_FL = PNDF_LBS_BLOB_VS + 3 * PNDF_LBS_GV;           // 10544
constexpr int PNDF_LBS_BLOB_FLOATS = 10752;                                    // 42 KiB = 42 LDS-DMA pieces of 1 KiB
static_assert(PNDF_LBS_BLOB_FL + PNDF_LBS_GV <= PNDF_LBS_BLOB_FLOATS && PNDF_LBS_BLOB_FLOATS % 256 == 0, "blob layout");

struct PndfLbsModel {                      // per-model constants, by value in the arguments of the per-frame kernels
    float J[PNDF_LBS_J][3];                // rest joints: J_regressor (v_template + shapedirs betas)
    float rel[PNDF_LBS_J][3];              // joint relative to its parent (root: the joint itself)
    int parent[PNDF_LBS_J];
};

struct PndfLbsArgs {
    const float* theta;        // [S*T, 69] axis-angle body pose (global orientation = SMPL's zero parameter)
    const float* joints0;      // [S*T, 24 + NE, 3] joints of the initial poses (data term), or null
    const float* blob;         // [NG][PNDF_LBS_BLOB_FLOATS]
    const float* g_verts;      // general reverse pass: d L / d vertices [S*T, V, 3], or null
    const float* g_joints;     // general reverse pass: d L / d joints [S*T, 24 + NE, 3], or null
    float* pfp;                // [S*T, 208] pose feature, k-permuted for the B operand (lane group g holds k = 4 s + g)
    float* Ap;                 // [S*T, 4, 12, 6] relative joint transforms, B-operand order
    float* Gt;                 // [S*T, 24, 3] posed joints
    float* gpf;                // [vsplit, S*T, 208]  d L / d pose feature, partial per vertex range
    float* gA;                 // [vsplit, S*T, 12, 32] d L / d joint transform entries
    float* halo_pf;            // [vsplit, S*cps, 208] the same for the frame a chunk shares with the next chunk
    float* halo_A;             // [vsplit, S*cps, 12, 32]
    float* red;                // [207 + 12 * 24][S*T] the partial results summed over vertex ranges and halos, value-major
    float* verts;              // forward output [S*T, V, 3] or null
    float* joints;             // forward output [S*T, 24 + NE, 3] or null
    float* g_theta;            // [S*T, 69]
    int S, T, V, NG, NE, cps, vsplit, it_gt0;
    float w_temp, w_data;      // per element: 10 (1 + it) / ((T - 1) V),  100 / (1 + it) / (T (24 + NE))
    PndfLbsModel model;
};
