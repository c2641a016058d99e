// Linear-blend skinning of an SMPL-shaped body model and the two body-model terms of the reference's motion-denoise
// objective, forward and reverse, for gfx950.
//
// What it replaces: experiments/body_model.py:27-40 (`smplx.SMPL(...)` called with betas, body_pose, global_orient = None)
// and experiments/motion_denoise.py:86-94 (vertex temporal term, joint data term) + the autograd pass through them.
// smplx is third-party code that is not under /root/reference: its PUBLISHED algorithm is restated here --
//   v_shaped = v_template + shapedirs betas;  J = J_regressor v_shaped                       (host, at create: betas are fixed)
//   R_j = Rodrigues(theta_j) (angle = |r + 1e-8|);  pose_feature = (R_1..R_23 - I)            [207]
//   v_posed = v_shaped + pose_feature posedirs                                                [V,3]   <- dense: 207 x 3V
//   G_j = G_parent(j) [R_j | J_j - J_parent(j)];  A_j = [G_R | G_t - G_R J_j]                 (rigid transform chain)
//   verts_v = sum_j W[v,j] (A_R[j] v_posed_v + A_t[j]);  joints = (G_t[0..23], verts[extra_joint_vertex])
// -- parity unpinned (oracle/lbs_np.py says the same).
//
// Work: per frame 2 x 4.28 M MACs for the pose blend shapes (forward + reverse) and 2 x 2 M for the skinning transforms
// against 276 B of pose in and 276 B of gradient out: this is dense fp32 contraction work, so it runs on the fp32 MFMA
// pipe (v_mfma_f32_16x16x4_f32), "transposed" like the distance engine: D rows = 16 vertices (or 16 pose-feature entries /
// joints in the reverse pass), D columns = 16 FRAMES of one wave.  Vertices, skinning matrices and vertex gradients of a
// (16 vertex x 16 frame) tile live in registers only; nothing per-vertex ever goes to HBM in the fused-terms mode.
//
// Two arithmetics for the vertex-side contractions: fp32 MFMAs as described here (PNDF_LBS_FP32), and -- the default for the
// forward and fused-terms passes -- fp16 MFMAs on operands split into hi + lo halves (PNDF_LBS_F16X3, "split precision" below:
// 212 MFMAs of 16 cycles per tile instead of 480 of 32).
//
// Three stages (fp32 form; the split-precision vertex kernels are described at their section):
//   pndf_lbs_pose_kernel            one thread per frame: Rodrigues, transform chain -> pose feature, A, posed joints
//                                   (_smpl_: SMPL's own tree as a compile-time table -- arrays in registers, no scratch)
//   pndf_lbs_vertex_kernel<MODE>    one wave per chunk of 16 frames, four chunks per workgroup sharing the model stream:
//       the packed model ("blob": 42 KiB per 16 vertices, lane-linear MFMA tiles) is streamed global -> LDS by DMA into
//       three buffers (the loop is software-pipelined one group deep), and read TWICE from LDS: as A operand of the forward pose-blend contraction (rows = vertices) and,
//       with a transposed conflict-free ds_read_b128, as A operand of the reverse contraction (rows = pose-feature entries).
//       MODE 0: vertices / vertex-picked joints out.  MODE 1: the two weighted terms of motion_denoise.py:86-94 formed in
//       registers (neighbouring frames are neighbouring lanes) and pushed straight back: d L / d pose_feature and
//       d L / d A accumulate in registers over all vertices.  MODE 2: general reverse pass for given d L / d verts.
//   pndf_lbs_pose_backward_kernel   one thread per frame: reverse of the transform chain and of Rodrigues -> d L / d theta
//                                   (SMPL's tree: pndf_lbs_pose_backward_smpl_kernel + pndf_lbs_rodrigues_vjp_kernel)
// In MODE 1 a chunk is 16 frames = 15 pairs; consecutive chunks share a frame, whose two partial results ("halo") are summed
// by the last kernel.  Small problems split the vertex range over blockIdx.y (`vsplit`) and sum the partials there too, in
// a fixed order: results are deterministic.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <cmath>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "../../include/posendf_amd.h"
#include "pndf_args.h"
#include "pndf_host.h"
#include "pndf_lbs_split.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int NJ = PNDF_LBS_J, PF = PNDF_LBS_PF, GV = PNDF_LBS_GV, BLOB = PNDF_LBS_BLOB_FLOATS;
constexpr int KS = PF / 4;                 // 52 k-steps of the pose-blend contraction
constexpr int KT = PF / 16;                // 13 row tiles of d L / d pose_feature
constexpr int C_STRIDE = PF * GV;          // floats per component of P in a blob (3328)
constexpr int A_FLOATS = 12 * 32;          // d L / d A per frame: [12 entries][32 joints (24 used)]

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// Neighbouring FRAMES are neighbouring lanes of a 16-lane row (lane = 16 g + p, p = frame of the chunk): one DPP move each,
// no LDS round trip.  row_shl:1 -- lane p reads lane p + 1 (frame t + 1), row_shr:1 -- lane p reads lane p - 1; lanes that
// would read across the row's edge get 0 (bound_ctrl).
__device__ __forceinline__ float next_frame(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x101, 0xf, 0xf, true));
}
__device__ __forceinline__ float prev_frame(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x111, 0xf, 0xf, true));
}

// smplx batch_rodrigues: angle = |r + 1e-8|, axis = r / angle, R = I + sin K + (1 - cos) K K   (row-major 3x3)
__device__ __forceinline__ void rodrigues(float rx, float ry, float rz, float* R) {
    const float ax = rx + 1e-8f, ay = ry + 1e-8f, az = rz + 1e-8f;
    const float th = sqrtf(ax * ax + ay * ay + az * az);
    const float nx = rx / th, ny = ry / th, nz = rz / th;
    float s, c;
    sincosf(th, &s, &c);
    const float c1 = 1.0f - c;
    R[0] = 1.0f - c1 * (nz * nz + ny * ny); R[1] = -s * nz + c1 * nx * ny;          R[2] = s * ny + c1 * nx * nz;
    R[3] = s * nz + c1 * nx * ny;          R[4] = 1.0f - c1 * (nz * nz + nx * nx); R[5] = -s * nx + c1 * ny * nz;
    R[6] = -s * ny + c1 * nx * nz;         R[7] = s * nx + c1 * ny * nz;           R[8] = 1.0f - c1 * (nx * nx + ny * ny);
}

// reverse of rodrigues: d <gR, R(r)> / d r
__device__ __forceinline__ void rodrigues_vjp(float rx, float ry, float rz, const float* gR, float* gr) {
    const float a[3] = {rx + 1e-8f, ry + 1e-8f, rz + 1e-8f};
    const float th = sqrtf(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
    const float nx = rx / th, ny = ry / th, nz = rz / th;
    float s, c;
    sincosf(th, &s, &c);
    const float c1 = 1.0f - c;
    const float K[9] = {0.f, -nz, ny, nz, 0.f, -nx, -ny, nx, 0.f};
    const float KK[9] = {-(nz * nz + ny * ny), nx * ny, nx * nz, nx * ny, -(nz * nz + nx * nx), ny * nz,
                         nx * nz, ny * nz, -(nx * nx + ny * ny)};
    float g_th = 0.f;
#pragma unroll
    for (int i = 0; i < 9; ++i) g_th += gR[i] * (c * K[i] + s * KK[i]);
    // gK = s gR + (1 - c) (gR K^T + K^T gR)
    float gK[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < 3; ++k) acc += gR[3 * i + k] * K[3 * j + k] + K[3 * k + i] * gR[3 * k + j];
            gK[3 * i + j] = s * gR[3 * i + j] + c1 * acc;
        }
    const float gn[3] = {gK[7] - gK[5], gK[2] - gK[6], gK[3] - gK[1]};
    g_th -= (gn[0] * rx + gn[1] * ry + gn[2] * rz) / (th * th);          // n = r / th
#pragma unroll
    for (int i = 0; i < 3; ++i) gr[i] = gn[i] / th + g_th * a[i] / th;
}

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): an unrolled loop whose index is a constant by construction
// (`#pragma unroll` may leave a long loop rolled, with its index arithmetic -- and every array it indexes -- at run time)
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// The kinematic tree of the per-frame kernels.  SMPL's own tree as a compile-time table: with every joint loop unrolled over
// constants the per-frame arrays (24 rotations, 24 global transforms, their gradients) are REGISTERS (up to ~500 of the 512
// a lone wave has); with the tree read from the model at run time they are dynamically indexed, i.e. 2 - 4 KB of scratch
// memory per frame, and the kernels are bound by its latency (0.83 + 1.57 ms at 153,600 frames against the time below).
struct SmplTree {
    static constexpr int tab[NJ] = {-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21};
    static constexpr bool STATIC = true;
    static __device__ __forceinline__ constexpr int parent(const PndfLbsModel&, int j) { return tab[j]; }
};
struct ModelTree {
    static constexpr bool STATIC = false;
    static __device__ __forceinline__ int parent(const PndfLbsModel& m, int j) { return m.parent[j]; }
};
// loop over joints j0 .. NJ - 1: unrolled over constants for the static tree, a plain loop otherwise
template <class Tree, int J0, class F>
__device__ __forceinline__ void for_joints(F&& f) {
    if constexpr (Tree::STATIC) static_for<NJ - J0>([&](auto jc) __attribute__((always_inline)) { f(J0 + decltype(jc)::value); });
    else for (int j = J0; j < NJ; ++j) f(j);
}
template <class Tree, int J0, class F>
__device__ __forceinline__ void for_joints_reverse(F&& f) {
    if constexpr (Tree::STATIC) static_for<NJ - J0>([&](auto jc) __attribute__((always_inline)) { f(NJ - 1 - decltype(jc)::value); });
    else for (int j = NJ - 1; j >= J0; --j) f(j);
}

// rotations, global rotations and global translations of the 24 joints of one frame (smplx batch_rigid_transform)
template <class Tree>
__device__ __forceinline__ void frame_transforms(const float* th, const PndfLbsModel& m, float (&R)[NJ][9], float (&GR)[NJ][9],
                                                 float (&Gt)[NJ][3]) {
    rodrigues(0.f, 0.f, 0.f, R[0]);          // SMPL's global_orient parameter: zeros (body_model.py:35-40 passes None)
    for_joints<Tree, 1>([&](int j) __attribute__((always_inline)) { rodrigues(th[3 * j - 3], th[3 * j - 2], th[3 * j - 1], R[j]); });
#pragma unroll
    for (int i = 0; i < 9; ++i) GR[0][i] = R[0][i];
#pragma unroll
    for (int i = 0; i < 3; ++i) Gt[0][i] = m.rel[0][i];
    for_joints<Tree, 1>([&](int j) __attribute__((always_inline)) {
        const int p = Tree::parent(m, j);
#pragma unroll
        for (int a = 0; a < 3; ++a) {
#pragma unroll
            for (int b = 0; b < 3; ++b)
                GR[j][3 * a + b] = GR[p][3 * a] * R[j][b] + GR[p][3 * a + 1] * R[j][3 + b] + GR[p][3 * a + 2] * R[j][6 + b];
            Gt[j][a] = GR[p][3 * a] * m.rel[j][0] + GR[p][3 * a + 1] * m.rel[j][1] + GR[p][3 * a + 2] * m.rel[j][2] + Gt[p][a];
        }
    });
}

// chunks of frames per sequence: fused-terms mode walks 15 pairs per chunk (consecutive chunks share a frame)
__host__ __device__ inline int chunks_fwd(int T) { return (T + 15) / 16; }
__host__ __device__ inline int chunks_pairs(int T) { return T > 1 ? (T - 1 + 14) / 15 : 1; }

}  // namespace

// ------------------------------------------------------------------ per-frame forward
template <class Tree>
__device__ __forceinline__ void lbs_pose_body(const PndfLbsArgs& a) {
    const long long n = (long long)blockIdx.x * 64 + threadIdx.x;
    if (n >= (long long)a.S * a.T) return;
    float R[NJ][9], GR[NJ][9], Gt[NJ][3];
    frame_transforms<Tree>(a.theta + n * 69, a.model, R, GR, Gt);
    // pose feature, k-permuted so that lane group g of the vertex kernel reads its 52 values contiguously:
    // pfp[g * 52 + s] = pose_feature[4 s + g]
    float* pf = a.pfp + n * PF;
    for_joints<Tree, 1>([&](int j) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < 9; ++e) {
            const int k = 9 * (j - 1) + e;
            pf[(k & 3) * KS + (k >> 2)] = R[j][e] - ((e % 4 == 0) ? 1.0f : 0.0f);
        }
    });
    pf[3 * KS + KS - 1] = 0.f;      // k = 207: padding
    // A_j = [G_R | G_t - G_R J_j] in B-operand order: Ap[g][entry][s] = A[joint 4 s + g][entry]
    float* Ap = a.Ap + n * 288;
    const int njt = NJ + a.NE;
    for_joints<Tree, 0>([&](int j) __attribute__((always_inline)) {
        const int g = j & 3, s = j >> 2;
#pragma unroll
        for (int e = 0; e < 9; ++e) Ap[g * 72 + e * 6 + s] = GR[j][e];
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            Ap[g * 72 + (9 + e) * 6 + s] = Gt[j][e] - (GR[j][3 * e] * a.model.J[j][0] + GR[j][3 * e + 1] * a.model.J[j][1] +
                                                        GR[j][3 * e + 2] * a.model.J[j][2]);
            if (a.Gt) a.Gt[n * (NJ * 3) + 3 * j + e] = Gt[j][e];
            if (a.joints) a.joints[(n * njt + j) * 3 + e] = Gt[j][e];
        }
    });
}
extern "C" __global__ void __launch_bounds__(64) pndf_lbs_pose_kernel(PndfLbsArgs a) { lbs_pose_body<ModelTree>(a); }
extern "C" __global__ void __launch_bounds__(64) pndf_lbs_pose_smpl_kernel(PndfLbsArgs a) { lbs_pose_body<SmplTree>(a); }

// ------------------------------------------------------------------ joints only (no vertices asked for)
// smplx's joints are the 24 posed chain joints + the vertices its VertexJointSelector picks (21 for SMPL).  When the caller
// wants only them (`joints_of()`: the Jtr of the noisy poses, motion_denoise.py:60,63), skinning all 6,890 vertices -- 431
// groups through the vertex kernel, 6.6 ms for 153,600 frames -- to read 21 of them is waste: one thread per frame does the
// kinematic chain and then skins the picked vertices alone, from a small table made at create time (per picked vertex:
// shaped template [3] | skinning weights [24] | pose blend shapes [207][3]; uniform addresses: scalar loads).  fp32 on the VALU.
constexpr int PICK_FLOATS = 3 + NJ + 3 * 9 * (NJ - 1);
template <class Tree>
__device__ __forceinline__ void lbs_joints_only_body(const PndfLbsArgs& a, const float* __restrict__ picked) {
    // v_posed of the picked vertices between the two passes: [x * 3 + c][thread] (one column per thread: conflict-free)
    __shared__ float vps[PNDF_LBS_MAX_EXTRA * 3][64];
    const long long n = (long long)blockIdx.x * 64 + threadIdx.x;
    if (n >= (long long)a.S * a.T) return;
    const float* th = a.theta + n * 69;
    // ONE set of 24 matrices: first the joints' own rotations (what the pose blend shapes need), then -- in place, parents
    // before children -- the global ones (what skinning needs); both sets at once do not fit a lone wave's registers
    float M[NJ][9], Gt[NJ][3];
    rodrigues(0.f, 0.f, 0.f, M[0]);
    for_joints<Tree, 1>([&](int j) __attribute__((always_inline)) { rodrigues(th[3 * j - 3], th[3 * j - 2], th[3 * j - 1], M[j]); });
    for (int x = 0; x < a.NE; ++x) {              // v_posed = v_shaped + posedirs^T (R - I)
        const float* P = picked + (size_t)x * PICK_FLOATS;
        float vp[3] = {P[0], P[1], P[2]};
        for_joints<Tree, 1>([&](int j) __attribute__((always_inline)) {
#pragma unroll
            for (int e = 0; e < 9; ++e) {
                const float pf = M[j][e] - ((e % 4 == 0) ? 1.0f : 0.0f);
                const float* pd = P + 3 + NJ + 3 * (9 * (j - 1) + e);
                vp[0] = fmaf(pd[0], pf, vp[0]);
                vp[1] = fmaf(pd[1], pf, vp[1]);
                vp[2] = fmaf(pd[2], pf, vp[2]);
            }
        });
#pragma unroll
        for (int c = 0; c < 3; ++c) vps[3 * x + c][threadIdx.x] = vp[c];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) Gt[0][i] = a.model.rel[0][i];
    for_joints<Tree, 1>([&](int j) __attribute__((always_inline)) {      // smplx batch_rigid_transform, as frame_transforms
        const int p = Tree::parent(a.model, j);
        float G[9];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
            for (int b = 0; b < 3; ++b) G[3 * r + b] = M[p][3 * r] * M[j][b] + M[p][3 * r + 1] * M[j][3 + b] + M[p][3 * r + 2] * M[j][6 + b];
            Gt[j][r] = M[p][3 * r] * a.model.rel[j][0] + M[p][3 * r + 1] * a.model.rel[j][1] + M[p][3 * r + 2] * a.model.rel[j][2] + Gt[p][r];
        }
#pragma unroll
        for (int i = 0; i < 9; ++i) M[j][i] = G[i];
    });
    const int njt = NJ + a.NE;
    for_joints<Tree, 0>([&](int j) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < 3; ++e) a.joints[(n * njt + j) * 3 + e] = Gt[j][e];
    });
    for (int x = 0; x < a.NE; ++x) {              // sum_j W[v, j] (G_R[j] (v_posed - J_j) + G_t[j])
        const float* P = picked + (size_t)x * PICK_FLOATS;
        const float vp[3] = {vps[3 * x][threadIdx.x], vps[3 * x + 1][threadIdx.x], vps[3 * x + 2][threadIdx.x]};
        float v[3] = {0.f, 0.f, 0.f};
        for_joints<Tree, 0>([&](int j) __attribute__((always_inline)) {
            const float w = P[3 + j];
            const float u0 = vp[0] - a.model.J[j][0], u1 = vp[1] - a.model.J[j][1], u2 = vp[2] - a.model.J[j][2];
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c] = fmaf(w, M[j][3 * c] * u0 + M[j][3 * c + 1] * u1 + M[j][3 * c + 2] * u2 + Gt[j][c], v[c]);
        });
#pragma unroll
        for (int c = 0; c < 3; ++c) a.joints[(n * njt + NJ + x) * 3 + c] = v[c];
    }
}
extern "C" __global__ void __launch_bounds__(64) pndf_lbs_joints_only_kernel(PndfLbsArgs a, const float* picked) { lbs_joints_only_body<ModelTree>(a, picked); }
extern "C" __global__ void __launch_bounds__(64) pndf_lbs_joints_only_smpl_kernel(PndfLbsArgs a, const float* picked) { lbs_joints_only_body<SmplTree>(a, picked); }

__device__ __forceinline__ void lbs_rotate(f32x4 (&off)[3], f32x4 (&Tm)[12], const f32x4 (&off_n)[3], const f32x4 (&Tm_n)[12]) {
#pragma unroll
    for (int i = 0; i < 3; ++i) off[i] = off_n[i];
#pragma unroll
    for (int i = 0; i < 12; ++i) Tm[i] = Tm_n[i];
}

// ------------------------------------------------------------------ per (16 vertices x 16 frames) tile
template <int MODE>
__device__ __forceinline__ void lbs_vertex_body(const PndfLbsArgs& a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];      // 3 x BLOB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, p = lane & 15;
    constexpr int STRIDE = (MODE == 1) ? 15 : 16;
    const int T = a.T, cps = a.cps, nch = a.S * cps, njt = NJ + a.NE;
    int cid = blockIdx.x * 4 + wave;
    const bool wave_on = cid < nch;
    if (!wave_on) cid = nch - 1;                    // idle waves shadow the last chunk (they take part in the DMA / barriers)
    const int s = cid / cps, c = cid - s * cps, f0 = c * STRIDE;
    const int t = f0 + p;
    const bool t_ok = wave_on && t < T;
    const long long n = (long long)s * T + (t < T ? t : T - 1);
    const bool owned = t_ok && (MODE != 1 || p < 15 || t == T - 1);     // MODE 1: lane 15 belongs to the next chunk
    const bool pair_ok = MODE == 1 && t_ok && p < 15 && t + 1 < T;
    const long long N = (long long)a.S * T;

    // ---- B operands of this wave's 16 frames (constant over the vertex groups)
    float pfB[KS], AB[72];
    {
        const f32x4* src = (const f32x4*)(a.pfp + n * PF + g * KS);
#pragma unroll
        for (int i = 0; i < KS / 4; ++i) {
            const f32x4 v = src[i];
#pragma unroll
            for (int r = 0; r < 4; ++r) pfB[4 * i + r] = v[r];
        }
        const f32x4* srcA = (const f32x4*)(a.Ap + n * 288 + g * 72);
#pragma unroll
        for (int i = 0; i < 18; ++i) {
            const f32x4 v = srcA[i];
#pragma unroll
            for (int r = 0; r < 4; ++r) AB[4 * i + r] = v[r];
        }
        // make hipcc wait for these loads HERE: left to the first use, its vmcnt(N) waits sit inside the group loop, where they
        // also count the model fetch (inline asm, invisible to it) and stall on it
#pragma unroll
        for (int i = 0; i < KS; ++i) asm volatile("" : : "v"(pfB[i]));
#pragma unroll
        for (int i = 0; i < 72; ++i) asm volatile("" : : "v"(AB[i]));
    }
    f32x4 gpf[KT], gA[12][2];
    if constexpr (MODE != 0) {
#pragma unroll
        for (int i = 0; i < KT; ++i) gpf[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 12; ++i) gA[i][0] = gA[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    const int vs = blockIdx.y;
    const int grp0 = (int)((long long)a.NG * vs / a.vsplit), grp1 = (int)((long long)a.NG * (vs + 1) / a.vsplit);
    // model stream: one blob = 42 pieces of 1 KiB, piece i moved by wave i % 4 (LDS destination = wave-uniform base + lane * 16)
    // (inline asm, as in the distance engine: when hipcc sees the builtin it waits for the whole fetch -- vmcnt(0) -- in front of
    // the next LDS read, whatever buffer that reads; the explicit waits below are what order fetch and use)
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) float*)smem);
    const uint32_t lane16 = (uint32_t)lane * 16u;
    auto dma = [&](int grp, int buf) {
        for (int i = wave; i < BLOB / 256; i += 4) {
            const float* src = a.blob + (size_t)grp * BLOB + (size_t)i * 256;
            const uint32_t dst = lds_base + (uint32_t)(buf * BLOB + i * 256) * 4u;
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(lane16), "s"(src), "s"(dst) : "memory", "m0");
        }
    };
    // The forward contractions of a group (pose blend shapes, skinning transforms: MFMA only) and the VALU section that
    // turns them into vertices and vertex gradients are independent ACROSS groups, so the loop is software-pipelined:
    // iteration `grp` issues the forward MFMAs of group grp + 1 next to the VALU section of group grp, then the reverse
    // MFMAs of group grp -- the matrix pipe no longer idles through ~300 VALU instructions per group.  Three blob buffers:
    // `grp` (reverse operands), `grp + 1` (forward operands), `grp + 2` in flight.
    auto forward_mfma = [&](const float* Bf, f32x4 (&off)[3], f32x4 (&Tm)[12]) {
        // (every contraction walks its independent accumulator chains round-robin: a 16x16x4 fp32 MFMA that accumulates
        // onto the result of the one issued just before it waits ~8 cycles beyond its 32 of issue)
#pragma unroll
        for (int c3 = 0; c3 < 3; ++c3) off[c3] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int c3 = 0; c3 < 3; ++c3) off[c3] = mfma4(Bf[c3 * C_STRIDE + ks * 64 + lane], pfB[ks], off[c3]);
#pragma unroll
        for (int e = 0; e < 12; ++e) Tm[e] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) {
            const float wv = Bf[PNDF_LBS_BLOB_W + ks * 64 + lane];
#pragma unroll
            for (int e = 0; e < 12; ++e) Tm[e] = mfma4(wv, AB[e * 6 + ks], Tm[e]);
        }
    };
    f32x4 off[3], Tm[12];          // pose-blend offsets (rows = vertices 4 g + r, columns = frames) and T = sum_j W[v, j] A_j
    if (grp0 < grp1) {
        dma(grp0, 0);
        if (grp0 + 1 < grp1) dma(grp0 + 1, 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        forward_mfma(smem, off, Tm);
    }
    for (int grp = grp0; grp < grp1; ++grp) {
        const int k = grp - grp0;
        const float* Bf = smem + (k % 3) * BLOB;
        f32x4 off_n[3], Tm_n[12];
        if (grp + 1 < grp1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of blob grp + 1 have landed ...
            __syncthreads();                                      // ... everyone's have, and everyone has left buffer (k + 2) % 3
            if (grp + 2 < grp1) dma(grp + 2, (k + 2) % 3);
            forward_mfma(smem + ((k + 1) % 3) * BLOB, off_n, Tm_n);
        }
        const i32x4 fl = *(const i32x4*)(Bf + PNDF_LBS_BLOB_FL + 4 * g);
        f32x4 vp[3], V[3];
#pragma unroll
        for (int c3 = 0; c3 < 3; ++c3) vp[c3] = *(const f32x4*)(Bf + PNDF_LBS_BLOB_VS + c3 * GV + 4 * g) + off[c3];
#pragma unroll
        for (int a3 = 0; a3 < 3; ++a3) V[a3] = Tm[3 * a3] * vp[0] + Tm[3 * a3 + 1] * vp[1] + Tm[3 * a3 + 2] * vp[2] + Tm[9 + a3];

        if constexpr (MODE == 0) {
            if (t_ok) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int v = grp * GV + 4 * g + r;
                    if (v < a.V) {
                        if (a.verts) {
                            float* dst = a.verts + ((size_t)n * a.V + v) * 3;
                            dst[0] = V[0][r]; dst[1] = V[1][r]; dst[2] = V[2][r];
                        }
                        if (fl[r] >= 0 && a.joints) {
                            float* dst = a.joints + ((size_t)n * njt + NJ + fl[r]) * 3;
                            dst[0] = V[0][r]; dst[1] = V[1][r]; dst[2] = V[2][r];
                        }
                    }
                }
            }
            lbs_rotate(off, Tm, off_n, Tm_n);
            continue;
        }

        // ---- d L / d vertices of the tile
        f32x4 gV[3];
        if constexpr (MODE == 1) {
            // temporal term: frame p + 1 is the next lane of the 16-lane row.  No epsilon under the root, as in the
            // reference (motion_denoise.py:89): two identical consecutive vertices give NaN there and here.
            f32x4 d[3], u[3];
#pragma unroll
            for (int a3 = 0; a3 < 3; ++a3)
#pragma unroll
                for (int r = 0; r < 4; ++r) d[a3][r] = V[a3][r] - next_frame(V[a3][r]);
            const f32x4 n2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float inv = a.w_temp * __builtin_amdgcn_rsqf(n2[r]);      // 0 -> inf -> 0 * inf = NaN, as d / sqrt(0)
                const bool ok = pair_ok && fl[r] != -2;
#pragma unroll
                for (int a3 = 0; a3 < 3; ++a3) u[a3][r] = ok ? d[a3][r] * inv : 0.f;
            }
#pragma unroll
            for (int a3 = 0; a3 < 3; ++a3)
#pragma unroll
                for (int r = 0; r < 4; ++r) gV[a3][r] = u[a3][r] - prev_frame(u[a3][r]);      // lane 0: no pair inside this chunk
            // data term on the joints that are picked from vertices (the 24 chain joints: pndf_lbs_pose_backward_kernel)
            if (a.it_gt0 && owned) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (fl[r] >= 0) {
                        const float* j0 = a.joints0 + ((size_t)n * njt + NJ + fl[r]) * 3;
                        const float dx = V[0][r] - j0[0], dy = V[1][r] - j0[1], dz = V[2][r] - j0[2];
                        const float inv = a.w_data / sqrtf(dx * dx + dy * dy + dz * dz);
                        gV[0][r] += dx * inv; gV[1][r] += dy * inv; gV[2][r] += dz * inv;
                    }
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int v = grp * GV + 4 * g + r;
                const bool ok = t_ok && v < a.V;
#pragma unroll
                for (int a3 = 0; a3 < 3; ++a3) {
                    float gv = (ok && a.g_verts) ? a.g_verts[((size_t)n * a.V + v) * 3 + a3] : 0.f;
                    if (ok && fl[r] >= 0 && a.g_joints) gv += a.g_joints[((size_t)n * njt + NJ + fl[r]) * 3 + a3];
                    gV[a3][r] = gv;
                }
            }
        }

        // ---- reverse of the skinning: d L / d v_posed = T_R^T gV
        f32x4 gvp[3];
#pragma unroll
        for (int b = 0; b < 3; ++b) gvp[b] = Tm[b] * gV[0] + Tm[3 + b] * gV[1] + Tm[6 + b] * gV[2];
        // ---- d L / d pose_feature[k] += sum_{v, comp} P[comp][k][v] gvp[comp][v]: rows = k, contraction = the tile's vertices
        {
            const float* Pt = Bf + p * GV + 4 * g;
#pragma unroll
            for (int c3 = 0; c3 < 3; ++c3) {
                f32x4 w[KT];
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) w[kt] = *(const f32x4*)(Pt + c3 * C_STRIDE + kt * 256);
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int kt = 0; kt < KT; ++kt) gpf[kt] = mfma4(w[kt][r], gvp[c3][r], gpf[kt]);
            }
        }
        // ---- d L / d A[j][entry] += sum_v W[v, j] gV (x) [v_posed, 1]: rows = joints
        {
            const float* Wt = Bf + PNDF_LBS_BLOB_W + p * GV + 4 * g;
            const f32x4 w0 = *(const f32x4*)Wt, w1 = *(const f32x4*)(Wt + 256);
            f32x4 X[12];
#pragma unroll
            for (int e = 0; e < 12; ++e) X[e] = (e < 9) ? gV[e / 3] * vp[e % 3] : gV[e - 9];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int e = 0; e < 12; ++e) {
                    gA[e][0] = mfma4(w0[r], X[e][r], gA[e][0]);
                    gA[e][1] = mfma4(w1[r], X[e][r], gA[e][1]);
                }
        }
        lbs_rotate(off, Tm, off_n, Tm_n);
    }
    if constexpr (MODE != 0) {
        if (t_ok) {
            float* o_pf = owned ? a.gpf + ((size_t)vs * N + n) * PF : a.halo_pf + ((size_t)vs * nch + cid) * PF;
            float* o_A = owned ? a.gA + ((size_t)vs * N + n) * A_FLOATS : a.halo_A + ((size_t)vs * nch + cid) * A_FLOATS;
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) *(f32x4*)(o_pf + 16 * kt + 4 * g) = gpf[kt];
#pragma unroll
            for (int e = 0; e < 12; ++e) {
                *(f32x4*)(o_A + e * 32 + 4 * g) = gA[e][0];
                *(f32x4*)(o_A + e * 32 + 16 + 4 * g) = gA[e][1];
            }
        }
    }
}

extern "C" __global__ void __launch_bounds__(256, 1) pndf_lbs_vertex_forward_kernel(PndfLbsArgs a) { lbs_vertex_body<0>(a); }
extern "C" __global__ void __launch_bounds__(256, 1) pndf_lbs_vertex_terms_kernel(PndfLbsArgs a) { lbs_vertex_body<1>(a); }
extern "C" __global__ void __launch_bounds__(256, 1) pndf_lbs_vertex_reverse_kernel(PndfLbsArgs a) { lbs_vertex_body<2>(a); }

// ------------------------------------------------------------------ partial results -> one value-major array
// The vertex kernels leave d L / d pose_feature and d L / d A per vertex range (and once more for the frame a chunk shares
// with the next one), frame-major.  The per-frame reverse kernels run one THREAD per frame; reading those rows directly,
// every load of a wave touched 64 cache lines (stride 208 / 384 floats between lanes): 0.51 ms for 19,200 frames, a fifth
// of the whole fused pass, all of it latency.  This kernel sums the ranges and halos in their fixed order (the same order
// as before: bit-identical sums) and TRANSPOSES through LDS: read with lanes along the values of a frame, written with lanes
// along the frames, so that the reverse kernels' loads are one line per wave.
constexpr int RED_PF = 9 * (NJ - 1);               // 207 pose-feature rows, then 12 x 24 transform-entry rows
constexpr int RED_ROWS = RED_PF + 12 * NJ;
extern "C" __global__ void __launch_bounds__(256) pndf_lbs_reduce_partials_kernel(PndfLbsArgs a) {
    __shared__ float tile[64][65];
    const long long N = (long long)a.S * a.T;
    const long long f0 = (long long)blockIdx.x * 64;
    const int r0 = blockIdx.y * 64, tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int nch = a.S * a.cps;
    const int r = r0 + tx;
    const bool pf = r < RED_PF;
    const int q = r - RED_PF;
    const int off = pf ? r : (q / NJ) * 32 + (q % NJ);
    const int stride = pf ? PF : A_FLOATS;
    const float* base = pf ? a.gpf : a.gA;
    const float* hbase = pf ? a.halo_pf : a.halo_A;
#pragma unroll 4
    for (int it = 0; it < 16; ++it) {
        const int fl = ty + 4 * it;
        const long long n = f0 + fl;
        float acc = 0.f;
        if (n < N && r < RED_ROWS) {
            for (int v = 0; v < a.vsplit; ++v) acc += base[((size_t)v * N + n) * stride + off];
            const int s = (int)(n / a.T), t = (int)(n - (long long)s * a.T);
            if (hbase && t > 0 && t % 15 == 0 && t / 15 < a.cps) {
                const long long hidx = (long long)s * a.cps + t / 15 - 1;
                for (int v = 0; v < a.vsplit; ++v) acc += hbase[((size_t)v * nch + hidx) * stride + off];
            }
        }
        tile[tx][fl] = acc;
    }
    __syncthreads();
#pragma unroll 4
    for (int it = 0; it < 16; ++it) {
        const int rl = ty + 4 * it;
        const long long n = f0 + tx;
        if (n < N && r0 + rl < RED_ROWS) a.red[(size_t)(r0 + rl) * N + n] = tile[rl][tx];
    }
}

// ------------------------------------------------------------------ per-frame reverse
template <class Tree>
__device__ __forceinline__ void lbs_pose_backward_body(const PndfLbsArgs& a) {
    const long long N = (long long)a.S * a.T;
    const long long n = (long long)blockIdx.x * 64 + threadIdx.x;
    if (n >= N) return;
    const int T = a.T, s = (int)(n / T), t = (int)(n - (long long)s * T), njt = NJ + a.NE;
    const int nch = a.S * a.cps;
    float R[NJ][9], GR[NJ][9], Gt[NJ][3];
    frame_transforms<Tree>(a.theta + n * 69, a.model, R, GR, Gt);
    // the frame a chunk of pairs shares with the next chunk got a second partial result there
    const bool halo = a.halo_pf && t > 0 && t % 15 == 0 && t / 15 < a.cps;
    const long long hidx = (long long)s * a.cps + t / 15 - 1;
    // (summed over vertex ranges and halos by pndf_lbs_reduce_partials_kernel: value-major, one line per wave and load)
    auto sum_pf = [&](int k) __attribute__((always_inline)) { return a.red[(size_t)k * N + n]; };
    auto sum_A = [&](int e, int j) __attribute__((always_inline)) { return a.red[(size_t)(RED_PF + e * NJ + j) * N + n]; };
    float gGR[NJ][9], gGt[NJ][3], gR[NJ][9];
    for_joints<Tree, 0>([&](int j) __attribute__((always_inline)) {
        float gAt[3];
#pragma unroll
        for (int e = 0; e < 3; ++e) gAt[e] = sum_A(9 + e, j);
#pragma unroll
        for (int e = 0; e < 9; ++e) gGR[j][e] = sum_A(e, j) - gAt[e / 3] * a.model.J[j][e % 3];      // A_t = G_t - G_R J
        // joints[:, :24] = G_t: data term (motion_denoise.py:93-94) or the caller's d L / d joints
        float gj[3] = {0.f, 0.f, 0.f};
        if (a.g_joints) {
#pragma unroll
            for (int e = 0; e < 3; ++e) gj[e] = a.g_joints[((size_t)n * njt + j) * 3 + e];
        } else if (a.it_gt0 && a.joints0) {
            const float* j0 = a.joints0 + ((size_t)n * njt + j) * 3;
            const float dx = Gt[j][0] - j0[0], dy = Gt[j][1] - j0[1], dz = Gt[j][2] - j0[2];
            const float inv = a.w_data / sqrtf(dx * dx + dy * dy + dz * dz);
            gj[0] = dx * inv; gj[1] = dy * inv; gj[2] = dz * inv;
        }
#pragma unroll
        for (int e = 0; e < 3; ++e) gGt[j][e] = gAt[e] + gj[e];
#pragma unroll
        for (int e = 0; e < 9; ++e) gR[j][e] = (j > 0) ? sum_pf(9 * (j - 1) + e) : 0.f;                // pose_feature = R_1.. - I
    });
    for_joints_reverse<Tree, 1>([&](int i) __attribute__((always_inline)) {          // reverse of G_i = G_parent [R_i | rel_i]
        const int pj = Tree::parent(a.model, i);
#pragma unroll
        for (int x = 0; x < 3; ++x)
#pragma unroll
            for (int y = 0; y < 3; ++y) {
                float acc = 0.f, acc2 = 0.f;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    acc += GR[pj][3 * k + x] * gGR[i][3 * k + y];          // G_R[p]^T gG_R[i]
                    acc2 += gGR[i][3 * x + k] * R[i][3 * y + k];           // gG_R[i] R_i^T
                }
                gR[i][3 * x + y] += acc;
                gGR[pj][3 * x + y] += acc2 + gGt[i][x] * a.model.rel[i][y];
            }
#pragma unroll
        for (int e = 0; e < 3; ++e) gGt[pj][e] += gGt[i][e];
    });
    for_joints<Tree, 1>([&](int j) __attribute__((always_inline)) {
        float gr[3];
        const float* th = a.theta + n * 69 + 3 * (j - 1);
        rodrigues_vjp(th[0], th[1], th[2], gR[j], gr);
#pragma unroll
        for (int e = 0; e < 3; ++e) a.g_theta[n * 69 + 3 * (j - 1) + e] = gr[e];
    });
}
extern "C" __global__ void __launch_bounds__(64) pndf_lbs_pose_backward_kernel(PndfLbsArgs a) { lbs_pose_backward_body<ModelTree>(a); }

// The same for SMPL's own tree, laid out for registers: the generic form above keeps ~1,000 values alive (R, G_R, G_t and
// their gradients for 24 joints), which does not fit even with the tree as constants.  Here a joint's rotation is recovered
// from the stored global rotations (R_i = G_R[p]^T G_R[i]), its partial results are read when the reverse sweep reaches
// it, d L / d R_i goes through the Rodrigues reverse pass at once, and what a joint hands to its parent lives in a pending
// slot that exists only between the parent's last child and the parent itself (a handful at a time): ~300 live values.
constexpr bool smpl_last_child(int i) {      // no later joint has the same parent: the reverse sweep meets this child first
    for (int k = i + 1; k < NJ; ++k)
        if (SmplTree::tab[k] == SmplTree::tab[i]) return false;
    return true;
}
constexpr bool smpl_has_children(int i) {
    for (int k = i + 1; k < NJ; ++k)
        if (SmplTree::tab[k] == i) return true;
    return false;
}
extern "C" __global__ void __launch_bounds__(64) pndf_lbs_pose_backward_smpl_kernel(PndfLbsArgs a) {
    const long long N = (long long)a.S * a.T;
    const long long n = (long long)blockIdx.x * 64 + threadIdx.x;
    if (n >= N) return;
    const int T = a.T, s = (int)(n / T), t = (int)(n - (long long)s * T), njt = NJ + a.NE;
    const int nch = a.S * a.cps;
    float GR[NJ][9], Gt[NJ][3];
    {
        float R[NJ][9];
        frame_transforms<SmplTree>(a.theta + n * 69, a.model, R, GR, Gt);
    }
    const bool halo = a.halo_pf && t > 0 && t % 15 == 0 && t / 15 < a.cps;
    const long long hidx = (long long)s * a.cps + t / 15 - 1;
    // partial results of joint i, summed over the vertex ranges (and the chunk that shares this frame) in a fixed order
    auto gather = [&](int i, float (&sA)[12], float (&sP)[9]) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < 12; ++e) sA[e] = a.red[(size_t)(RED_PF + e * NJ + i) * N + n];
#pragma unroll
        for (int e = 0; e < 9; ++e) sP[e] = (i >= 1) ? a.red[(size_t)(9 * (i - 1) + e) * N + n] : 0.f;
    };
    float pGR[NJ][9], pGt[NJ][3];      // pending: what a joint's children handed up (static indices: only the live ones cost registers)
    static_for<NJ>([&](auto ic) __attribute__((always_inline)) {
        constexpr int i = NJ - 1 - decltype(ic)::value;
        float sA[12], sP[9], gAt[3], gGR[9], gGt[3];
        gather(i, sA, sP);
#pragma unroll
        for (int e = 0; e < 3; ++e) gAt[e] = sA[9 + e];
#pragma unroll
        for (int e = 0; e < 9; ++e) gGR[e] = sA[e] - gAt[e / 3] * a.model.J[i][e % 3];      // A_t = G_t - G_R J
        float gj[3] = {0.f, 0.f, 0.f};      // joints[:, :24] = G_t: data term (motion_denoise.py:93-94) or the caller's d L / d joints
        if (a.g_joints) {
#pragma unroll
            for (int e = 0; e < 3; ++e) gj[e] = a.g_joints[((size_t)n * njt + i) * 3 + e];
        } else if (a.it_gt0 && a.joints0) {
            const float* j0 = a.joints0 + ((size_t)n * njt + i) * 3;
            const float dx = Gt[i][0] - j0[0], dy = Gt[i][1] - j0[1], dz = Gt[i][2] - j0[2];
            const float inv = a.w_data / sqrtf(dx * dx + dy * dy + dz * dz);
            gj[0] = dx * inv; gj[1] = dy * inv; gj[2] = dz * inv;
        }
#pragma unroll
        for (int e = 0; e < 3; ++e) gGt[e] = gAt[e] + gj[e];
        if constexpr (smpl_has_children(i)) {
#pragma unroll
            for (int e = 0; e < 9; ++e) gGR[e] += pGR[i][e];
#pragma unroll
            for (int e = 0; e < 3; ++e) gGt[e] += pGt[i][e];
        }
        if constexpr (i >= 1) {
            constexpr int pj = SmplTree::tab[i];
            float Ri[9], gRi[9];
#pragma unroll
            for (int x = 0; x < 3; ++x)
#pragma unroll
                for (int y = 0; y < 3; ++y) {
                    float r = 0.f, acc = 0.f;
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        r += GR[pj][3 * k + x] * GR[i][3 * k + y];             // R_i = G_R[p]^T G_R[i]
                        acc += GR[pj][3 * k + x] * gGR[3 * k + y];            // G_R[p]^T gG_R[i]
                    }
                    Ri[3 * x + y] = r;
                    gRi[3 * x + y] = sP[3 * x + y] + acc;                     // pose_feature = R_1.. - I
                }
            // d L / d R_i takes the place of this frame's (first) partial d L / d pose_feature[9 (i - 1) ..]: the Rodrigues reverse
            // pass is its own kernel, one thread per (frame, joint)
            float* outR = a.gpf + (size_t)n * PF + 9 * (i - 1);
#pragma unroll
            for (int e = 0; e < 9; ++e) outR[e] = gRi[e];
            // to the parent: gG_R[i] R_i^T + gG_t[i] (x) rel_i, and gG_t[i]
#pragma unroll
            for (int x = 0; x < 3; ++x) {
#pragma unroll
                for (int y = 0; y < 3; ++y) {
                    float acc2 = gGt[x] * a.model.rel[i][y];
#pragma unroll
                    for (int k = 0; k < 3; ++k) acc2 += gGR[3 * x + k] * Ri[3 * y + k];
                    pGR[pj][3 * x + y] = smpl_last_child(i) ? acc2 : pGR[pj][3 * x + y] + acc2;
                }
                pGt[pj][x] = smpl_last_child(i) ? gGt[x] : pGt[pj][x] + gGt[x];
            }
        }
    });
}

// d L / d R_i [S*T, 23, 9] (in the first partial-result slice, see above) -> d L / d theta: one thread per (frame, joint)
extern "C" __global__ void __launch_bounds__(256) pndf_lbs_rodrigues_vjp_kernel(PndfLbsArgs a) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)a.S * a.T * (NJ - 1)) return;
    const long long n = idx / (NJ - 1);
    const int j = (int)(idx - n * (NJ - 1));
    const float* gRp = a.gpf + (size_t)n * PF + 9 * j;
    float gRi[9], gr[3];
#pragma unroll
    for (int e = 0; e < 9; ++e) gRi[e] = gRp[e];
    const float* th = a.theta + n * 69 + 3 * j;
    rodrigues_vjp(th[0], th[1], th[2], gRi, gr);
#pragma unroll
    for (int e = 0; e < 3; ++e) a.g_theta[n * 69 + 3 * j + e] = gr[e];
}

// ====================================================================================== split precision
// The same three stages with the two vertex-side contractions on v_mfma_f32_16x16x32_f16 (16 cycles for 16 x 16 x 32 MACs
// against 32 cycles for 16 x 16 x 4 on the fp32 pipe), every fp32 operand carried as fp16 hi + fp16 lo and every product
// as hi hi + hi lo + lo hi with fp32 accumulation (dropped lo lo term: 2^-22 relative) -- the arithmetic of the distance
// engine's f16x3 kernels (pndf_kernel_split.hip).  Operands are scaled by powers of two so that they sit high in the fp16
// range without leaving it: the model's by the packer (p_scale, w_scale from the largest |entry|), the pose feature by 2^12
// (|R - I| <= 2), the joint transforms by a_scale (from the skeleton's extent), the reverse operands by g_scale / x_scale
// from a-priori bounds the host derives from the term weights (|d L / d verts| <= 2 w_temp + w_data) and the model.
// Per (16 vertex x 16 frame) tile: 63 + 36 MFMAs forward, 65 + 48 reverse = 3,392 matrix-pipe cycles against 15,360.
//   * ONE copy of the model in LDS serves both contractions (pndf_lbs_split.h): planes [row][16 v] of halfs; the reverse
//     pass (contraction over vertices) reads 4 v of a row per lane, the forward pass (contraction over rows) gets its
//     operand -- one vertex per lane, four consecutive rows -- from ds_read_b64_tr_b16.
//   * reverse k-blocks: a tile has 16 vertices, a k-block 32 slots.  d L / d pose_feature contracts (component, vertex):
//     components 0, 1 fill one k-block (three MFMAs), component 2 half a k-block -- its other half carries the LO half of
//     the gradient against the same (duplicated) weights, so hi hi + hi lo is ONE MFMA and lo hi the second.  The same
//     trick serves d L / d A (contraction over the 16 vertices only): two MFMAs per (entry, joint tile).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int KB = PNDF_LBS_KB, SBB = PNDF_LBS_SB_BYTES, PLANE = PNDF_LBS_SB_PLANE;
// (defaults: pndf_experiment.h)  PNDF_LBS_DIAG: timing diagnostics (WRONG results): 1 = no wait for the model fetch, 2 = no fetch,
// 4 = no barrier, 8 = no operand splits in the reverse pass, 16 / 32 = no forward / reverse tile reads after the first, 64 = no reverse MFMAs
// PNDF_LBS_FLA / PNDF_LBS_RLA: forward steps / reverse row tiles whose LDS reads are in flight ahead of the MFMAs that use them

__device__ __forceinline__ f32x4 mf16(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
// (a, b) -> packed fp16 pairs: hi = rtz(a, b), lo = rne(a - hi_a, b - hi_b); the remainders are exact in fp32
__device__ __forceinline__ void lbs_split2(float a, float b, unsigned& hi, unsigned& lo) {
    const unsigned hp = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(a, b));
    hi = hp;
    // (fma in fp32, the result rounded to fp16 straight into one half of the register: the same bits as v_fma_mix_f32 +
    // v_cvt_pk_f16_f32 for every input, tools/ubench/split_probe.hip -- three instructions per pair instead of four)
    unsigned l;
    // (one statement: between two, hipcc pads a wait state it cannot know to be unnecessary)
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=&v"(l) : "v"(hp), "v"(a), "v"(b));
    lo = l;
}
__device__ __forceinline__ void lbs_split4(const f32x4& v, f16x4& hi, f16x4& lo) {
    unsigned h0, l0, h1, l1;
    lbs_split2(v[0], v[1], h0, l0);
    lbs_split2(v[2], v[3], h1, l1);
    hi = __builtin_bit_cast(f16x4, u32x2{h0, h1});
    lo = __builtin_bit_cast(f16x4, u32x2{l0, l1});
}
__device__ __forceinline__ f16x8 cat(f16x4 a, f16x4 b) { return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7); }
// (static_for: see the top of the file -- `#pragma unroll` leaves the 99-trip loop over the forward MFMAs rolled)

// ds_read_b64_tr_b16: within a 16-lane row, lane i element j <- element i % 4 of the 8 bytes addressed by lane 4 j + i / 4
// (profiles/r02/tr_b16_probe.txt).  With lane m of row g pointing at plane row 4 g + m / 4, halfs 4 (m % 4) .. + 3, lane i
// receives vertex i of rows 4 g .. 4 g + 3.
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
__device__ __forceinline__ f16x4 lds_tr(const char* p) {
    return __builtin_bit_cast(f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p));
}
// One ds_read_b64 per row read, through LDS pointers whose lane offset is opaque to the compiler (one base register per
// plane).  (hipcc still pairs the reads of two tiles off one base into ds_read2st64_b64, served in 16-lane groups over 32
// banks at 128 B / clk: the tile layout of pndf_lbs_split.h is free of bank conflicts for that form.)
typedef __attribute__((address_space(3))) const char lds_char;
typedef __attribute__((address_space(3))) const f16x4 lds_f16x4;
__device__ __forceinline__ lds_char* lds_opaque(lds_char* p) {
    unsigned v = (unsigned)(size_t)p;
    asm volatile("" : "+v"(v));
    return (lds_char*)(size_t)v;
}
__device__ __forceinline__ f16x4 lds_row(lds_char* p) { return *(lds_f16x4*)p; }

}  // namespace

// ------------------------------------------------------------------ per-frame forward, split operands out
template <class Tree>
__device__ __forceinline__ void lbs_pose_split_body(const PndfLbsSplitArgs& sa) {
    const PndfLbsArgs& a = sa.base;
    const long long n = (long long)blockIdx.x * 64 + threadIdx.x;
    if (n >= (long long)a.S * a.T) return;
    float R[NJ][9], GR[NJ][9], Gt[NJ][3];
    frame_transforms<Tree>(a.theta + n * 69, a.model, R, GR, Gt);
    // pose feature x 2^12, hi / lo, in B-operand order: element i of k-block kb in lane group g = entry 32 kb + 16 (i / 4) + 4 g + i % 4
    auto pf_at = [&](int k) __attribute__((always_inline)) {
        return (k < 9 * (NJ - 1)) ? (R[1 + k / 9][k % 9] - (((k % 9) % 4 == 0) ? 1.0f : 0.0f)) * PNDF_LBS_PF_SCALE : 0.f;
    };
    u32x4* dpf = (u32x4*)sa.pfs + n * (4 * KB * 2);
    auto pf_block = [&](int blk) __attribute__((always_inline)) {      // blk = g * KB + kb
        const int g = blk / KB, kb = blk % KB;
        unsigned h[4], l[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k0 = kb * 32 + 16 * (q / 2) + 4 * g + 2 * (q % 2);
            lbs_split2(pf_at(k0), pf_at(k0 + 1), h[q], l[q]);
        }
        dpf[blk * 2] = u32x4{h[0], h[1], h[2], h[3]};
        dpf[blk * 2 + 1] = u32x4{l[0], l[1], l[2], l[3]};
    };
    if constexpr (Tree::STATIC) static_for<4 * KB>([&](auto bc) __attribute__((always_inline)) { pf_block(decltype(bc)::value); });
    else for (int blk = 0; blk < 4 * KB; ++blk) pf_block(blk);
    // A_j = [G_R | G_t - G_R J_j] x a_scale: element i of entry e in lane group g = joint 16 (i / 4) + 4 g + i % 4
    auto A_at = [&](int e, int j) __attribute__((always_inline)) {
        if (j >= NJ) return 0.f;
        if (e < 9) return GR[j][e] * sa.a_scale;
        const int c = e - 9;
        return (Gt[j][c] - (GR[j][3 * c] * a.model.J[j][0] + GR[j][3 * c + 1] * a.model.J[j][1] + GR[j][3 * c + 2] * a.model.J[j][2])) * sa.a_scale;
    };
    u32x4* dA = (u32x4*)sa.Aps + n * (4 * 12 * 2);
    auto A_block = [&](int blk) __attribute__((always_inline)) {      // blk = g * 12 + e
        const int g = blk / 12, e = blk % 12;
        unsigned h[4], l[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j0 = 16 * (q / 2) + 4 * g + 2 * (q % 2);
            lbs_split2(A_at(e, j0), A_at(e, j0 + 1), h[q], l[q]);
        }
        dA[blk * 2] = u32x4{h[0], h[1], h[2], h[3]};
        dA[blk * 2 + 1] = u32x4{l[0], l[1], l[2], l[3]};
    };
    if constexpr (Tree::STATIC) static_for<48>([&](auto bc) __attribute__((always_inline)) { A_block(decltype(bc)::value); });
    else for (int blk = 0; blk < 48; ++blk) A_block(blk);
    const int njt = NJ + a.NE;
    for_joints<Tree, 0>([&](int j) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            if (a.Gt) a.Gt[n * (NJ * 3) + 3 * j + e] = Gt[j][e];
            if (a.joints) a.joints[(n * njt + j) * 3 + e] = Gt[j][e];
        }
    });
}
extern "C" __global__ void __launch_bounds__(64) pndf_lbs_pose_split_kernel(PndfLbsSplitArgs sa) { lbs_pose_split_body<ModelTree>(sa); }
extern "C" __global__ void __launch_bounds__(64) pndf_lbs_pose_split_smpl_kernel(PndfLbsSplitArgs sa) { lbs_pose_split_body<SmplTree>(sa); }

// ------------------------------------------------------------------ per (16 vertices x 16 frames) tile, split precision
template <int MODE>      // 0: vertices / vertex-picked joints out   1: the two fused terms and their reverse pass
__device__ __forceinline__ void lbs_vertex_split_body(const PndfLbsSplitArgs& sa) {
    static_assert(MODE == 0 || MODE == 1, "the general reverse pass (arbitrary d L / d verts: no a-priori bound) stays on fp32");
    const PndfLbsArgs& a = sa.base;
    extern __shared__ __attribute__((aligned(16))) char smem_s[];      // 3 x SBB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, p = lane & 15;
    constexpr int STRIDE = (MODE == 1) ? 15 : 16;
    const int T = a.T, cps = a.cps, nch = a.S * cps, njt = NJ + a.NE;
    // 1-D grid, vertex range fastest (workgroup i runs on XCD i % 8: with a split of 8 every XCD streams one eighth of the model)
    const int vs = (int)(blockIdx.x % (unsigned)a.vsplit);
    int cid = (int)(blockIdx.x / (unsigned)a.vsplit) * 4 + wave;
    const bool wave_on = cid < nch;
    if (!wave_on) cid = nch - 1;
    const int s = cid / cps, c = cid - s * cps, f0 = c * STRIDE;
    const int t = f0 + p;
    const bool t_ok = wave_on && t < T;
    const long long n = (long long)s * T + (t < T ? t : T - 1);
    const bool owned = t_ok && (MODE != 1 || p < 15 || t == T - 1);
    const bool pair_ok = MODE == 1 && t_ok && p < 15 && t + 1 < T;
    const long long N = (long long)a.S * T;

    // ---- B operands of this wave's 16 frames
    f16x8 pfh[KB], pfl[KB], Ah[12], Al[12];
    {
        const f16x8* src = (const f16x8*)sa.pfs + ((size_t)n * 4 + g) * (KB * 2);
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            pfh[kb] = src[2 * kb];
            pfl[kb] = src[2 * kb + 1];
        }
        const f16x8* srcA = (const f16x8*)sa.Aps + ((size_t)n * 4 + g) * 24;
#pragma unroll
        for (int e = 0; e < 12; ++e) {
            Ah[e] = srcA[2 * e];
            Al[e] = srcA[2 * e + 1];
        }
        // make hipcc wait for these loads HERE: left to the first use, its vmcnt(N) waits sit inside the group loop, where they
        // also count the model fetch (inline asm, invisible to it) and stall on it
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) asm volatile("" : : "v"(pfh[kb]), "v"(pfl[kb]));
#pragma unroll
        for (int e = 0; e < 12; ++e) asm volatile("" : : "v"(Ah[e]), "v"(Al[e]));
    }
    f32x4 gpf[KT], gA[12][2];
    if constexpr (MODE != 0) {
#pragma unroll
        for (int i = 0; i < KT; ++i) gpf[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 12; ++i) gA[i][0] = gA[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    const int grp0 = (int)((long long)a.NG * vs / a.vsplit), grp1 = (int)((long long)a.NG * (vs + 1) / a.vsplit);
    const int lane_tr = 8 * pndf_lbs_sb_unit(4 * g + (p >> 2), p & 3);      // transposed reads: tile row 4 g + p / 4, vertex quad p % 4
    const int lane_rv = 8 * pndf_lbs_sb_unit(p, g);                         // row reads: tile row p, vertex quad g
    // One wave per SIMD, in-order issue.  Two rules shape a group (tools/ubench/mfma_valu_overlap.hip, profiles/r03):
    //   * an MFMA takes 4 cycles of issue and 16 of the matrix pipe; the ~3 instructions that sit DIRECTLY behind it in
    //     program order run in its shadow, anything behind a second MFMA does not (the wave stalls at that MFMA until the
    //     pipe is free): 12 MFMAs + 24 VALU instructions take 105 ns one-behind-one, 133 ns as MMM vvvvvv, 141 ns as blocks;
    //   * LDS latency is hidden only by reads issued ahead in program order.
    // Left alone hipcc does neither (tile reads right in front of their use, the VALU section as a block), so a group is laid
    // out by hand, one MFMA at a time, pinned with sched_barriers -- the method of the distance engine's chunk epilogues
    // (pndf_kernel_split.hip).  Behind every MFMA: at most two tile reads for a later MFMA, one piece of the model fetch
    // (global_load_lds, the first MFMAs of the forward pass only: issued as a burst the 45 pieces stall all four waves for
    // ~0.6 us per group), and a piece of two to four VALU instructions:
    //   forward pass of group grp + 1 (63 + 36 MFMAs) <- the VALU section of group grp (v_posed, V, the temporal term);
    //   reverse pass of group grp: d L / d A (48 MFMAs) <- the operand split of the next entry and T_R^T g; d L / d
    //   pose_feature (65 MFMAs, from entry 6 on two row tiles behind every entry) <- row reads RLA tiles ahead and the hand-over
    //   of the next group's accumulators.
    struct PTile { f16x4 h0, h1, l0, l1; };
    struct RTile { f16x4 c0h, c1h, c2h, c0l, c1l, c2l; };
    constexpr int FSTEPS = 3 * KB, FMFMA = 3 * FSTEPS + 36, FLA = PNDF_LBS_FLA, RLA = PNDF_LBS_RLA;
    constexpr int NP1 = (MODE == 0) ? 22 : 47;      // pieces of the VALU section
    struct GState {
        i32x4 fl;
        f32x4 vs[3], vp[3], V[3], d[3], n2, inv, u[3], gV[3];
    };
    // model fetch: piece j (0 .. 11) of this wave -- KiB 11 wave + j of the 45 of blob `grp` -- into buffer `buf`; the twelfth
    // piece exists for wave 0 only (KiB 44): the other waves fetch it as well (same bytes) rather than branch
    constexpr int DMA_PIECES = (SBB / 1024 + 3) / 4;
    static_assert(SBB / 1024 == 45 && DMA_PIECES == 12, "pieces of the model fetch");
    // (inline asm, as in the distance engine: when hipcc sees the builtin it waits for the fetch -- vmcnt(0) -- in front of
    // the next LDS read, whatever buffer that reads; the explicit wait at the top of a group is what orders fetch and use)
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(lds_char*)smem_s);
    const uint32_t lane16 = (uint32_t)lane * 16u;
    // A wave's eleven pieces are consecutive KiB (11 wave .. 11 wave + 10): four pieces share one source pointer and one M0,
    // the instruction offset moves both addresses (as the distance engine's ring does, pndf_device.h) -- two scalar
    // instructions per piece less than one pointer + one M0 per piece.  M0 is written by nothing else in these kernels.
    auto dma_piece = [&](int grp, uint32_t buf_off, int j) __attribute__((always_inline)) {      // buf_off: the buffer's byte offset
        const int jb = (j < DMA_PIECES - 1) ? (j & ~3) : j;
        const int kib = (j < DMA_PIECES - 1) ? wave * (DMA_PIECES - 1) + jb : SBB / 1024 - 1;
        const char* src = (const char*)sa.sblob + (size_t)grp * SBB + (size_t)kib * 1024;
        const uint32_t dst = lds_base + buf_off + (uint32_t)(kib * 1024);
        if (j == jb) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(lane16), "s"(src), "s"(dst) : "memory", "m0");
        else if (j - jb == 1) asm volatile("global_load_lds_dwordx4 %0, %1 offset:1024" : : "v"(lane16), "s"(src) : "memory");
        else if (j - jb == 2) asm volatile("global_load_lds_dwordx4 %0, %1 offset:2048" : : "v"(lane16), "s"(src) : "memory");
        else asm volatile("global_load_lds_dwordx4 %0, %1 offset:3072" : : "v"(lane16), "s"(src) : "memory");
    };
    // half h (0: hi tiles, 1: lo tiles) of the operand of forward step `st` (k-block st / 3, component st % 3; st == FSTEPS: W)
    auto fwd_ld = [&](const char* B, int st, int h, PTile& t) __attribute__((always_inline)) {
        const char* q = (st < FSTEPS) ? B + (st % 3) * PLANE + (st / 3) * 1024 + lane_tr + (h ? PNDF_LBS_SB_PL : PNDF_LBS_SB_PH)
                                      : B + lane_tr + (h ? PNDF_LBS_SB_WL : PNDF_LBS_SB_WH);
        if (h == 0) { t.h0 = lds_tr(q); t.h1 = lds_tr(q + 512); }
        else { t.l0 = lds_tr(q); t.l1 = lds_tr(q + 512); }
    };
    // forward MFMA m (0 .. FMFMA - 1): three per step, pose blend shapes first, then the 12 skinning-transform entries
    auto fwd_mfma = [&](int m, const PTile (&tl)[FLA + 1], f32x4 (&off)[3], f32x4 (&Tm)[12]) __attribute__((always_inline)) {
        const int st = m / 3, j = m % 3;
        const PTile& tc = tl[(st < FSTEPS ? st : FSTEPS) % (FLA + 1)];
        const f16x8 A = (j == 2) ? cat(tc.l0, tc.l1) : cat(tc.h0, tc.h1);
        if (st < FSTEPS) {
            const int c3 = st % 3, kb = st / 3;
            off[c3] = mf16(A, (j == 1) ? pfl[kb] : pfh[kb], (kb == 0 && j == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : off[c3]);
        } else {
            const int e = st - FSTEPS;
            Tm[e] = mf16(A, (j == 1) ? Al[e] : Ah[e], (j == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : Tm[e]);
        }
    };
    // the tile reads that ride behind forward MFMA m: behind the first two MFMAs of a step, the halves of step st + FLA
    auto fwd_reads = [&](const char* B, int m, PTile (&tl)[FLA + 1]) __attribute__((always_inline)) {
        const int st = m / 3, j = m % 3;
        if (j < 2 && st + FLA <= FSTEPS) fwd_ld(B, st + FLA, j, tl[(st + FLA) % (FLA + 1)]);
    };
    auto forward_all = [&](const char* B, f32x4 (&off)[3], f32x4 (&Tm)[12]) __attribute__((always_inline)) {      // (first group of a wave)
        PTile tl[FLA + 1];
#pragma unroll
        for (int st = 0; st < FLA; ++st) { fwd_ld(B, st, 0, tl[st]); fwd_ld(B, st, 1, tl[st]); }
        static_for<FMFMA>([&](auto mc) __attribute__((always_inline)) {
            constexpr int m = decltype(mc)::value;
            __builtin_amdgcn_sched_barrier(0);
            fwd_mfma(m, tl, off, Tm);
            __builtin_amdgcn_sched_barrier(0);
            fwd_reads(B, m, tl);
        });
        __builtin_amdgcn_sched_barrier(0);
    };
    // piece i of the VALU section: accumulators of the forward pass -> v_posed, V and (MODE 1) d L / d V of the temporal term
    // (true scale; the arithmetic of the fp32 kernel).  No epsilon under the root, as in the reference (motion_denoise.py:89):
    // two identical consecutive vertices give NaN there and here.
    auto valu_piece = [&](int i, GState& q, const char* B, const f32x4 (&off)[3], const f32x4 (&Tm)[12]) __attribute__((always_inline)) {
        if (i == 0) {
            q.fl = *(const i32x4*)(B + PNDF_LBS_SB_FL + 16 * g);
        } else if (i <= 3) {
            q.vs[i - 1] = *(const f32x4*)(B + PNDF_LBS_SB_VS + (i - 1) * (GV * 4) + 16 * g);
        } else if (i <= 6) {
            q.vp[i - 4] = q.vs[i - 4] + off[i - 4] * sa.off_true;
        } else if (i <= 21) {
            const int a3 = (i - 7) / 5, sub = (i - 7) % 5;
            if (sub == 0) q.V[a3] = Tm[3 * a3] * q.vp[0];
            else if (sub <= 2) q.V[a3] = q.V[a3] + Tm[3 * a3 + sub] * q.vp[sub];
            else if (sub == 3) q.V[a3] = q.V[a3] + Tm[9 + a3];
            else q.V[a3] = q.V[a3] * sa.tm_true;
        } else if (i <= 27) {
            const int a3 = (i - 22) / 2, r0 = 2 * ((i - 22) % 2);
#pragma unroll
            for (int r = r0; r < r0 + 2; ++r) q.d[a3][r] = q.V[a3][r] - next_frame(q.V[a3][r]);      // frame p + 1: the next lane of the row
        } else if (i <= 30) {
            const int a3 = i - 28;
            q.n2 = (a3 == 0) ? q.d[0] * q.d[0] : q.n2 + q.d[a3] * q.d[a3];
        } else if (i <= 34) {
            const int r = i - 31;
            q.inv[r] = a.w_temp * __builtin_amdgcn_rsqf(q.n2[r]);      // 0 -> inf -> 0 * inf = NaN, as d / sqrt(0)
        } else if (i <= 40) {
            const int a3 = (i - 35) / 2, r0 = 2 * ((i - 35) % 2);
#pragma unroll
            for (int r = r0; r < r0 + 2; ++r) q.u[a3][r] = (pair_ok && q.fl[r] != -2) ? q.d[a3][r] * q.inv[r] : 0.f;
        } else if (i <= 46) {
            const int a3 = (i - 41) / 2, r0 = 2 * ((i - 41) % 2);
#pragma unroll
            for (int r = r0; r < r0 + 2; ++r) q.gV[a3][r] = q.u[a3][r] - prev_frame(q.u[a3][r]);      // lane 0: no pair inside this chunk
        }
    };
#if PNDF_LBS_PAIR_READS
    // Row reads of the reverse pass: ONE lane-dependent base per group and constant offsets.  The two planes an MFMA operand is
    // made of lie a multiple of 512 bytes apart (PLANE = 14 x 512, PL - PH = 42 x 512), so hipcc merges the pair into one
    // ds_read2st64_b64 whose four result registers ARE the operand: 39 instead of 78 LDS instructions per group.
    struct RBase { lds_char *q, *wm; };
    auto rev_base = [&](const char* B) __attribute__((always_inline)) {
        // wm: second joint tile (joints 16 .. 23 + eight rows of padding): tile rows 0 .. 7 read the hi plane's rows, tile rows
        // 8 .. 15 the LO plane's rows of the same eight joints
        return RBase{lds_opaque((lds_char*)B + lane_rv),
                     lds_opaque((lds_char*)B + (p < 8 ? PNDF_LBS_SB_WH : PNDF_LBS_SB_WL) + 512 + 8 * pndf_lbs_sb_unit(p & 7, g))};
    };
    // pair `w` (0: components 0, 1 hi   1: components 0, 1 lo   2: component 2 hi, lo) of row tile kt
    auto rev_ld = [&](const RBase& rb, int kt, int w, RTile& t) __attribute__((always_inline)) {
        lds_char* q = rb.q + kt * 512;
        if (w == 0) { t.c0h = lds_row(q + PNDF_LBS_SB_PH); t.c1h = lds_row(q + PNDF_LBS_SB_PH + PLANE); }
        else if (w == 1) { t.c0l = lds_row(q + PNDF_LBS_SB_PL); t.c1l = lds_row(q + PNDF_LBS_SB_PL + PLANE); }
        else { t.c2h = lds_row(q + PNDF_LBS_SB_PH + 2 * PLANE); t.c2l = lds_row(q + PNDF_LBS_SB_PL + 2 * PLANE); }
    };
#else
    // Row reads of the reverse pass: one ds_read_b64 each, through one opaque base per plane (off ONE base hipcc merges the two
    // planes of an MFMA operand -- a multiple of 512 bytes apart -- into a ds_read2st64_b64: 39 instead of 78 LDS instructions
    // per group, and 2 % SLOWER, same box: profiles/r04/lbs/ablation_r04.txt)
    struct RBase { lds_char *c0h, *c1h, *c2h, *c0l, *c1l, *c2l, *q, *wm; };
    auto rev_base = [&](const char* B) __attribute__((always_inline)) {
        lds_char* q = (lds_char*)B + lane_rv;
        return RBase{lds_opaque(q + PNDF_LBS_SB_PH), lds_opaque(q + PNDF_LBS_SB_PH + PLANE), lds_opaque(q + PNDF_LBS_SB_PH + 2 * PLANE),
                     lds_opaque(q + PNDF_LBS_SB_PL), lds_opaque(q + PNDF_LBS_SB_PL + PLANE), lds_opaque(q + PNDF_LBS_SB_PL + 2 * PLANE),
                     lds_opaque(q),
                     // second joint tile (joints 16 .. 23 + eight rows of padding): tile rows 0 .. 7 read the hi plane's rows,
                     // tile rows 8 .. 15 the LO plane's rows of the same eight joints
                     lds_opaque((lds_char*)B + (p < 8 ? PNDF_LBS_SB_WH : PNDF_LBS_SB_WL) + 512 + 8 * pndf_lbs_sb_unit(p & 7, g))};
    };
    // row read `w` (0 .. 5) of row tile kt
    auto rev_ld = [&](const RBase& rb, int kt, int w, RTile& t) __attribute__((always_inline)) {
        if (w == 0) t.c0h = lds_row(rb.c0h + kt * 512);
        else if (w == 1) t.c1h = lds_row(rb.c1h + kt * 512);
        else if (w == 2) t.c0l = lds_row(rb.c0l + kt * 512);
        else if (w == 3) t.c1l = lds_row(rb.c1l + kt * 512);
        else if (w == 4) t.c2h = lds_row(rb.c2h + kt * 512);
        else t.c2l = lds_row(rb.c2l + kt * 512);
    };
#endif

    // One group.  HAS_NEXT: the forward pass of group grp + 1 runs through the VALU section of group grp -- the pose-blend
    // MFMAs accumulate into off_n while `off` is still read, the skinning MFMAs come last and write Tm IN PLACE: by then every
    // reader of this group's Tm (V, T_R^T g) is done.  The last group of a wave is its own instantiation (no branch around
    // the forward pass in the loop body); FETCH: blob grp + 2 exists and is fetched here.
    //   part 1a  forward MFMAs 0 .. 46  <- VALU pieces 0 .. 46 (v_posed, V, temporal term)
    //   (rare branch: data term on the vertex-picked joints)
    //   part 1b  forward MFMAs 47 .. 98 <- the 15 pieces of g_scale x T_R^T gV behind the first of them; skinning MFMAs bare
    //   part 2   per entry e of d L / d A: four MFMAs <- the operand split of entry e + 1; one row tile of d L / d pose_feature
    //            (five MFMAs) <- its row reads RLA tiles ahead; the thirteenth tile at the end
    constexpr int NG1 = (MODE == 0) ? 0 : 15, M1A = (MODE == 0) ? FMFMA : NP1;
    static_assert(NP1 + NG1 <= 3 * FSTEPS, "all readers of Tm sit behind pose-blend MFMAs, in front of the skinning MFMAs");
    // byte offsets of the three model buffers by role -- this group's blob, the next one's, the fetch target -- rotated at
    // the end of every group (a group index modulo 3 costs a dozen scalar instructions per group)
    uint32_t rot0 = 0u, rot1 = (uint32_t)SBB, rot2 = 2u * (uint32_t)SBB;
    // (PNDF_LBS_DIAG & 256: s_memtime stamps at the region boundaries of a group, printed by one wave at the end)
    unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0}, tlast = 0;
    auto stamp = [&](int r) __attribute__((always_inline)) {
        if (PNDF_LBS_DIAG & 256) {
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            tacc[r] += now - tlast;
            tlast = now;
        }
    };
    auto group = [&](auto has_next, auto has_fetch, int grp, f32x4 (&off)[3], f32x4 (&Tm)[12], f32x4 (&off_n)[3]) __attribute__((always_inline)) {
        constexpr bool HAS_NEXT = decltype(has_next)::value;
        constexpr bool FETCH = decltype(has_fetch)::value && !(PNDF_LBS_DIAG & 2);
        const char* B = smem_s + rot0;
        const char* Bn = smem_s + rot1;
        GState q;
        PTile tl[FLA + 1];
        f32x4 gt[3];
        f16x4 gh[3], gl[3];
        const float cgu = sa.tm_true * sa.g_scale;      // T_R = tm_true Tm; operand scale of d L / d v_posed
        // g_scale x T_R^T gV, component b = i / 5: three products, two splits
        auto g_piece = [&](int i) __attribute__((always_inline)) {
            const int b = i / 5, pc = i % 5;
            unsigned h, l;
            if (pc == 0) gt[b] = Tm[b] * q.gV[0];
            else if (pc == 1) gt[b] = gt[b] + Tm[3 + b] * q.gV[1];
            else if (pc == 2) gt[b] = (gt[b] + Tm[6 + b] * q.gV[2]) * cgu;
            else {
                const int r0 = 2 * (pc - 3);
                lbs_split2(gt[b][r0], gt[b][r0 + 1], h, l);
                gh[b][r0] = __builtin_bit_cast(f16x2, h)[0]; gh[b][r0 + 1] = __builtin_bit_cast(f16x2, h)[1];
                gl[b][r0] = __builtin_bit_cast(f16x2, l)[0]; gl[b][r0 + 1] = __builtin_bit_cast(f16x2, l)[1];
            }
        };
        // forward MFMA m with what rides behind it
        auto fwd_slot = [&](auto mc) __attribute__((always_inline)) {
            constexpr int m = decltype(mc)::value;
            __builtin_amdgcn_sched_barrier(0);
            fwd_mfma(m, tl, off_n, Tm);
            __builtin_amdgcn_sched_barrier(0);
            if (!(PNDF_LBS_DIAG & 16)) fwd_reads(Bn, m, tl);
            if constexpr (FETCH && m % 4 == 1 && m / 4 < DMA_PIECES) dma_piece(grp + 2, rot2, m / 4);
            if constexpr (m < NP1) valu_piece(m, q, B, off, Tm);
            else if constexpr (m < NP1 + NG1) g_piece(m - NP1);
        };
        stamp(5);      // (loop overhead since the last group's end)
        if constexpr (HAS_NEXT) {
            if (!(PNDF_LBS_DIAG & 1)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of blob grp + 1 have landed ...
            if (!(PNDF_LBS_DIAG & 4)) __syncthreads();            // ... everyone's have, and everyone has left buffer (k + 2) % 3
            stamp(0);
#pragma unroll
            for (int st = 0; st < FLA; ++st) { fwd_ld(Bn, st, 0, tl[st]); fwd_ld(Bn, st, 1, tl[st]); }
            static_for<M1A>([&](auto mc) __attribute__((always_inline)) { fwd_slot(mc); });
            __builtin_amdgcn_sched_barrier(0);
            stamp(1);
        } else {
            static_for<NP1>([&](auto ic) __attribute__((always_inline)) { valu_piece(decltype(ic)::value, q, B, off, Tm); });
        }

        if constexpr (MODE == 0) {
            if (t_ok) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int v = grp * GV + 4 * g + r;
                    if (v < a.V) {
                        if (a.verts) {
                            float* dst = a.verts + ((size_t)n * a.V + v) * 3;
                            dst[0] = q.V[0][r]; dst[1] = q.V[1][r]; dst[2] = q.V[2][r];
                        }
                        if (q.fl[r] >= 0 && a.joints) {
                            float* dst = a.joints + ((size_t)n * njt + NJ + q.fl[r]) * 3;
                            dst[0] = q.V[0][r]; dst[1] = q.V[1][r]; dst[2] = q.V[2][r];
                        }
                    }
                }
            }
        } else {
            // data term on the vertex-picked joints (the 24 chain joints: pndf_lbs_pose_backward_kernel): 21 of SMPL's 6,890
            // vertices -- one rarely taken branch per group
            if (a.it_gt0 && owned && (q.fl[0] >= 0 || q.fl[1] >= 0 || q.fl[2] >= 0 || q.fl[3] >= 0)) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (q.fl[r] >= 0) {
                        const float* j0 = a.joints0 + ((size_t)n * njt + NJ + q.fl[r]) * 3;
                        const float dx = q.V[0][r] - j0[0], dy = q.V[1][r] - j0[1], dz = q.V[2][r] - j0[2];
                        const float inv = a.w_data / sqrtf(dx * dx + dy * dy + dz * dz);
                        q.gV[0][r] += dx * inv; q.gV[1][r] += dy * inv; q.gV[2][r] += dz * inv;
                    }
                }
            }
            if constexpr (HAS_NEXT) {
                static_for<FMFMA - M1A>([&](auto mc) __attribute__((always_inline)) { fwd_slot(std::integral_constant<int, M1A + decltype(mc)::value>{}); });
                __builtin_amdgcn_sched_barrier(0);
                stamp(2);
            } else {
                static_for<NG1>([&](auto ic) __attribute__((always_inline)) { g_piece(decltype(ic)::value); });
            }

            // ---- reverse pass
            const f16x4 zero4 = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
            const RBase rb = rev_base(B);
            RTile rt[RLA + 1];
            // W^T: rows = joints.  Joints 0 .. 15: hi with the tile's 16 vertices twice (hi hi + hi lo in one k-block) and lo
            // once.  Joints 16 .. 23 fill half a tile: its other eight rows carry the LO halves of the same joints (twice as
            // well: lo hi + lo lo), so ONE MFMA does all of it -- rows 8 .. 15 of the accumulator are added to rows 0 .. 7
            // when the kernel stores its results (36 instead of 48 MFMAs per group for d L / d A).
            f16x8 AW[3];
            {
                const f16x4 wh = lds_row(rb.q + PNDF_LBS_SB_WH), wl = lds_row(rb.q + PNDF_LBS_SB_WL), wm = lds_row(rb.wm);
                AW[0] = cat(wh, wh);
                AW[1] = cat(wl, zero4);
                AW[2] = cat(wm, wm);
            }
#pragma unroll
            for (int kt = 0; kt < RLA; ++kt)
#pragma unroll
                for (int w = 0; w < (PNDF_LBS_PAIR_READS ? 3 : 6); ++w) rev_ld(rb, kt, w, rt[kt]);
            f32x4 X;
            f16x4 xh, xl;
            f16x8 Xe[2];
            f32x4 (&gVs)[3] = q.gV;      // from here on x_scale x d L / d verts
            // d L / d verts (x) [v_posed, 1], entry e, split: three pieces
            auto x_piece = [&](int e, int pc) __attribute__((always_inline)) {
                unsigned h, l;
                if (pc == 0) {
                    X = (e < 9) ? gVs[e / 3] * q.vp[e % 3] : gVs[e - 9];
                } else {
                    const int r0 = 2 * (pc - 1);
                    lbs_split2(X[r0], X[r0 + 1], h, l);
                    xh[r0] = __builtin_bit_cast(f16x2, h)[0]; xh[r0 + 1] = __builtin_bit_cast(f16x2, h)[1];
                    xl[r0] = __builtin_bit_cast(f16x2, l)[0]; xl[r0 + 1] = __builtin_bit_cast(f16x2, l)[1];
                    if (pc == 2) Xe[e & 1] = cat(xh, xl);
                }
            };
            const f16x8 B0h = cat(gh[0], gh[1]), B0l = cat(gl[0], gl[1]);
            // component 2: the row tile's [hi | lo] registers as they were read serve both MFMAs (no copies in front of them)
            const f16x8 B1 = cat(gh[2], gh[2]), B2 = cat(gl[2], zero4);
            // MFMA t (0 .. 4) of row tile kt of d L / d pose_feature; behind it, row reads of tile kt + RLA
            auto gpf_mfma = [&](int kt, int t) __attribute__((always_inline)) {
                const RTile& tc = rt[kt % (RLA + 1)];
                __builtin_amdgcn_sched_barrier(0);
                if (PNDF_LBS_DIAG & 64) asm volatile("" : : "v"(tc.c0h), "v"(tc.c1h), "v"(tc.c0l), "v"(tc.c1l), "v"(tc.c2h), "v"(tc.c2l));
                else if (t == 0) gpf[kt] = mf16(cat(tc.c0h, tc.c1h), B0h, gpf[kt]);
                else if (t == 1) gpf[kt] = mf16(cat(tc.c0h, tc.c1h), B0l, gpf[kt]);
                else if (t == 2) gpf[kt] = mf16(cat(tc.c0l, tc.c1l), B0h, gpf[kt]);
                else if (t == 3) gpf[kt] = mf16(cat(tc.c2h, tc.c2l), B1, gpf[kt]);       // hi hi + lo hi of component 2 in one k-block
                else gpf[kt] = mf16(cat(tc.c2h, tc.c2l), B2, gpf[kt]);                   // hi lo (the lo half meets zeros)
                __builtin_amdgcn_sched_barrier(0);
                // (the registers of tile kt + RLA are those of tile kt - 1: free since its last MFMA)
                if (kt + RLA < KT && !(PNDF_LBS_DIAG & 32)) {
#if PNDF_LBS_PAIR_READS
                    if (t < 3) rev_ld(rb, kt + RLA, t, rt[(kt + RLA) % (RLA + 1)]);
#else
                    rev_ld(rb, kt + RLA, t, rt[(kt + RLA) % (RLA + 1)]);
                    if (t == 4) rev_ld(rb, kt + RLA, 5, rt[(kt + RLA) % (RLA + 1)]);
#endif
                }
            };
#pragma unroll
            for (int a3 = 0; a3 < 3; ++a3) gVs[a3] = gVs[a3] * sa.x_scale;
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) x_piece(0, pc);
            stamp(3);
#pragma unroll
            for (int i = 0; i < 12; ++i) {
#pragma unroll
                for (int jm = 0; jm < 3; ++jm) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (PNDF_LBS_DIAG & 64) asm volatile("" : : "v"(AW[jm]), "v"(Xe[i & 1]));
                    else gA[i][jm / 2] = mf16(AW[jm], Xe[i & 1], gA[i][jm / 2]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (i + 1 < 12 && !(PNDF_LBS_DIAG & 8)) x_piece(i + 1, jm);
                }
#pragma unroll
                for (int t = 0; t < 5; ++t) gpf_mfma(i, t);
            }
#pragma unroll
            for (int t = 0; t < 5; ++t) gpf_mfma(KT - 1, t);
            __builtin_amdgcn_sched_barrier(0);
            stamp(4);
            static_assert(KT == 13, "row tiles of d L / d pose_feature: one per entry of d L / d A, one left over");
        }
        if constexpr (HAS_NEXT) {
#pragma unroll
            for (int c3 = 0; c3 < 3; ++c3) off[c3] = off_n[c3];
        }
        const uint32_t r = rot0;
        rot0 = rot1; rot1 = rot2; rot2 = r;
    };
    f32x4 off[3], Tm[12], off_n[3];      // accumulators: p_scale 2^12 x pose-blend offset, w_scale a_scale x sum_j W[v, j] A_j
    if (grp0 < grp1) {
#pragma unroll
        for (int j = 0; j < DMA_PIECES; ++j) dma_piece(grp0, 0u, j);
        if (grp0 + 1 < grp1) {
#pragma unroll
            for (int j = 0; j < DMA_PIECES; ++j) dma_piece(grp0 + 1, (uint32_t)SBB, j);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        forward_all(smem_s, off, Tm);
    }
    if (PNDF_LBS_DIAG & 256) tlast = __builtin_amdgcn_s_memtime();
    for (int grp = grp0; grp + 2 < grp1; ++grp) group(std::true_type{}, std::true_type{}, grp, off, Tm, off_n);
    if (grp0 + 1 < grp1) group(std::true_type{}, std::false_type{}, grp1 - 2, off, Tm, off_n);
    if (grp0 < grp1) group(std::false_type{}, std::false_type{}, grp1 - 1, off, Tm, off_n);
    if ((PNDF_LBS_DIAG & 256) && MODE == 1 && blockIdx.x == 37 && threadIdx.x == 64)
        printf("lbs regions (cycles of %d groups, wave 1 of workgroup 37): wait+barrier %llu  fwd 1a %llu  fwd 1b %llu  rev prologue %llu  rev %llu  between groups %llu\n",
               grp1 - grp0, tacc[0], tacc[1], tacc[2], tacc[3], tacc[4], tacc[5]);
    if constexpr (MODE != 0) {
        float* o_A = nullptr;
        if (t_ok) {
            float* o_pf = owned ? a.gpf + ((size_t)vs * N + n) * PF : a.halo_pf + ((size_t)vs * nch + cid) * PF;
            o_A = owned ? a.gA + ((size_t)vs * N + n) * A_FLOATS : a.halo_A + ((size_t)vs * nch + cid) * A_FLOATS;
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) *(f32x4*)(o_pf + 16 * kt + 4 * g) = gpf[kt] * sa.gpf_true;
#pragma unroll
            for (int e = 0; e < 12; ++e) *(f32x4*)(o_A + e * 32 + 4 * g) = gA[e][0] * sa.gA_true;
        }
        // second joint tile: accumulator rows 8 .. 15 (lanes 32 .. 63) hold the lo halves' share of joints 16 .. 23 (every lane
        // takes part in the exchange; rows 24 .. 31 of the output are the padding joints: zero, as W's padding rows made them)
#pragma unroll
        for (int e = 0; e < 12; ++e) {
            f32x4 hi_part = gA[e][1], lo_part;
#pragma unroll
            for (int r = 0; r < 4; ++r) lo_part[r] = __shfl_down(hi_part[r], 32, 64);
            if (t_ok) *(f32x4*)(o_A + e * 32 + 16 + 4 * g) = (g < 2) ? (hi_part + lo_part) * sa.gA_true : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
}

extern "C" __global__ void __launch_bounds__(256, 1) pndf_lbs_vertex_split_forward_kernel(PndfLbsSplitArgs a) { lbs_vertex_split_body<0>(a); }
extern "C" __global__ void __launch_bounds__(256, 1) pndf_lbs_vertex_split_terms_kernel(PndfLbsSplitArgs a) { lbs_vertex_split_body<1>(a); }

// ------------------------------------------------------------------------------------------ host: handle, packing, launches
struct pndf_lbs_model {
    int device = 0;
    int V = 0, NG = 0, NE = 0;
    float* d_blob = nullptr;
    void* d_sblob = nullptr;          // the model for the split-precision kernels (pndf_lbs_split.h)
    float* d_picked = nullptr;        // [NE][PICK_FLOATS]: the vertex-picked joints' own rows of the model (joints-only forward)
    int precision = PNDF_LBS_F16X3;   // forward and fused-terms passes; the general reverse pass is always fp32
    float p_scale = 1.f, w_scale = 1.f, a_scale = 1.f;      // powers of two: model and joint-transform operands
    float w_rowsum = 1.f;             // max_v sum_j |W[v, j]|                        (bounds |T_R^T g|)
    float vp_bound = 1.f;             // max_v,c |v_shaped| + 2 sum_k |posedirs|      (bounds |v_posed|)
    bool smpl_tree = false;           // the kinematic tree is SMPL's own: per-frame kernels with the tree as a compile-time table
    PndfLbsModel consts;
    int sm_count = 256;
    std::string err;
};

// fp32 -> fp16 bits, round to nearest even (host side of the split: any hi + lo = x decomposition serves the kernels)
static uint16_t f32_to_f16(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint16_t sign = (uint16_t)((x >> 16) & 0x8000u);
    x &= 0x7fffffffu;
    if (x > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);
    if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);                          // >= 65520: inf
    if (x < 0x38800000u) {                                                            // below 2^-14: subnormal, units of 2^-24
        float af;
        memcpy(&af, &x, 4);
        return (uint16_t)(sign | (uint16_t)std::nearbyint((double)af * 16777216.0));
    }
    uint32_t h = (((x >> 23) - 112u) << 10) | ((x & 0x7fffffu) >> 13);
    const uint32_t rem = x & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) ++h;                           // (a carry runs into the exponent)
    return (uint16_t)(sign | h);
}
static float f16_to_f32(uint16_t h) {
    const int e = (h >> 10) & 31, m = h & 1023;
    const float v = (e == 0) ? std::ldexp((float)m, -24) : std::ldexp((float)(m | 1024), e - 25);
    return (h & 0x8000u) ? -v : v;
}
// power of two s with bound * s in [2^13, 2^14): the operand's hi half cannot overflow (65504), and values down to 2^-16 of
// the bound keep a normal lo half.  Clamped: a zero bound must not turn into an infinite factor.
static float lbs_scale_for(float bound) {
    if (!(bound > 1e-30f)) bound = 1e-30f;
    if (bound > 1e30f) bound = 1e30f;
    return std::ldexp(1.0f, 13 - std::ilogb(bound));
}

// the split-precision model from the fp32 one (same [row][16 v] order inside a plane)
static void lbs_pack_split(const float* blob, int NG, float p_scale, float w_scale, uint8_t* out) {
    memset(out, 0, (size_t)NG * PNDF_LBS_SB_BYTES);
    for (int grp = 0; grp < NG; ++grp) {
        const float* b = blob + (size_t)grp * BLOB;
        uint8_t* o = out + (size_t)grp * PNDF_LBS_SB_BYTES;
        auto put = [&](int hi_off, int lo_off, int row, int vi, float x) {
            const uint16_t h = f32_to_f16(x);
            const uint16_t l = f32_to_f16(x - f16_to_f32(h));
            memcpy(o + hi_off + pndf_lbs_sb_at(row, vi), &h, 2);
            memcpy(o + lo_off + pndf_lbs_sb_at(row, vi), &l, 2);
        };
        for (int c = 0; c < 3; ++c)
            for (int k = 0; k < PF; ++k)
                for (int vi = 0; vi < GV; ++vi)
                    put(PNDF_LBS_SB_PH + c * PNDF_LBS_SB_PLANE, PNDF_LBS_SB_PL + c * PNDF_LBS_SB_PLANE, k, vi,
                        b[PNDF_LBS_BLOB_P + c * C_STRIDE + k * GV + vi] * p_scale);
        for (int j = 0; j < 32; ++j)
            for (int vi = 0; vi < GV; ++vi)
                put(PNDF_LBS_SB_WH, PNDF_LBS_SB_WL, j, vi, b[PNDF_LBS_BLOB_W + j * GV + vi] * w_scale);
        memcpy(o + PNDF_LBS_SB_VS, b + PNDF_LBS_BLOB_VS, 3 * GV * 4);
        memcpy(o + PNDF_LBS_SB_FL, b + PNDF_LBS_BLOB_FL, GV * 4);
    }
}

static thread_local std::string g_lbs_create_err;
static int lbs_fail(pndf_lbs_model* h, int code, const std::string& msg) {
    if (h) h->err = msg; else g_lbs_create_err = msg;
    return code;
}

extern "C" const char* pndf_lbs_last_error(pndf_lbs_handle h) { return h ? h->err.c_str() : g_lbs_create_err.c_str(); }

extern "C" int64_t pndf_lbs_packed_floats(int32_t V) { return V < 1 ? 0 : (int64_t)((V + GV - 1) / GV) * BLOB; }

// Host-only packer (needs no device): the model in MFMA tile order + the rest joints.  `J_out` (72 floats) and `rel_out`
// (72 floats) may be null.  Returns 0 or a negative pndf_status.
extern "C" int pndf_lbs_pack_host(int32_t V, int32_t NB, const float* v_template, const float* shapedirs, const float* betas,
                                  const float* posedirs, const float* J_regressor, const int32_t* parents,
                                  const float* lbs_weights, const int32_t* extra_joint_vertex, int32_t n_extra, float* blob,
                                  float* J_out, float* rel_out) {
    if (V < 1 || NB < 0 || !v_template || !posedirs || !J_regressor || !parents || !lbs_weights || !blob) return PNDF_ERR_BAD_ARG;
    if (NB > 0 && (!shapedirs || !betas)) return PNDF_ERR_BAD_ARG;
    if (n_extra < 0 || n_extra > PNDF_LBS_MAX_EXTRA || (n_extra > 0 && !extra_joint_vertex)) return PNDF_ERR_BAD_ARG;
    if (parents[0] >= 0) return PNDF_ERR_UNSUPPORTED;
    for (int j = 1; j < NJ; ++j)
        if (parents[j] < 0 || parents[j] >= j) return PNDF_ERR_UNSUPPORTED;      // parents precede their children (SMPL)
    for (int e = 0; e < n_extra; ++e)
        if (extra_joint_vertex[e] < 0 || extra_joint_vertex[e] >= V) return PNDF_ERR_BAD_ARG;
    // v_shaped = v_template + blend_shapes(betas, shapedirs); J = J_regressor v_shaped   (smplx lbs(), first two steps)
    std::vector<float> vsh((size_t)V * 3);
    for (int i = 0; i < V * 3; ++i) {
        double acc = v_template[i];
        for (int l = 0; l < NB; ++l) acc += (double)shapedirs[(size_t)i * NB + l] * betas[l];
        vsh[i] = (float)acc;
    }
    float J[NJ][3];
    for (int j = 0; j < NJ; ++j)
        for (int e = 0; e < 3; ++e) {
            double acc = 0.0;
            for (int v = 0; v < V; ++v) acc += (double)J_regressor[(size_t)j * V + v] * vsh[(size_t)v * 3 + e];
            J[j][e] = (float)acc;
        }
    for (int j = 0; j < NJ; ++j)
        for (int e = 0; e < 3; ++e) {
            if (J_out) J_out[3 * j + e] = J[j][e];
            if (rel_out) rel_out[3 * j + e] = (j == 0) ? J[j][e] : J[j][e] - J[parents[j]][e];
        }
    const int NG = (V + GV - 1) / GV;
    std::vector<int> flag((size_t)NG * GV, -2);
    for (int v = 0; v < V; ++v) flag[v] = -1;
    // The vertex kernels find a vertex-picked joint through this per-vertex flag, i.e. one joint per vertex: a vertex named
    // twice would leave the earlier joint's row unwritten (smplx's VertexJointSelector is an index_select and would serve
    // both).  SMPL's own table has no duplicates; anything else is refused rather than half served.
    for (int e = 0; e < n_extra; ++e) {
        if (flag[extra_joint_vertex[e]] >= 0) return PNDF_ERR_BAD_ARG;
        flag[extra_joint_vertex[e]] = e;
    }
    memset(blob, 0, (size_t)NG * BLOB * sizeof(float));
    const int npf = 9 * (NJ - 1);
    for (int grp = 0; grp < NG; ++grp) {
        float* b = blob + (size_t)grp * BLOB;
        for (int vi = 0; vi < GV; ++vi) {
            const int v = grp * GV + vi;
            ((int*)(b + PNDF_LBS_BLOB_FL))[vi] = flag[(size_t)grp * GV + vi];
            if (v >= V) continue;
            for (int c = 0; c < 3; ++c) {
                for (int k = 0; k < npf; ++k)
                    b[PNDF_LBS_BLOB_P + c * C_STRIDE + k * GV + vi] = posedirs[(size_t)k * V * 3 + (size_t)v * 3 + c];
                b[PNDF_LBS_BLOB_VS + c * GV + vi] = vsh[(size_t)v * 3 + c];
            }
            for (int j = 0; j < NJ; ++j) b[PNDF_LBS_BLOB_W + j * GV + vi] = lbs_weights[(size_t)v * NJ + j];
        }
    }
    return PNDF_OK;
}

// operand scales of the split-precision model (powers of two, from the largest |entry|: [2^12, 2^13)) and the bounds the
// host derives the reverse operand scales from
struct LbsSplitScales { float p_scale, w_scale, w_rowsum, vp_bound; };
static LbsSplitScales lbs_split_scales(const float* blob, int NG) {
    float maxP = 0.f, maxW = 0.f, rowsum = 0.f, vpb = 0.f;
    for (int grp = 0; grp < NG; ++grp) {
        const float* b = blob + (size_t)grp * BLOB;
        for (int vi = 0; vi < GV; ++vi) {
            float ws = 0.f;
            for (int j = 0; j < NJ; ++j) {
                const float w = std::fabs(b[PNDF_LBS_BLOB_W + j * GV + vi]);
                ws += w;
                maxW = std::fmax(maxW, w);
            }
            rowsum = std::fmax(rowsum, ws);
            for (int c = 0; c < 3; ++c) {
                float ps = 0.f;
                for (int k = 0; k < PF; ++k) {
                    const float pv = std::fabs(b[PNDF_LBS_BLOB_P + c * C_STRIDE + k * GV + vi]);
                    ps += pv;
                    maxP = std::fmax(maxP, pv);
                }
                vpb = std::fmax(vpb, std::fabs(b[PNDF_LBS_BLOB_VS + c * GV + vi]) + 2.0f * ps);      // |R - I| <= 2
            }
        }
    }
    return LbsSplitScales{lbs_scale_for(maxP) * 0.5f, lbs_scale_for(maxW) * 0.5f, std::fmax(rowsum, 1e-6f), std::fmax(vpb, 1.0f)};
}

extern "C" int64_t pndf_lbs_packed_split_bytes(int32_t V) { return V < 1 ? 0 : (int64_t)((V + GV - 1) / GV) * PNDF_LBS_SB_BYTES; }

// Host-only: the split-precision model (what pndf_lbs_create uploads for PNDF_LBS_F16X3) from the fp32 one of
// pndf_lbs_pack_host.  scales_out (may be null): p_scale, w_scale.
extern "C" int pndf_lbs_pack_split_host(int32_t V, const float* blob, void* sblob, float* scales_out) {
    if (V < 1 || !blob || !sblob) return PNDF_ERR_BAD_ARG;
    const int NG = (V + GV - 1) / GV;
    const LbsSplitScales sc = lbs_split_scales(blob, NG);
    lbs_pack_split(blob, NG, sc.p_scale, sc.w_scale, (uint8_t*)sblob);
    if (scales_out) { scales_out[0] = sc.p_scale; scales_out[1] = sc.w_scale; }
    return PNDF_OK;
}

extern "C" int pndf_lbs_create(pndf_lbs_handle* out, int32_t V, int32_t NB, const float* v_template, const float* shapedirs,
                               const float* betas, const float* posedirs, const float* J_regressor, const int32_t* parents,
                               const float* lbs_weights, const int32_t* extra_joint_vertex, int32_t n_extra, int device) {
    if (!out) return lbs_fail(nullptr, PNDF_ERR_BAD_ARG, "out is null");
    *out = nullptr;
    if (V < 1) return lbs_fail(nullptr, PNDF_ERR_BAD_ARG, "V < 1");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev)
        return lbs_fail(nullptr, PNDF_ERR_NO_DEVICE, "no HIP device " + std::to_string(device) + " (the body model has no CPU fallback)");
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess || std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
        return lbs_fail(nullptr, PNDF_ERR_NO_DEVICE, "kernels are built for gfx950 only");
    const int NG = (V + GV - 1) / GV;
    std::vector<float> blob((size_t)NG * BLOB);
    float J[NJ * 3], rel[NJ * 3];
    const int rc = pndf_lbs_pack_host(V, NB, v_template, shapedirs, betas, posedirs, J_regressor, parents, lbs_weights,
                                      extra_joint_vertex, n_extra, blob.data(), J, rel);
    if (rc == PNDF_ERR_UNSUPPORTED)
        return lbs_fail(nullptr, rc, "kinematic tree: 24 joints, parents[0] = -1 and every parent before its children (SMPL)");
    if (rc != PNDF_OK) return lbs_fail(nullptr, rc, "null pointer, or an extra-joint vertex outside [0, V) or named twice, or more than 32 of them");
    DeviceGuard guard(device);
    if (!guard.ok) return lbs_fail(nullptr, PNDF_ERR_HIP, "hipSetDevice failed");
    pndf_lbs_model* h = new pndf_lbs_model();
    h->device = device; h->V = V; h->NG = NG; h->NE = n_extra;
    h->sm_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    memcpy(h->consts.J, J, sizeof(J));
    memcpy(h->consts.rel, rel, sizeof(rel));
    if (n_extra > 0) {
        std::vector<float> pk((size_t)n_extra * PICK_FLOATS);
        const int npf = 9 * (NJ - 1);
        for (int x = 0; x < n_extra; ++x) {
            const int v = extra_joint_vertex[x];
            float* P = pk.data() + (size_t)x * PICK_FLOATS;
            for (int c = 0; c < 3; ++c) {
                double acc = v_template[(size_t)v * 3 + c];
                for (int l = 0; l < NB; ++l) acc += (double)shapedirs[((size_t)v * 3 + c) * NB + l] * betas[l];
                P[c] = (float)acc;
            }
            for (int j = 0; j < NJ; ++j) P[3 + j] = lbs_weights[(size_t)v * NJ + j];
            for (int k = 0; k < npf; ++k)
                for (int c = 0; c < 3; ++c) P[3 + NJ + 3 * k + c] = posedirs[(size_t)k * V * 3 + (size_t)v * 3 + c];
        }
        if (hipMalloc((void**)&h->d_picked, pk.size() * sizeof(float)) != hipSuccess ||
            hipMemcpy(h->d_picked, pk.data(), pk.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
            if (h->d_picked) (void)hipFree(h->d_picked);
            delete h;
            return lbs_fail(nullptr, PNDF_ERR_HIP, "pndf_lbs_create: picked-joint table");
        }
    }
    h->smpl_tree = true;
    for (int j = 0; j < NJ; ++j) {
        h->consts.parent[j] = parents[j];
        h->smpl_tree = h->smpl_tree && (j == 0 || parents[j] == SmplTree::tab[j]);
    }
    hipError_t e = hipMalloc((void**)&h->d_blob, blob.size() * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(h->d_blob, blob.data(), blob.size() * sizeof(float), hipMemcpyHostToDevice);
    {
        // operand scales and bounds of the split-precision kernels, from the packed model
        const LbsSplitScales sc = lbs_split_scales(blob.data(), NG);
        float extent = 0.f, maxJ = 0.f;      // |G_t[j]| <= sum of the bone lengths, |A_t| <= |G_t| + |J|
        for (int j = 0; j < NJ; ++j) {
            extent += std::sqrt(rel[3 * j] * rel[3 * j] + rel[3 * j + 1] * rel[3 * j + 1] + rel[3 * j + 2] * rel[3 * j + 2]);
            maxJ = std::fmax(maxJ, std::sqrt(J[3 * j] * J[3 * j] + J[3 * j + 1] * J[3 * j + 1] + J[3 * j + 2] * J[3 * j + 2]));
        }
        h->p_scale = sc.p_scale;
        h->w_scale = sc.w_scale;
        h->a_scale = lbs_scale_for(std::fmax(1.0f, extent + maxJ)) * 0.5f;
        h->w_rowsum = sc.w_rowsum;
        h->vp_bound = sc.vp_bound;
        std::vector<uint8_t> sblob((size_t)NG * PNDF_LBS_SB_BYTES);
        lbs_pack_split(blob.data(), NG, h->p_scale, h->w_scale, sblob.data());
        if (e == hipSuccess) e = hipMalloc(&h->d_sblob, sblob.size());
        if (e == hipSuccess) e = hipMemcpy(h->d_sblob, sblob.data(), sblob.size(), hipMemcpyHostToDevice);
        const int lds_s = 3 * PNDF_LBS_SB_BYTES;
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)pndf_lbs_vertex_split_forward_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_s);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)pndf_lbs_vertex_split_terms_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_s);
    }
    const int lds = 3 * BLOB * (int)sizeof(float);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)pndf_lbs_vertex_forward_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)pndf_lbs_vertex_terms_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)pndf_lbs_vertex_reverse_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) {
        const std::string m = std::string("pndf_lbs_create: ") + hipGetErrorString(e);
        if (h->d_blob) (void)hipFree(h->d_blob);
        if (h->d_sblob) (void)hipFree(h->d_sblob);
        if (h->d_picked) (void)hipFree(h->d_picked);
        delete h;
        return lbs_fail(nullptr, PNDF_ERR_HIP, m);
    }
    *out = h;
    return PNDF_OK;
}

extern "C" int pndf_lbs_destroy(pndf_lbs_handle h) {
    if (!h) return PNDF_OK;
    DeviceGuard guard(h->device);
    if (h->d_blob) (void)hipFree(h->d_blob);
    if (h->d_sblob) (void)hipFree(h->d_sblob);
    if (h->d_picked) (void)hipFree(h->d_picked);
    delete h;
    return PNDF_OK;
}

extern "C" int pndf_lbs_set_precision(pndf_lbs_handle h, int32_t precision) {
    if (!h) return PNDF_ERR_BAD_ARG;
    if (precision != PNDF_LBS_FP32 && precision != PNDF_LBS_F16X3) return lbs_fail(h, PNDF_ERR_BAD_ARG, "precision: PNDF_LBS_FP32 or PNDF_LBS_F16X3");
    h->precision = precision;
    return PNDF_OK;
}
extern "C" int32_t pndf_lbs_precision(pndf_lbs_handle h) { return h ? h->precision : -1; }

extern "C" int32_t pndf_lbs_num_joints(pndf_lbs_handle h) { return h ? NJ + h->NE : 0; }
extern "C" int32_t pndf_lbs_num_vertices(pndf_lbs_handle h) { return h ? h->V : 0; }

// Vertex ranges per chunk quad (blockIdx.y): the grid is quads x vsplit workgroups of one CU each, executed in rounds of
// `sm_count`.  Pick the split (<= 8, <= NG) whose last round is fullest -- 64 sequences x 300 frames are 320 quads: a split
// of 4 makes exactly 5 rounds where 1 or 2 leave the last round a quarter / half empty -- preferring, at equal fill, the
// smaller split (fewer partial sums).  Deterministic in (S, T) only.
static int lbs_vsplit(const pndf_lbs_model* h, int nch) {
    const long long quads = (nch + 3) / 4, sm = h->sm_count;
    const int vmax = h->NG < 8 ? h->NG : 8;
    if (quads * vmax <= sm) return vmax;          // less than one round whatever the split: take all the parallelism there is
    int best = 1;
    double best_fill = 0.0;
    for (int v = 1; v <= vmax; ++v) {
        const long long wgs = quads * v, rounds = (wgs + sm - 1) / sm;
        const double fill = (double)wgs / (double)(rounds * sm);
        if (fill > best_fill + 0.02) { best_fill = fill; best = v; }
    }
    return best;
}

// workspace layout (floats): pfp | Ap | Gt | gpf | gA | halo_pf | halo_A   (sized for the fused-terms mode, the largest)
static int64_t lbs_workspace(const pndf_lbs_model* h, int64_t S, int64_t T, PndfLbsArgs* a, float* base, int mode,
                             PndfLbsSplitArgs* sa = nullptr) {
    const int64_t N = S * T;
    const int cps = (mode == 1) ? chunks_pairs((int)T) : chunks_fwd((int)T);
    const int64_t nch = S * cps;
    const int vsplit = lbs_vsplit(h, (int)nch);
    int64_t off = 0;
    auto take = [&](int64_t n) { const int64_t o = off; off += (n + 3) & ~(int64_t)3; return base ? base + o : nullptr; };
    float* pfp = take(N * PF);
    float* Ap = take(N * 288);
    float* Gt = take(N * NJ * 3);
    float* gpf = take((int64_t)vsplit * N * PF);
    float* gA = take((int64_t)vsplit * N * A_FLOATS);
    float* hpf = take((int64_t)vsplit * nch * PF);
    float* hA = take((int64_t)vsplit * nch * A_FLOATS);
    float* red = take(N * RED_ROWS);
    float* pfs = take(N * (PNDF_LBS_PFS_HALFS / 2));       // split-precision B operands (halfs)
    float* Aps = take(N * (PNDF_LBS_APS_HALFS / 2));
    if (a) {
        a->pfp = pfp; a->Ap = Ap; a->Gt = Gt; a->gpf = gpf; a->gA = gA; a->red = red;
        a->halo_pf = (mode == 1) ? hpf : nullptr; a->halo_A = (mode == 1) ? hA : nullptr;
        a->cps = cps; a->vsplit = vsplit;
    }
    if (sa) { sa->pfs = pfs; sa->Aps = Aps; }
    return off;
}

extern "C" int64_t pndf_lbs_workspace_floats(pndf_lbs_handle h, int32_t S, int32_t T) {
    if (!h || S < 1 || T < 1) return 0;
    const int64_t a = lbs_workspace(h, S, T, nullptr, nullptr, 1), b = lbs_workspace(h, S, T, nullptr, nullptr, 2);
    return a > b ? a : b;
}

static int lbs_launch(pndf_lbs_model* h, int mode, PndfLbsArgs& a, void* workspace, void* stream) {
    if (((uintptr_t)workspace) & 15) return lbs_fail(h, PNDF_ERR_BAD_ARG, "workspace must be 16-byte aligned");
    a.blob = h->d_blob; a.V = h->V; a.NG = h->NG; a.NE = h->NE; a.model = h->consts;
    const bool split = h->precision == PNDF_LBS_F16X3 && mode != 2;
    PndfLbsSplitArgs sa;
    memset(&sa, 0, sizeof(sa));
    (void)lbs_workspace(h, a.S, a.T, &a, (float*)workspace, mode, &sa);
    DeviceGuard guard(h->device);
    if (!guard.ok) return lbs_fail(h, PNDF_ERR_HIP, "hipSetDevice failed");
    const long long N = (long long)a.S * a.T;
    const dim3 fgrid((unsigned)((N + 63) / 64)), fblock(64);
    if (mode == 0 && !a.verts && a.joints) {      // joints only: the chain + the picked vertices alone (lbs_joints_only_body)
        if (h->smpl_tree) hipLaunchKernelGGL(pndf_lbs_joints_only_smpl_kernel, fgrid, fblock, 0, (hipStream_t)stream, a, (const float*)h->d_picked);
        else hipLaunchKernelGGL(pndf_lbs_joints_only_kernel, fgrid, fblock, 0, (hipStream_t)stream, a, (const float*)h->d_picked);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return lbs_fail(h, PNDF_ERR_HIP, std::string("launch: ") + hipGetErrorString(e));
        return PNDF_OK;
    }
    const dim3 vgrid((unsigned)(((long long)a.S * a.cps + 3) / 4), (unsigned)(mode == 0 ? 1 : a.vsplit)), vblock(256);
    const int lds = 3 * BLOB * (int)sizeof(float);
    if (mode == 0) a.vsplit = split ? (h->NG < 8 ? h->NG : 8) : 1;      // (forward: the vertex ranges are independent outputs)
    if (split) {
        const dim3 sgrid((unsigned)(vgrid.x * (unsigned)a.vsplit));
        // operand scales (powers of two) from a-priori bounds: |d L / d verts| <= 2 w_temp + w_data per component (two pairs
        // share a vertex; the data term touches the picked vertices), |T_R^T g| <= 3 max_v sum_j |W| |g|, |v_posed| <= vp_bound
        const float gb = 2.0f * std::fabs(a.w_temp) + std::fabs(a.w_data);
        sa.base = a;
        sa.sblob = h->d_sblob;
        sa.a_scale = h->a_scale;
        sa.off_true = 1.0f / (h->p_scale * PNDF_LBS_PF_SCALE);
        sa.tm_true = 1.0f / (h->w_scale * h->a_scale);
        sa.g_scale = lbs_scale_for(3.0f * h->w_rowsum * gb);
        sa.x_scale = lbs_scale_for(gb * h->vp_bound);
        sa.gpf_true = (1.0f / h->p_scale) / sa.g_scale;
        sa.gA_true = (1.0f / h->w_scale) / sa.x_scale;
        const int lds_s = 3 * PNDF_LBS_SB_BYTES;
        if (h->smpl_tree) hipLaunchKernelGGL(pndf_lbs_pose_split_smpl_kernel, fgrid, fblock, 0, (hipStream_t)stream, sa);
        else hipLaunchKernelGGL(pndf_lbs_pose_split_kernel, fgrid, fblock, 0, (hipStream_t)stream, sa);
        if (mode == 0) {
            if (a.verts || (a.joints && a.NE > 0))
                hipLaunchKernelGGL(pndf_lbs_vertex_split_forward_kernel, sgrid, vblock, lds_s, (hipStream_t)stream, sa);
        } else {
            hipLaunchKernelGGL(pndf_lbs_vertex_split_terms_kernel, sgrid, vblock, lds_s, (hipStream_t)stream, sa);
            hipLaunchKernelGGL(pndf_lbs_reduce_partials_kernel, dim3((unsigned)((N + 63) / 64), (unsigned)((RED_ROWS + 63) / 64)), dim3(256), 0, (hipStream_t)stream, a);
            if (h->smpl_tree) {
                hipLaunchKernelGGL(pndf_lbs_pose_backward_smpl_kernel, fgrid, fblock, 0, (hipStream_t)stream, a);
                hipLaunchKernelGGL(pndf_lbs_rodrigues_vjp_kernel, dim3((unsigned)((N * (NJ - 1) + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
            } else hipLaunchKernelGGL(pndf_lbs_pose_backward_kernel, fgrid, fblock, 0, (hipStream_t)stream, a);
        }
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return lbs_fail(h, PNDF_ERR_HIP, std::string("launch: ") + hipGetErrorString(e));
        return PNDF_OK;
    }
    if (h->smpl_tree) hipLaunchKernelGGL(pndf_lbs_pose_smpl_kernel, fgrid, fblock, 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(pndf_lbs_pose_kernel, fgrid, fblock, 0, (hipStream_t)stream, a);
    if (mode == 0) {
        if (a.verts || (a.joints && a.NE > 0))
            hipLaunchKernelGGL(pndf_lbs_vertex_forward_kernel, vgrid, vblock, lds, (hipStream_t)stream, a);
    } else {
        if (mode == 1) hipLaunchKernelGGL(pndf_lbs_vertex_terms_kernel, vgrid, vblock, lds, (hipStream_t)stream, a);
        else hipLaunchKernelGGL(pndf_lbs_vertex_reverse_kernel, vgrid, vblock, lds, (hipStream_t)stream, a);
        hipLaunchKernelGGL(pndf_lbs_reduce_partials_kernel, dim3((unsigned)((N + 63) / 64), (unsigned)((RED_ROWS + 63) / 64)), dim3(256), 0, (hipStream_t)stream, a);
        if (h->smpl_tree) {
            hipLaunchKernelGGL(pndf_lbs_pose_backward_smpl_kernel, fgrid, fblock, 0, (hipStream_t)stream, a);
            hipLaunchKernelGGL(pndf_lbs_rodrigues_vjp_kernel, dim3((unsigned)((N * (NJ - 1) + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
        } else hipLaunchKernelGGL(pndf_lbs_pose_backward_kernel, fgrid, fblock, 0, (hipStream_t)stream, a);
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return lbs_fail(h, PNDF_ERR_HIP, std::string("launch: ") + hipGetErrorString(e));
    return PNDF_OK;
}

static void lbs_clear(PndfLbsArgs& a) { memset(&a, 0, sizeof(a)); }

extern "C" int pndf_lbs_forward(pndf_lbs_handle h, const float* theta, int64_t N, float* verts, float* joints, void* workspace,
                                void* stream) {
    PndfRange range("pndf_lbs_forward");
    if (!h) return PNDF_ERR_BAD_ARG;
    if (N < 0 || N > 0x7fffffff) return lbs_fail(h, PNDF_ERR_BAD_ARG, "bad frame count");
    if (N == 0) return PNDF_OK;
    if (!theta || !workspace || (!verts && !joints)) return lbs_fail(h, PNDF_ERR_BAD_ARG, "null pointer");
    PndfLbsArgs a;
    lbs_clear(a);
    a.theta = theta; a.verts = verts; a.joints = joints; a.S = 1; a.T = (int)N;
    return lbs_launch(h, 0, a, workspace, stream);
}

extern "C" int pndf_lbs_terms_grad_w(pndf_lbs_handle h, const float* theta, const float* joints0, int32_t S, int32_t T,
                                     float temp_coef, float data_coef, float* g_theta, void* workspace, void* stream) {
    PndfRange range("pndf_lbs_terms_grad_w");
    if (!h) return PNDF_ERR_BAD_ARG;
    if (S < 0 || T < 0) return lbs_fail(h, PNDF_ERR_BAD_ARG, "negative size");
    if (S == 0 || T == 0) return PNDF_OK;
    const bool data = data_coef != 0.f;
    if (!theta || !g_theta || !workspace || (data && !joints0)) return lbs_fail(h, PNDF_ERR_BAD_ARG, "null pointer");
    PndfLbsArgs a;
    lbs_clear(a);
    a.theta = theta; a.joints0 = joints0; a.g_theta = g_theta; a.S = S; a.T = T; a.it_gt0 = data ? 1 : 0;
    // the weights over the means of motion_denoise.py:89,94
    a.w_temp = T > 1 ? temp_coef / ((float)(T - 1) * (float)h->V) : 0.f;
    a.w_data = data ? data_coef / ((float)T * (float)(NJ + h->NE)) : 0.f;
    return lbs_launch(h, 1, a, workspace, stream);
}

extern "C" int pndf_lbs_terms_grad(pndf_lbs_handle h, const float* theta, const float* joints0, int32_t S, int32_t T, int32_t it,
                                   float* g_theta, void* workspace, void* stream) {
    if (!h) return PNDF_ERR_BAD_ARG;
    if (it < 0) return lbs_fail(h, PNDF_ERR_BAD_ARG, "negative iteration");
    // motion_denoise.py:31-32: temp 10 (1 + it), data 100 / (1 + it) for it > 0 (:92)
    return pndf_lbs_terms_grad_w(h, theta, joints0, S, T, 10.0f * (float)(1 + it), it > 0 ? 100.0f / (float)(1 + it) : 0.0f, g_theta,
                                 workspace, stream);
}

extern "C" int pndf_lbs_backward(pndf_lbs_handle h, const float* theta, const float* g_verts, const float* g_joints, int64_t N,
                                 float* g_theta, void* workspace, void* stream) {
    PndfRange range("pndf_lbs_backward");
    if (!h) return PNDF_ERR_BAD_ARG;
    if (N < 0 || N > 0x7fffffff) return lbs_fail(h, PNDF_ERR_BAD_ARG, "bad frame count");
    if (N == 0) return PNDF_OK;
    if (!theta || !g_theta || !workspace) return lbs_fail(h, PNDF_ERR_BAD_ARG, "null pointer");
    PndfLbsArgs a;
    lbs_clear(a);
    a.theta = theta; a.g_verts = g_verts; a.g_joints = g_joints; a.g_theta = g_theta; a.S = 1; a.T = (int)N;
    return lbs_launch(h, 2, a, workspace, stream);
}

PNDF_EXPORT_EXPERIMENT_WORD(lbs)
