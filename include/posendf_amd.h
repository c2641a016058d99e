/* posendf_amd.h -- C ABI of the MI355X-native Pose-NDF distance / projection engine.
 *
 * The reference (garvita-tiwari/PoseNDF) has no FFI layer: its seam is the Python class
 * PoseNDF(nn.Module) (reference model/posendf.py:30-101) as called from
 * experiments/sample_poses.py:71-74 and experiments/motion_denoise.py:82.  These entry points are what a
 * binding for that seam needs (INTEGRATION.md shows the ctypes stub); each cites the reference lines it
 * replaces.  Plain pointers and sizes only -- no torch types.
 *
 * Conventions
 *   - all tensors are contiguous fp32; poses are [B, 21, 4] = [B, 84] row-major (model/posendf.py:64),
 *     distances are [B] (the reference's [B, 1]); device pointers must be 16-byte aligned;
 *   - every compute call enqueues work on the caller's HIP stream (`stream` is a hipStream_t passed as void*;
 *     NULL = the default stream) and returns without synchronising, allocating or freeing anything (safe under
 *     stream capture); pndf_create and pndf_load_weights allocate / synchronise;
 *   - every entry point runs on the device of its handle (stateless helpers: of their buffers) and restores the
 *     caller's current device before it returns;
 *   - the caller owns every buffer; the engine owns its packed copy of the weights and, for softplus, one derivative
 *     scratch (843 KB per compute unit, allocated by pndf_create).  Softplus launches of one handle share that scratch:
 *     a launch on another stream than the previous one first waits (on the device) for the previous one's completion.
 *     Launches recorded during stream capture are not tracked that way: replays of a captured softplus launch must be
 *     ordered by the caller against other launches of the same handle (use one handle per concurrently replayed graph);
 *   - return value: 0 on success, a negative pndf_status otherwise; pndf_last_error() has the text;
 *   - a handle belongs to one device and must not be used from two threads at once.
 */
#ifndef POSENDF_AMD_H
#define POSENDF_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pndf_engine* pndf_handle;

typedef enum {
    PNDF_OK = 0,
    PNDF_ERR_BAD_ARG = -1,        /* null / misaligned pointer, negative size */
    PNDF_ERR_BAD_SHAPE = -2,      /* weight tensor count or shape does not match the architecture */
    PNDF_ERR_HIP = -3,            /* a HIP runtime call failed (text in pndf_last_error) */
    PNDF_ERR_UNSUPPORTED = -4,    /* activation / architecture the kernels do not implement */
    PNDF_ERR_NO_WEIGHTS = -5,     /* compute call before pndf_load_weights */
    PNDF_ERR_NO_DEVICE = -6,      /* no gfx950 device visible */
    PNDF_ERR_HOST = -7            /* host twins only: a worker thread or an allocation failed (text in pndf_cpu_last_error) */
} pndf_status;

typedef enum { PNDF_ACT_RELU = 0, PNDF_ACT_LRELU = 1, PNDF_ACT_SOFTPLUS = 2 } pndf_act;

/* Mirrors the keys PoseNDF.__init__ actually reads (reference model/posendf.py:35-55,
 * model/network/net_modules.py:14-41,116-128; configs/amass.yaml:22-43). */
typedef struct {
    int32_t act;            /* pndf_act: model.DFNet.act (and model.StrEnc.act unless enc_act says otherwise) */
    float beta;             /* Softplus beta (amass.yaml:32,43); ignored for relu / lrelu */
    int32_t num_joints;     /* 21 */
    int32_t n_dims;         /* 8: in_dim, dims..., 1 */
    int32_t dims[16];       /* 126,256,512,1024,512,256,64,1 (amass.yaml:26,30); dims[0] = 84 selects the
                               encoder-less model (model.StrEnc.use = False, model/posendf.py:40-42,73-74).  `dims` is the
                               reference's free list (net_modules.py:14-28): six hidden widths within amass.yaml's run on the
                               fused kernels (narrower ones zero padded; softplus with a hidden width <= 8 under split precision: as the
                               next case); any other n_dims 3 .. 9 with hidden widths 1 .. 1024
                               runs on the runtime-planned kernels (exact fp32, or split-precision
                               fp16 MFMAs for PNDF_PREC_F16X3 / _F16); beyond that PNDF_ERR_UNSUPPORTED */
    int32_t parent[32];     /* net_utils.py:46 */
    int32_t precision;      /* pndf_precision: arithmetic of the trunk (engine knob, no reference counterpart) */
    int32_t enc_act;        /* model.StrEnc.act when it differs from model.DFNet.act (net_modules.py:128 reads its own key);
                               -1 (pndf_default_config) = the same as `act`.  A mixed pair runs on the runtime-planned kernels */
    float enc_beta;         /* model.StrEnc.beta for a softplus encoder; <= 0 = the same as `beta` */
} pndf_config;

/* PNDF_PREC_FP32 : exact fp32 MFMA (v_mfma_f32_16x16x4_f32), bit-comparable to an fmaf chain.
 * PNDF_PREC_F16X3: every fp32 operand split into fp16 hi + lo, three v_mfma_f32_16x16x32_f16 per product block,
 *                  fp32 accumulate: ~2^-22 relative product error (fp32-class), ~3x the throughput.
 *                  All scaling is by exact powers of two: weights per layer (any magnitude packs; pndf_load_weights
 *                  refuses -- PNDF_ERR_UNSUPPORTED -- only a trunk layer without a finite non-zero weight), activations
 *                  and gradients PER POSE from bounds measured on the chip or derived from the layers' norms, so that
 *                  no operand can overflow fp16 for any finite weights and poses and none is flushed for poses whose
 *                  gradients are tiny (small-gain networks).
 *                  When every trunk weight is exactly representable in fp16 at its layer scale (e.g. a checkpoint saved
 *                  in half precision), all lo halves of the weights are zero and the lo*hi term vanishes identically:
 *                  pndf_load_weights then selects the two-term kernels (2 MFMAs per block, no lo-tile reads) -- the same
 *                  results bit for bit, ~1.2x the throughput.  The environment variable PNDF_THREE_TERMS=1 keeps the
 *                  three-term kernels (A/B runs); pndf_kernel_name() tells which one a handle launches.
 * PNDF_PREC_F16  : plain fp16 operands (round to nearest), ONE MFMA per product block, fp32 accumulate.  A measured
 *                  comparison point (BASELINE.json configs[2] "fp32 vs bf16"): ~1e-3 relative, NOT within the 1e-4
 *                  parity bar of the two modes above; never selected implicitly; relu / lrelu only.
 * PNDF_PREC_BF16 : plain bfloat16 operands (round to nearest even), ONE v_mfma_f32_16x16x32_bf16 per product block, fp32
 *                  accumulate: the literal second half of BASELINE.json configs[2] "fp32 vs bf16".  Same schedule and rate
 *                  as PNDF_PREC_F16 with three fewer significant bits (~1e-2 relative): a measured comparison point only;
 *                  never selected implicitly; relu / lrelu only; amass.yaml-shaped networks only. */
typedef enum { PNDF_PREC_FP32 = 0, PNDF_PREC_F16X3 = 1, PNDF_PREC_F16 = 2, PNDF_PREC_BF16 = 3 } pndf_precision;

/* Fills cfg with the configs/amass.yaml architecture and the SMPL parent table. */
void pndf_default_config(pndf_config* cfg, int32_t act, float beta);

/* PoseNDF(opt) + .to(device) (model/posendf.py:32-55; sample_poses.py:88,93). */
int pndf_create(pndf_handle* out, const pndf_config* cfg, int device);
int pndf_destroy(pndf_handle h);

/* load_state_dict (sample_poses.py:90-91).  `tensors` are HOST pointers in state-dict order:
 * for i in 0..20: enc.net.i.net.0.weight [10,in], .bias [10], enc.net.i.net.2.weight [6,10], .bias [6];
 * then for l in 0..6: dfnet.lin{l}.weight [out,in] row-major, .bias [out]   (98 tensors; the 14 dfnet tensors
 * only when dims[0] == 84).
 * `numel[i]` is checked against the architecture.  Weights are re-packed into MFMA tile order and
 * uploaded; the call synchronises. */
int pndf_load_weights(pndf_handle h, const float* const* tensors, const int64_t* numel, int n_tensors);

/* PoseNDF.forward(pose, train=False)['dist_pred'] (model/posendf.py:62-76,100-101). */
int pndf_forward(pndf_handle h, const float* q, float* d, int64_t B, void* stream);

/* forward + torch.autograd.grad(d, q, grad_outputs) (model/posendf.py:18-27; sample_poses.py:71-73;
 * motion_denoise.py:82-83,97-98).  dq[b] = grad_out[b] * d d_b / d q_b; grad_out == NULL means ones. */
int pndf_forward_grad(pndf_handle h, const float* q, const float* grad_out, float* d, float* dq,
                      int64_t B, void* stream);

/* The projection loop of experiments/sample_poses.py:67-74: `steps` times q <- q - d(q) * grad d(q), in one
 * persistent launch.  q_out may alias q_in.  d_last[b] (may be NULL) is dist_pred of the last iteration. */
int pndf_project(pndf_handle h, const float* q_in, float* q_out, float* d_last, int64_t B, int steps,
                 void* stream);

/* ---- host twins (SURVEY.md 8b `pndf_*_cpu`): the same three operations on HOST pointers, for a caller whose
 * `train.device` is "cpu" (model/posendf.py:35,64 runs wherever the config says).  A separate handle type with no device
 * behind it: plain C++ on the host cores, fp32, PyTorch's activation conventions; blocks of 32 poses are dealt to threads
 * (PNDF_CPU_THREADS, default: all hardware threads).  Same configurations as pndf_create (all three activations, the
 * encoder-less model, narrower hidden layers; `precision` is ignored), same status codes; no stream argument: the calls
 * return when the result is in memory.  Never used as a fallback: the device entry points above fail when there is no
 * gfx950 device. */
typedef struct pndf_cpu_engine* pndf_cpu_handle;
int pndf_cpu_create(pndf_cpu_handle* out, const pndf_config* cfg);
int pndf_cpu_destroy(pndf_cpu_handle h);
int pndf_cpu_load_weights(pndf_cpu_handle h, const float* const* tensors, const int64_t* numel, int n_tensors);   /* as pndf_load_weights */
int pndf_forward_cpu(pndf_cpu_handle h, const float* q, float* d, int64_t B);                                       /* posendf.py:62-76 */
int pndf_forward_grad_cpu(pndf_cpu_handle h, const float* q, const float* grad_out, float* d, float* dq, int64_t B); /* posendf.py:18-27 */
int pndf_project_cpu(pndf_cpu_handle h, const float* q_in, float* q_out, float* d_last, int64_t B, int steps);      /* sample_poses.py:67-74 */
const char* pndf_cpu_last_error(pndf_cpu_handle h);   /* h may be NULL: last error of a failed pndf_cpu_create */

/* Host-only weight packer (what pndf_load_weights uploads); needs no device.  Output sizes in floats come
 * from pndf_packed_sizes.  Used by the CPU tests that check the MFMA tile order against a lane-level model. */
void pndf_packed_sizes(int64_t* stream_floats, int64_t* bias_floats);
int pndf_pack_host(const float* const* tensors, const int64_t* numel, int n_tensors, float* stream,
                   float* bias);
/* same for the split-precision stream (same size in bytes; trunk tiles hold fp16 pairs) */
int pndf_pack_host_split(const float* const* tensors, const int64_t* numel, int n_tensors, float* stream,
                         float* bias);

/* ---- motion-denoise optimiser step around the engine (experiments/motion_denoise.py:29-45,70-99).
 * Stateless helpers (no handle; status codes as above, no error text).  One Adam step =
 *   pndf_forward_grad(h, q, NULL, d, dq, S*T, stream);  pndf_denoise_update(...)   -- q for the first step from
 * pndf_aa2quat.  Terms: pose prior 1e7 c_s^2 / (1+it), c_s = mean_t d (:81-83,33) and the pose-space surrogates of the
 * SMPL temporal / data terms (:31-32,88-94; the body model itself is out of scope).  Adam(lr, 0.9, 0.999, 1e-8) (:70). */
/* theta [N,69] axis-angle (SMPL body pose) -> q [N,21,4] real-part-first quaternions of the first 21 joints
 * (pytorch3d.transforms.axis_angle_to_quaternion, :81). */
int pndf_aa2quat(const float* theta, float* q, int64_t N, void* stream);
/* theta_in/theta_out/theta0/m/v: [S,T,69]; d: [S*T]; dq, q_next: [S*T,21,4].  theta_in != theta_out (frames read their
 * neighbours; swap the buffers every step).  `it` = outer iteration (weights), `adam_step` = 1, 2, ... */
int pndf_denoise_update(const float* theta_in, float* theta_out, const float* theta0, const float* d, const float* dq,
                        float* m, float* v, float* q_next, int32_t S, int32_t T, int32_t it, int32_t adam_step, float lr,
                        void* stream);

/* The same step with EXPLICIT loss weights of the current outer iteration -- for callers whose schedule differs from
 * motion_denoise.py's, e.g. the copy of the loop in experiments/partial_observation.py:29-35 (temp 1e2 (1+it), data 1e1/(1+it),
 * pose prior 1e2 c/(1+it): LINEAR in c).  Objective: prior_coef * c^prior_power + temp_coef * temp + data_coef * data with
 * c = mean_t d; prior_power 1 or 2; data_coef 0 switches the data term off (the reference does for it == 0).  g_body NULL:
 * pose-space surrogates; otherwise the body-model gradient of pndf_lbs_terms_grad_w, evaluated with the same temp / data weights. */
typedef struct {
    float prior_coef;
    int32_t prior_power;
    float temp_coef;
    float data_coef;
} pndf_denoise_weights;
int pndf_denoise_update_w(const float* theta_in, float* theta_out, const float* theta0, const float* d, const float* dq,
                          const float* g_body, float* m, float* v, float* q_next, int32_t S, int32_t T,
                          const pndf_denoise_weights* w, int32_t adam_step, float lr, void* stream);

/* The same Adam step with the gradient of the BODY-MODEL terms (SMPL vertex temporal term + joint data term,
 * motion_denoise.py:86-94) in place of the pose-space surrogates: g_body [S,T,69] = d(weighted temp + data terms)/d theta as
 * produced by pndf_lbs_terms_grad below.  All 23 joints of the body pose are updated (the surrogates leave the two hand
 * joints alone; the body model moves vertices with them). */
int pndf_denoise_update_body(const float* theta_in, float* theta_out, const float* theta0, const float* d, const float* dq,
                             const float* g_body, float* m, float* v, float* q_next, int32_t S, int32_t T, int32_t it,
                             int32_t adam_step, float lr, void* stream);

/* ---- SMPL-shaped body model: linear-blend skinning and the body-model terms of the motion-denoise objective
 * (SURVEY 8f-3).  Replaces experiments/body_model.py:27-40 (smplx.SMPL called with betas, body_pose, global_orient = None)
 * and experiments/motion_denoise.py:86-94 (vertex temporal term, joint data term) with their reverse pass.  smplx and the
 * SMPL model files are third-party and absent from the reference: the published lbs() algorithm is restated, parity
 * unpinned.  The model is supplied by the caller as plain arrays (HOST pointers, fp32, the shapes of the SMPL model file):
 *   v_template [V,3]; shapedirs [V,3,NB] and betas [NB] (fixed during the optimisation, motion_denoise.py:27,67; NB may be 0);
 *   posedirs [207, 3V] (row k = entry k of the pose feature (R_1..R_23 - I), as smplx stores it); J_regressor [24,V];
 *   parents [24] (parents[0] = -1, every parent before its children); lbs_weights [V,24];
 *   extra_joint_vertex [n_extra <= 32]: joints picked from vertices (smplx VertexJointSelector; SMPL: 21 -> 45 joints).
 * Compute calls take DEVICE pointers, enqueue on `stream`, allocate nothing: scratch comes from the caller
 * (`workspace`: pndf_lbs_workspace_floats(h, S, T) floats, 16-byte aligned, for S sequences of T frames; the flat calls
 * with N frames use S = 1, T = N). */
typedef struct pndf_lbs_model* pndf_lbs_handle;
int pndf_lbs_create(pndf_lbs_handle* out, int32_t V, int32_t NB, const float* v_template, const float* shapedirs,
                    const float* betas, const float* posedirs, const float* J_regressor, const int32_t* parents,
                    const float* lbs_weights, const int32_t* extra_joint_vertex, int32_t n_extra, int device);
int pndf_lbs_destroy(pndf_lbs_handle h);
/* Arithmetic of the forward and fused-terms passes.  PNDF_LBS_F16X3 (default): the vertex-side contractions on fp16 MFMAs
 * with every fp32 operand split into fp16 hi + lo (three products per block, fp32 accumulate) -- the arithmetic of the
 * distance engine's f16x3 kernels, fp32-class accuracy; PNDF_LBS_FP32: the same on fp32 MFMAs (exact fp32 products).  The
 * general reverse pass (pndf_lbs_backward) always runs on fp32.  Returns 0 or a negative pndf_status. */
enum { PNDF_LBS_FP32 = 0, PNDF_LBS_F16X3 = 1 };
int pndf_lbs_set_precision(pndf_lbs_handle h, int32_t precision);
int32_t pndf_lbs_precision(pndf_lbs_handle h);         /* -1 for a NULL handle */
int32_t pndf_lbs_num_joints(pndf_lbs_handle h);        /* 24 + n_extra */
int32_t pndf_lbs_num_vertices(pndf_lbs_handle h);
int64_t pndf_lbs_workspace_floats(pndf_lbs_handle h, int32_t S, int32_t T);
/* BodyModel.forward(pose_body = theta, betas) (body_model.py:33-52): theta [N,69] -> verts [N,V,3] (may be NULL),
 * joints [N, 24 + n_extra, 3] (may be NULL; `Jtr` of the reference). */
int pndf_lbs_forward(pndf_lbs_handle h, const float* theta, int64_t N, float* verts, float* joints, void* workspace,
                     void* stream);
/* d / d theta of  10 (1 + it) mean_{t,v} |V[t,v] - V[t+1,v]|  +  [it > 0] 100 / (1 + it) mean_{t,j} |Jtr[t,j] - joints0[t,j]|
 * (motion_denoise.py:31-32,88-89,92-94) per sequence: theta, g_theta [S,T,69]; joints0 [S,T,24 + n_extra,3] (may be NULL
 * when it == 0).  One fused pass: vertices never reach HBM.  Like the reference there is no epsilon under the roots
 * (identical consecutive vertices give NaN). */
int pndf_lbs_terms_grad(pndf_lbs_handle h, const float* theta, const float* joints0, int32_t S, int32_t T, int32_t it,
                        float* g_theta, void* workspace, void* stream);
/* The same with explicit weights: d / d theta of temp_coef * temp + data_coef * data (data_coef == 0: no data term). */
int pndf_lbs_terms_grad_w(pndf_lbs_handle h, const float* theta, const float* joints0, int32_t S, int32_t T, float temp_coef,
                          float data_coef, float* g_theta, void* workspace, void* stream);
/* General reverse pass: g_theta [N,69] = d (<g_verts, verts> + <g_joints, joints>) / d theta (either may be NULL). */
int pndf_lbs_backward(pndf_lbs_handle h, const float* theta, const float* g_verts, const float* g_joints, int64_t N,
                      float* g_theta, void* workspace, void* stream);
/* Host-only model packer (what pndf_lbs_create uploads; needs no device): `blob` takes pndf_lbs_packed_floats(V) floats in
 * MFMA tile order, J_out / rel_out (72 floats each, may be NULL) the rest joints and their parent-relative offsets. */
int64_t pndf_lbs_packed_floats(int32_t V);
int pndf_lbs_pack_host(int32_t V, int32_t NB, const float* v_template, const float* shapedirs, const float* betas,
                       const float* posedirs, const float* J_regressor, const int32_t* parents, const float* lbs_weights,
                       const int32_t* extra_joint_vertex, int32_t n_extra, float* blob, float* J_out, float* rel_out);
/* The model for PNDF_LBS_F16X3 from the packed fp32 one (host only): sblob takes pndf_lbs_packed_split_bytes(V) bytes -- per
 * 16 vertices, planes of fp16 hi / lo halves [row][16 v] of posedirs x p_scale and weights x w_scale (powers of two, returned in
 * scales_out[0..1], may be NULL), laid out in bank-conflict-free tiles (csrc/pndf_lbs_split.h). */
int64_t pndf_lbs_packed_split_bytes(int32_t V);
int pndf_lbs_pack_split_host(int32_t V, const float* blob, void* sblob, float* scales_out);
const char* pndf_lbs_last_error(pndf_lbs_handle h);    /* h may be NULL: last error of a failed pndf_lbs_create */

/* ---- quaternion pose distance + k nearest candidates (data/dist_utils.py:9-50, classes euc / geo; caller
 * data/prepare_traindata.py:159).  noise [B,21,4], valid [B,K,21,4] (device, 16-byte aligned);
 * metric 0 = geo: sum_j w_j (1 - |<q_valid_j, q_noise_j>|), 1 = euc: sum_j w_j ||q_noise_j - q_valid_j||;
 * weights = 21 HOST floats (the reference's normalised joint ranks) or NULL for the unweighted mean;
 * vals [B,k] ascending, idx [B,k] int64, ties towards the lower index; k <= min(K, 16), K <= ~1850. */
int pndf_quat_topk(const float* noise, const float* valid, int64_t B, int32_t K, int32_t metric, const float* weights,
                   int32_t k, float* vals, long long* idx, void* stream);

const char* pndf_last_error(pndf_handle h);   /* h may be NULL: last error of a failed pndf_create */
const char* pndf_version(void);
/* 0 in every product build: one bit per compile-time tuning / ablation macro that differed from its product default when the
 * library was built (posendf_amd/csrc/pndf_experiment.h); pndf_version() prints it as `experiments=0x........`.  A library that
 * reports anything else is a lab build (tools/build_variants.py) and may compute wrong results on purpose. */
unsigned pndf_experiment_word(void);
/* Name of the device kernel the compute calls of this handle launch (valid after pndf_load_weights): for logs, profiles
 * and tests; "" for a NULL handle. */
const char* pndf_kernel_name(pndf_handle h);

#ifdef __cplusplus
}
#endif
#endif /* POSENDF_AMD_H */
