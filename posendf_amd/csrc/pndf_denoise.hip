// Motion-denoise optimiser step around the distance engine (reference experiments/motion_denoise.py).
// Two HBM-bound kernels bracket the engine's forward+grad launch, so that one Adam step is three launches instead
// of ~60 PyTorch ones:
//   pndf_aa2quat_kernel      theta [N,69] -> q [N,21,4]                       (motion_denoise.py:81)
//   pndf_denoise_update_kernel   c_s = mean_t d (:83); d loss / d theta of the weighted terms (:29-45,88-94) with the
//                                axis-angle -> quaternion Jacobian applied to the engine's d d / d q; Adam (:70,98-99);
//                                and the NEXT step's quaternions.
// Terms: pose prior 1e7 c^2 / (1+it) on the engine's distances, and the pose-space surrogates of the reference's SMPL
// terms used when no body model is plugged in (posendf_amd/motion_denoise.py): temporal 10 (1+it) mean ||th_t - th_t+1||,
// data 100 / (1+it) mean ||th - th_0|| (it > 0 only, :92), norms over the 3-vector of each of the first 21 joints.
// One thread per (frame, joint); a workgroup = 12 frames of one sequence; every workgroup reduces its sequence's T
// distances itself (T floats from L2) rather than waiting for a separate reduction launch.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/posendf_amd.h"
#include "pndf_args.h"
#include "pndf_host.h"

namespace {

constexpr int NJ = 21, NJ_ALL = 23, TH = 69, NQ = 84;
constexpr int FRAMES_PER_WG = 11, WG = 256;      // 11 frames x 23 joints = 253 threads

// pytorch3d.transforms.axis_angle_to_quaternion restated from its documented convention (parity unpinned)
__device__ __forceinline__ void aa2quat(float ax, float ay, float az, float& k, float& angle, float (&q)[4]) {
    angle = sqrtf(ax * ax + ay * ay + az * az);
    const float half = 0.5f * angle;
    const bool small = angle < 1e-6f;
    k = small ? (0.5f - angle * angle / 48.0f) : (sinf(half) / angle);
    q[0] = cosf(half);
    q[1] = ax * k;
    q[2] = ay * k;
    q[3] = az * k;
}

}  // namespace

extern "C" __global__ void __launch_bounds__(256) pndf_aa2quat_kernel(const float* __restrict__ theta, float* __restrict__ q,
                                                                      long long N) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;     // (frame, joint)
    if (i >= N * NJ) return;
    const long long n = i / NJ;
    const int j = (int)(i - n * NJ);
    const float* a = theta + n * TH + 3 * j;
    float k, angle, qq[4];
    aa2quat(a[0], a[1], a[2], k, angle, qq);
    *(float4*)(q + n * NQ + 4 * j) = make_float4(qq[0], qq[1], qq[2], qq[3]);
}

extern "C" __global__ void __launch_bounds__(WG) pndf_denoise_update_kernel(PndfDenoiseArgs a) {
    __shared__ float red[WG];
    const int s = blockIdx.y, tid = threadIdx.x;
    const int T = a.T;
    // ---- c_s = mean_t d  (fixed order: strided partial sums, then a tree)
    float part = 0.f;
    for (int t = tid; t < T; t += WG) part += a.d[(long long)s * T + t];
    red[tid] = part;
    __syncthreads();
    for (int w = WG / 2; w > 0; w >>= 1) {
        if (tid < w) red[tid] += red[tid + w];
        __syncthreads();
    }
    const float c = red[0] / (float)T;

    // one thread per (frame, joint of the 23 of SMPL's body pose); the pose prior and the pose-space surrogates see the
    // first 21 (motion_denoise.py:81), the body-model terms (g_extra) all 23
    const int f = tid / NJ_ALL, j = tid - f * NJ_ALL;
    const int t = blockIdx.x * FRAMES_PER_WG + f;
    if (f >= FRAMES_PER_WG || t >= T) return;
    const long long n = (long long)s * T + t;
    const float* th = a.theta_in + n * TH + 3 * j;
    const float x = th[0], y = th[1], z = th[2];
    float gx = 0.f, gy = 0.f, gz = 0.f;

    if (j < NJ) {
        // ---- pose prior: d/d theta of 1e7 c^2 / (1+it)  =  2e7 c / ((1+it) T) * J^T(theta) d d/d q
        float k, angle, qq[4];
        aa2quat(x, y, z, k, angle, qq);
        const float4 g4 = *(const float4*)(a.dq + n * NQ + 4 * j);
        // k'(angle) / angle: (cos(angle/2)/2 - k) / angle^2, series -1/24 below the small-angle switch
        const float kp = (angle < 1e-6f) ? (-1.0f / 24.0f) : ((0.5f * qq[0] - k) / (angle * angle));
        const float gv_dot_a = g4.y * x + g4.z * y + g4.w * z;
        const float common = -0.5f * k * g4.x + kp * gv_dot_a;
        // d / d c of prior_coef c^p, spread over the T frames of the mean: p = 2 (motion_denoise.py:33: 1e7 c^2 / (1 + it)) or
        // p = 1 (partial_observation.py:33: 1e2 c / (1 + it))
        const float wp = (a.prior_power == 2 ? 2.0f * a.prior_coef * c : a.prior_coef) / (float)T;
        gx = wp * (common * x + k * g4.y);
        gy = wp * (common * y + k * g4.z);
        gz = wp * (common * z + k * g4.w);
    }
    if (a.g_extra) {
        // ---- body-model terms (SMPL vertex temporal + joint data term, motion_denoise.py:86-94): their weighted gradient
        // comes from pndf_lbs_terms_grad (pndf_lbs.hip); the order of the sum is pose prior first, as stacked at :37-45
        const float* ge = a.g_extra + n * TH + 3 * j;
        gx += ge[0]; gy += ge[1]; gz += ge[2];
    } else if (j < NJ) {
        // ---- temporal surrogate: temp_coef * mean_{t<T-1, j} sqrt(|th_t - th_t+1|^2 + 1e-20)
        if (T > 1) {
            const float wt = a.temp_coef / ((float)(T - 1) * (float)NJ);
            if (t + 1 < T) {
                const float* nx = th + TH;
                const float dx = x - nx[0], dy = y - nx[1], dz = z - nx[2];
                const float r = wt / sqrtf(dx * dx + dy * dy + dz * dz + 1e-20f);
                gx += r * dx; gy += r * dy; gz += r * dz;
            }
            if (t > 0) {
                const float* pv = th - TH;
                const float dx = pv[0] - x, dy = pv[1] - y, dz = pv[2] - z;
                const float r = wt / sqrtf(dx * dx + dy * dy + dz * dz + 1e-20f);
                gx -= r * dx; gy -= r * dy; gz -= r * dz;
            }
        }
        // ---- data surrogate (it > 0: data_coef != 0): data_coef * mean_{t, j} sqrt(|th - th0|^2 + 1e-20)
        if (a.data_coef != 0.f) {
            const float* t0 = a.theta0 + n * TH + 3 * j;
            const float dx = x - t0[0], dy = y - t0[1], dz = z - t0[2];
            const float r = a.data_coef / ((float)T * (float)NJ) / sqrtf(dx * dx + dy * dy + dz * dz + 1e-20f);
            gx += r * dx; gy += r * dy; gz += r * dz;
        }
    }
    // ---- Adam (torch.optim.Adam defaults: no amsgrad, no weight decay).  The two hand joints have a zero gradient
    // without a body model: m = v = 0 and the step is exactly 0, as in torch.
    const float bc1 = 1.0f - powf(a.beta1, (float)a.adam_step);
    const float bc2 = 1.0f - powf(a.beta2, (float)a.adam_step);
    const float step = a.lr / bc1, rs = 1.0f / sqrtf(bc2);
    float* mm = a.m + n * TH + 3 * j;
    float* vv = a.v + n * TH + 3 * j;
    const float g[3] = {gx, gy, gz};
    float nw[3];
    const float cur[3] = {x, y, z};
#pragma unroll
    for (int e = 0; e < 3; ++e) {
        const float m1 = a.beta1 * mm[e] + (1.0f - a.beta1) * g[e];
        const float v1 = a.beta2 * vv[e] + (1.0f - a.beta2) * g[e] * g[e];
        mm[e] = m1;
        vv[e] = v1;
        nw[e] = cur[e] - step * m1 / (sqrtf(v1) * rs + a.eps);
    }
    if (j < NJ) {
        float k2, ang2, q2[4];
        aa2quat(nw[0], nw[1], nw[2], k2, ang2, q2);
        *(float4*)(a.q_next + n * NQ + 4 * j) = make_float4(q2[0], q2[1], q2[2], q2[3]);
    }
    float* out = a.theta_out + n * TH + 3 * j;
    out[0] = nw[0]; out[1] = nw[1]; out[2] = nw[2];
}

// ------------------------------------------------------------------ C ABI (include/posendf_amd.h)
extern "C" int pndf_aa2quat(const float* theta, float* q, int64_t N, void* stream) {
    PndfRange range("pndf_aa2quat");
    if (N < 0 || (N > 0 && (!theta || !q))) return -1;
    if (((uintptr_t)q) & 15) return -1;
    if (N == 0) return 0;
    DeviceGuard guard(pndf_pointer_device(q));
    if (!guard.ok) return -3;
    const long long items = (long long)N * NJ;
    hipLaunchKernelGGL(pndf_aa2quat_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, (hipStream_t)stream, theta, q,
                       (long long)N);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

static int denoise_update(const float* theta_in, float* theta_out, const float* theta0, const float* d, const float* dq,
                          float* m, float* v, float* q_next, const float* g_extra, int32_t S, int32_t T,
                          const pndf_denoise_weights& w, int32_t adam_step, float lr, void* stream) {
    if (S < 0 || T < 0 || adam_step < 1 || (w.prior_power != 1 && w.prior_power != 2)) return -1;
    if (S == 0 || T == 0) return 0;
    if (!theta_in || !theta_out || !theta0 || !d || !dq || !m || !v || !q_next || theta_in == theta_out) return -1;
    if ((((uintptr_t)dq) | ((uintptr_t)q_next)) & 15) return -1;
    DeviceGuard guard(pndf_pointer_device(q_next));
    if (!guard.ok) return -3;
    PndfDenoiseArgs a;
    a.theta_in = theta_in; a.theta_out = theta_out; a.theta0 = theta0; a.d = d; a.dq = dq; a.m = m; a.v = v;
    a.q_next = q_next; a.g_extra = g_extra; a.S = S; a.T = T; a.adam_step = adam_step; a.prior_power = w.prior_power;
    a.prior_coef = w.prior_coef; a.temp_coef = w.temp_coef; a.data_coef = w.data_coef; a.reserved0 = 0;
    a.lr = lr; a.beta1 = 0.9f; a.beta2 = 0.999f; a.eps = 1e-8f;       // motion_denoise.py:70, torch.optim.Adam defaults
    const dim3 grid((unsigned)((T + FRAMES_PER_WG - 1) / FRAMES_PER_WG), (unsigned)S);
    hipLaunchKernelGGL(pndf_denoise_update_kernel, grid, dim3(WG), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// experiments/motion_denoise.py:29-35 at outer iteration `it`; the data term only for it > 0 (:92)
static pndf_denoise_weights motion_denoise_weights(int32_t it) {
    pndf_denoise_weights w;
    w.prior_coef = 1.0e7f / (float)(1 + it);
    w.prior_power = 2;
    w.temp_coef = 10.0f * (float)(1 + it);
    w.data_coef = it > 0 ? 100.0f / (float)(1 + it) : 0.0f;
    return w;
}

extern "C" int pndf_denoise_update(const float* theta_in, float* theta_out, const float* theta0, const float* d,
                                   const float* dq, float* m, float* v, float* q_next, int32_t S, int32_t T, int32_t it,
                                   int32_t adam_step, float lr, void* stream) {
    if (it < 0) return -1;
    return denoise_update(theta_in, theta_out, theta0, d, dq, m, v, q_next, nullptr, S, T, motion_denoise_weights(it), adam_step,
                          lr, stream);
}

extern "C" int pndf_denoise_update_w(const float* theta_in, float* theta_out, const float* theta0, const float* d, const float* dq,
                                     const float* g_body, float* m, float* v, float* q_next, int32_t S, int32_t T,
                                     const pndf_denoise_weights* w, int32_t adam_step, float lr, void* stream) {
    PndfRange range("pndf_denoise_update_w");
    if (!w) return -1;
    return denoise_update(theta_in, theta_out, theta0, d, dq, m, v, q_next, g_body, S, T, *w, adam_step, lr, stream);
}

extern "C" int pndf_denoise_update_body(const float* theta_in, float* theta_out, const float* theta0, const float* d,
                                        const float* dq, const float* g_body, float* m, float* v, float* q_next, int32_t S,
                                        int32_t T, int32_t it, int32_t adam_step, float lr, void* stream) {
    if (!g_body || it < 0) return -1;
    return denoise_update(theta_in, theta_out, theta0, d, dq, m, v, q_next, g_body, S, T, motion_denoise_weights(it), adam_step,
                          lr, stream);
}
