// Quaternion pose distance + k-smallest (reference data/dist_utils.py:9-50, caller
// data/prepare_traindata.py:159): for every query pose, the distance to each of its K candidate poses
//   geo: sum_j w_j (1 - |<q_valid_j, q_noise_j>|)        euc: sum_j w_j ||q_noise_j - q_valid_j||_2
// (w_j = 1/21, or the L2-normalised joint ranks) and the k smallest with their indices.
// HBM-bound: K * 336 B are read once per query; everything else stays on chip.  One workgroup per query:
//   pass 1  the query's K*21 candidate quaternions are read as ONE contiguous float4 stream (fully coalesced,
//           16 B per lane); each lane turns its quaternion into a per-joint term and parks it in LDS
//   pass 2  one lane per candidate sums its 21 terms in joint order (LDS stride 21 words: conflict-free)
//   pass 3  k rounds of (value, index) arg-min over the workgroup: wave shuffles, then 4 partials through LDS.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pndf_args.h"
#include "pndf_host.h"

namespace {
constexpr int NJ = 21, WG = 256, MAX_K_OUT = 16;
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void argmin_pair(float& v, int& i, float ov, int oi) {
    if (ov < v || (ov == v && oi < i)) { v = ov; i = oi; }
}
}  // namespace

extern "C" __global__ void __launch_bounds__(WG) pndf_quat_topk_kernel(PndfQuatDistArgs a) {
    extern __shared__ float smem[];                   // terms[K*21] | dist[K] | q[21*4] | w[21] | partials
    const int K = a.K, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* terms = smem;
    float* dist = terms + (size_t)K * NJ;
    f4* qn = (f4*)(dist + ((K + 3) & ~3));
    float* wj = (float*)(qn + NJ);
    float* pv = wj + 24;                              // 4 partial minima
    int* pi = (int*)(pv + 4);
    const long long b = blockIdx.x;
    if (tid < NJ) {
        qn[tid] = ((const f4*)a.noise)[b * NJ + tid];
        wj[tid] = a.w[tid];
    }
    __syncthreads();
    // ---- pass 1
    const f4* src = (const f4*)a.valid + b * (long long)K * NJ;
    const int n = K * NJ;
    auto term = [&](const f4& v, int e) {
        const int j = e % NJ;
        const f4 q = qn[j];
        float t;
        if (a.metric == 0) {
            t = 1.0f - fabsf(v.x * q.x + v.y * q.y + v.z * q.z + v.w * q.w);
        } else {
            const float dx = q.x - v.x, dy = q.y - v.y, dz = q.z - v.z, dw = q.w - v.w;
            t = sqrtf(dx * dx + dy * dy + dz * dz + dw * dw);
        }
        terms[e] = t * wj[j];
    };
    constexpr int UN = 8;                             // 8 independent 16-byte loads in flight per lane
    int e0 = tid;
    for (; e0 + (UN - 1) * WG < n; e0 += UN * WG) {
        f4 v[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) v[u] = __builtin_nontemporal_load(src + e0 + u * WG);
#pragma unroll
        for (int u = 0; u < UN; ++u) term(v[u], e0 + u * WG);
    }
    for (; e0 < n; e0 += WG) term(src[e0], e0);
    __syncthreads();
    // ---- pass 2
    for (int c = tid; c < K; c += WG) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) s += terms[c * NJ + j];
        dist[c] = s;
    }
    __syncthreads();
    // ---- pass 3
    for (int r = 0; r < a.k; ++r) {
        float v = __builtin_inff();
        int i = 0x7fffffff;
        for (int c = tid; c < K; c += WG) argmin_pair(v, i, dist[c], c);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_xor(v, off);
            const int oi = __shfl_xor(i, off);
            argmin_pair(v, i, ov, oi);
        }
        if (lane == 0) { pv[wave] = v; pi[wave] = i; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < WG / 64; ++w) argmin_pair(v, i, pv[w], pi[w]);
            const bool found = i < K;                   // false when fewer than r + 1 distances compare (NaN inputs)
            a.vals[b * a.k + r] = found ? v : __builtin_nanf("");
            a.idx[b * a.k + r] = found ? i : -1;
            if (found) dist[i] = __builtin_nanf("");   // taken: NaN never compares, a genuine +inf distance still can
        }
        __syncthreads();
    }
}

// dist_calc of data/dist_utils.py (classes geo / euc): metric 0 = geo, 1 = euc; `weights` = 21 host floats or NULL
// (unweighted mean over joints).  vals [B,k] ascending, idx [B,k] int64 (ties -> lower index).
extern "C" int pndf_quat_topk(const float* noise, const float* valid, int64_t B, int32_t K, int32_t metric,
                              const float* weights, int32_t k, float* vals, long long* idx, void* stream) {
    PndfRange range("pndf_quat_topk");
    if (B < 0 || K < 1 || k < 1 || k > K || k > MAX_K_OUT || (metric != 0 && metric != 1)) return -1;
    if (B == 0) return 0;
    if (!noise || !valid || !vals || !idx) return -1;
    if ((((uintptr_t)noise) | ((uintptr_t)valid)) & 15) return -1;
    const size_t lds = ((size_t)K * NJ + ((K + 3) & ~3) + NJ * 4 + 24 + 8) * sizeof(float);
    if (lds > 160 * 1024) return -4;                  // K <= ~1,850 candidates per query
    DeviceGuard guard(pndf_pointer_device(valid));
    if (!guard.ok) return -3;
    // the dynamic-LDS limit is a per-device function attribute; setting it is a host-side table write (no device work), so
    // it is simply set on every call: no shared state between threads, no bound on the device index
    if (hipFuncSetAttribute((const void*)pndf_quat_topk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
        return -3;
    PndfQuatDistArgs a;
    a.noise = noise; a.valid = valid; a.vals = vals; a.idx = idx; a.K = K; a.k = k; a.metric = metric;
    for (int j = 0; j < NJ; ++j) a.w[j] = weights ? weights[j] : 1.0f / (float)NJ;
    hipLaunchKernelGGL(pndf_quat_topk_kernel, dim3((unsigned)B), dim3(WG), lds, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
