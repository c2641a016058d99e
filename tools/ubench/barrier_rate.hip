// What does one s_barrier cost between groups of MFMAs (no LDS, no DMA)?  4 or 8 waves per workgroup.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x4 mf(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

template <int NM, int BAR, int NT>
__global__ void __launch_bounds__(NT) k(float* out, unsigned long long* cyc, int iters) {
    const int lane = threadIdx.x & 63;
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(0.01f * (lane + j)); b[j] = (_Float16)(0.02f * j); }
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < NM; ++m) acc[m % 8] = mf(a, b, acc[m % 8]);
        if (BAR) asm volatile("s_barrier" ::: "memory");
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1];
    out[blockIdx.x * NT + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * (NT / 64) + threadIdx.x / 64] = t1 - t0;
}
template <int NM, int BAR, int NT>
void run(float* out, unsigned long long* cyc) {
    const int iters = 4000, grid = 256, nw = NT / 64;
    hipLaunchKernelGGL((k<NM, BAR, NT>), dim3(grid), dim3(NT), 0, 0, out, cyc, iters);
    (void)hipDeviceSynchronize();
    static unsigned long long h[256 * 8];
    (void)hipMemcpy(h, cyc, grid * nw * 8, hipMemcpyDeviceToHost);
    double mean = 0, mx = 0;
    for (int i = 0; i < grid * nw; ++i) { mean += h[i]; if (h[i] > mx) mx = h[i]; }
    mean /= grid * nw;
    printf("%d waves, %2d MFMAs/iter, barrier %d: %8.1f cycles/iter (max wave %8.1f), MFMA floor %d per SIMD\n", nw, NM, BAR,
           mean / iters, mx / iters, NM * 16 * (nw / 4));
}
int main() {
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 256 * 8 * 8);
    run<24, 0, 256>(out, cyc); run<24, 1, 256>(out, cyc);
    run<12, 0, 256>(out, cyc); run<12, 1, 256>(out, cyc);
    run<12, 0, 512>(out, cyc); run<12, 1, 512>(out, cyc);
    run<24, 0, 512>(out, cyc); run<24, 1, 512>(out, cyc);
    return 0;
}
