// Micro-benchmark: do VALU instructions of ONE wave overlap with its own fp16 MFMAs, and does it matter where they sit in
// program order?  One wave per SIMD (256 threads, one workgroup per CU), registers only.  Per trip of the loop: 12 MFMAs
// (v_mfma_f32_16x16x32_f16) and NV independent VALU instructions (v_fma_f32 on private registers), in several arrangements:
//   0  MFMAs only (chains of three on one accumulator, four accumulators)         3  MMM v.. MMM v.. (VALU behind every chain)
//   1  VALU only                                                                   4  M v M v M v (VALU behind every MFMA)
//   2  all 12 MFMAs, then all the VALU                                             5  as 4, every MFMA on its own accumulator
// Cycles per trip from s_memtime (100 MHz constant clock scaled by the measured shader clock is avoided: the ratio between
// the arrangements is what counts).
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_valu_overlap.hip -o gpurun_ab/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int ARR, int NV>
__global__ void __launch_bounds__(256, 1) k(float* out, unsigned long long* cyc, int iters) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(1.0f + 0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.5f + 0.002f * i); }
    f32x4 acc[12];
    for (int i = 0; i < 12; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = 0.001f * threadIdx.x + i;
    const float m1 = 1.0001f, m2 = 0.0001f;
    auto valu = [&](int n0, int n1) __attribute__((always_inline)) {
#pragma unroll
        for (int n = n0; n < n1; ++n) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[n % 8]) : "v"(m1), "v"(m2));
    };
    auto mfma = [&](int m) __attribute__((always_inline)) {
        const int ic = (ARR == 5) ? m : m / 3;
        acc[ic] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[ic], 0, 0, 0);
    };
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (ARR == 0) {
#pragma unroll
            for (int m = 0; m < 12; ++m) { mfma(m); __builtin_amdgcn_sched_barrier(0); }
        } else if (ARR == 1) {
            valu(0, NV);
        } else if (ARR == 2) {
#pragma unroll
            for (int m = 0; m < 12; ++m) { mfma(m); __builtin_amdgcn_sched_barrier(0); }
            valu(0, NV);
        } else if (ARR == 3) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                mfma(3 * c); __builtin_amdgcn_sched_barrier(0); mfma(3 * c + 1); __builtin_amdgcn_sched_barrier(0); mfma(3 * c + 2);
                __builtin_amdgcn_sched_barrier(0);
                valu(c * NV / 4, (c + 1) * NV / 4);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int m = 0; m < 12; ++m) {
                mfma(m);
                __builtin_amdgcn_sched_barrier(0);
                valu(m * NV / 12, (m + 1) * NV / 12);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0;
    for (int i = 0; i < 12; ++i) r += acc[i][0];
    for (int i = 0; i < 8; ++i) r += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int ARR, int NV> void run(const char* what, float* out, unsigned long long* cyc, int ncu) {
    const int iters = 20000;
    hipLaunchKernelGGL((k<ARR, NV>), dim3(ncu), dim3(256), 0, 0, out, cyc, 100);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<ARR, NV>), dim3(ncu), dim3(256), 0, 0, out, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-58s NV=%3d  %8.1f ns per trip  (memtime %6.1f ticks)\n", what, NV, ms * 1e6 / iters, (double)c / iters);
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int ncu = p.multiProcessorCount;
    float* out; unsigned long long* cyc;
    hipMalloc(&out, ncu * 256 * 4); hipMalloc(&cyc, ncu * 8);
    run<0, 0>("12 MFMAs (chains of 3)", out, cyc, ncu);
    run<5, 0>("12 MFMAs (12 accumulators)", out, cyc, ncu);
    run<1, 24>("VALU only", out, cyc, ncu);
    run<1, 48>("VALU only", out, cyc, ncu);
    run<2, 24>("12 MFMAs then VALU", out, cyc, ncu);
    run<3, 24>("MMM v.. (behind every chain)", out, cyc, ncu);
    run<4, 24>("M v.. (behind every MFMA)", out, cyc, ncu);
    run<5, 24>("M v.. own accumulators", out, cyc, ncu);
    run<2, 48>("12 MFMAs then VALU", out, cyc, ncu);
    run<3, 48>("MMM v.. (behind every chain)", out, cyc, ncu);
    run<4, 48>("M v.. (behind every MFMA)", out, cyc, ncu);
    run<5, 48>("M v.. own accumulators", out, cyc, ncu);
    return 0;
}
