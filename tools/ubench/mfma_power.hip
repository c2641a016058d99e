// Micro-benchmark: SUSTAINED rate of dense fp16 MFMA under the chip's power limit, for the two instruction shapes and
// for operand data of different "busyness".  No memory traffic at all: one wave per SIMD (as in the fused kernel) issues
// MFMAs back to back from registers for ~40 ms per case.  Question behind it: the split-precision
// kernel runs at MFMA-busy x clock ~ 1.17 GHz whatever its cycle count -- is that the MFMA power wall, and would the
// 32x32x16 shape (twice the MACs per operand element read) sit higher?
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_power.hip -o gpurun_ab/mfma_power
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned rnd(unsigned& s) { s = s * 1664525u + 1013904223u; return s; }
// a half in [0.5, 2) with random mantissa and sign: exponent field 14 or 15
__device__ __forceinline__ unsigned half_bits(unsigned r) { return (r & 0x83ffu) | (((r >> 10) & 1u) ? 0x3c00u : 0x3800u); }
template <int MODE> __device__ __forceinline__ f16x8 make_operand(unsigned& s) {
    u32x4 v;
    for (int i = 0; i < 4; ++i) {
        unsigned r = rnd(s);
        if (MODE == 0) v[i] = 0x3c003c00u;                              // all ones (1.0): no toggling between operands
        else v[i] = half_bits(r) | (half_bits(r >> 16) << 16);          // random sign / mantissa
    }
    return __builtin_bit_cast(f16x8, v);
}

// PAT: which operand registers consecutive MFMAs use.  0: A and B both change every MFMA (8 register sets each);
// 1: B fixed, A changes every MFMA; 2: A fixed, B changes every MFMA; 3: kernel-like part B (B fixed for 4, then the
// other half of the same block; A changes every MFMA); 4: A changes every second MFMA, B alternates between two sets
template <int SHAPE, int MODE, int PAT = 0>
__global__ void __launch_bounds__(256, 1) k(float* out, unsigned long long* cyc, int iters) {
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 97u + 12345u;
    f16x8 a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = make_operand<MODE>(s); b[i] = make_operand<MODE>(s); }
    f32x4 c4[8];
    f32x16 c16[4];
    for (int i = 0; i < 8; ++i) c4[i] = f32x4{0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) c16[i][j] = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 32; ++m) {
            constexpr int dummy = 0;
            // 5: a wave owning 32 poses (A kept for four MFMAs: xh0 xl0 xh1 xl1); 6: pattern 4 with the two MFMAs that share A
            // also sharing the accumulator (the kernel's hh_i, hl_i); 7: pattern 0 on two alternating accumulators only
            const int ia = PAT == 2 ? 0 : (PAT == 4 || PAT == 6) ? (m / 2) % 8 : PAT == 5 ? (m / 4) % 8
                           : PAT == 20 ? ((m / 3) * 2 + (m % 3 == 2)) % 8 : m % 8;
            const int ib = PAT == 0 || PAT == 7 ? (m + m / 8) % 8 : PAT == 1 ? 0 : PAT == 2 ? m % 8 : PAT == 3 ? (m / 4) % 2
                           : PAT == 5 ? m % 4 : m % 2;
            // 10 + N: pattern 0 rotating over N accumulators (N = 1 .. 4): how close must the producer of C be?
            // 20: the kernel's pair-major order (chains of three on one accumulator, A kept for the first two, eight accumulators)
            const int ic = PAT == 6 ? (m / 2) % 8 : PAT == 7 ? m % 2 : PAT > 10 && PAT < 20 ? m % (PAT - 10) : PAT == 20 ? (m / 3) % 8 : m % 8;
            if (SHAPE == 16) c4[ic] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[ia], b[ib], c4[ic], 0, 0, 0);
            else c16[m % 4] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m % 8], b[(m + m / 8) % 8], c16[m % 4], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0;
    for (int i = 0; i < 8; ++i) r += c4[i][0];
    for (int i = 0; i < 4; ++i) r += c16[i][0];
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int SHAPE, int MODE, int PAT = 0>
void run(const char* name, float* out, unsigned long long* cyc) {
    const int grid = 1024;                                   // four rounds over 256 CUs, like the benchmark batch
    const double flop_per_mfma = (SHAPE == 16) ? 2.0 * 16 * 16 * 32 : 2.0 * 32 * 32 * 16;
    const int iters = (SHAPE == 16) ? 20000 : 10000;         // 32 MFMAs per iteration: ~10 ms per round at full clock
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<SHAPE, MODE, PAT>), dim3(grid), dim3(256), 0, 0, out, cyc, iters);     // warm-up
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<SHAPE, MODE, PAT>), dim3(grid), dim3(256), 0, 0, out, cyc, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    static unsigned long long h[1024];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double mean = 0;
    for (int i = 0; i < grid; ++i) mean += h[i];
    mean /= grid;
    const double mfmas = (double)grid * 4 * iters * 32;
    const double cyc_per = mean / (iters * 32.0);
    printf("%-34s %7.2f ms  %8.1f TFLOP/s issued  %6.2f cycles/MFMA  clock %.2f GHz (busy x clock %.2f GHz)\n", name, ms,
           mfmas * flop_per_mfma / (ms * 1e-3) / 1e12, cyc_per, mean * 4 / (ms * 1e-3) / 1e9,
           ((SHAPE == 16) ? 16.0 : 32.0) / cyc_per * mean * 4 / (ms * 1e-3) / 1e9);
}

int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&cyc, 1024 * 8);
    run<16, 1>("16x16x32 f16, random operands", out, cyc);
    run<32, 1>("32x32x16 f16, random operands", out, cyc);
    run<16, 0>("16x16x32 f16, constant operands", out, cyc);
    run<32, 0>("32x32x16 f16, constant operands", out, cyc);
    run<16, 1>("16x16x32 f16, random (again)", out, cyc);
    run<16, 1, 1>("16x16x32 random, B fixed", out, cyc);
    run<16, 1, 2>("16x16x32 random, A fixed", out, cyc);
    run<16, 1, 3>("16x16x32 random, B fixed x4", out, cyc);
    run<16, 1, 4>("16x16x32 random, A x2, B alt", out, cyc);
    run<16, 1, 5>("16x16x32 random, A x4, B cyc4", out, cyc);
    run<16, 1, 6>("16x16x32 random, A x2 + acc x2", out, cyc);
    run<16, 1, 7>("16x16x32 random, 2 accumulators", out, cyc);
    run<16, 1, 11>("16x16x32 random, 1 accumulator", out, cyc);
    run<16, 1, 13>("16x16x32 random, 3 accumulators", out, cyc);
    run<16, 1, 14>("16x16x32 random, 4 accumulators", out, cyc);
    run<16, 1, 20>("16x16x32 random, kernel order 3", out, cyc);
    run<16, 1, 0>("16x16x32 f16, random (3rd)", out, cyc);
    return 0;
}
