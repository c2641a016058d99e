// Micro-benchmark: issue rate of v_mfma_f32_16x16x4_f32 in the access pattern of the fused kernel
// (one wave per SIMD, A operands prefetched from LDS one group ahead, NACC round-robin accumulators).
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_rate.hip -o gpurun_out/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, bool LDS, int GAP>
__global__ void __launch_bounds__(256, 1) k(float* out, unsigned long long* cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16384 / 4; i += 256) ((float*)smem)[i] = 1.0f + i * 1e-6f;
    __syncthreads();
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    f32x4 b[8];
    for (int i = 0; i < 8; ++i) b[i] = f32x4{1.f + lane, 2.f, 3.f, 4.f};
    f32x4 cur[4], nxt[4];
    for (int i = 0; i < 4; ++i) cur[i] = *(f32x4*)(smem + i * 1024 + lane * 16);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int grp = 0; grp < 4; ++grp) {
            if (LDS) {
#pragma unroll
                for (int i = 0; i < 4; ++i) nxt[i] = *(f32x4*)(smem + ((grp * 4 + i) & 15) * 1024 + lane * 16);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) nxt[i] = cur[i] * 1.0001f;
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                acc[m % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[m / 4][m % 4], b[(m / 2) % 8][m % 4], acc[m % NACC], 0, 0, 0);
                if (GAP > 0 && (m % GAP) == GAP - 1) asm volatile("s_nop 0");
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i) cur[i] = nxt[i];
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC, bool LDS, int GAP>
void run(const char* name, float* out, unsigned long long* cyc) {
    const int iters = 2000, grid = 256;
    hipLaunchKernelGGL((k<NACC, LDS, GAP>), dim3(grid), dim3(256), 65536, 0, out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h[256];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double mean = 0;
    for (int i = 0; i < grid; ++i) mean += h[i];
    mean /= grid;
    printf("%-28s %8.3f cycles / MFMA\n", name, mean / (iters * 64.0));
}

int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    hipFuncSetAttribute((const void*)k<2, true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    run<2, true, 0>("2 acc, LDS prefetch", out, cyc);
    run<4, true, 0>("4 acc, LDS prefetch", out, cyc);
    run<8, true, 0>("8 acc, LDS prefetch", out, cyc);
    run<2, false, 0>("2 acc, no LDS", out, cyc);
    run<4, false, 0>("4 acc, no LDS", out, cyc);
    run<1, false, 0>("1 acc, no LDS", out, cyc);
    run<2, true, 4>("2 acc, LDS, s_nop every 4", out, cyc);
    return 0;
}
