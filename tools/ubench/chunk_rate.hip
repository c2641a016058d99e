// Micro-benchmark of ONE (lin2,lin3) chunk body of the fused kernel without barriers / DMA: 16 part-A groups
// (2 chunk accumulators, B operands = 128 resident VGPRs) + epilogue + 16 part-B groups (32 accumulator
// tiles), tiles prefetched from LDS one group ahead.  Ideal: 512 MFMAs x 32 = 16,384 cycles per chunk.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

struct R { const char* g; char* smem; int cur; int next; int wave; int lane; };

template <int T0, int MODE>
__device__ __forceinline__ void load_group(f32x4 (&a)[4], R& r) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int t = (T0 + i) % 16;
        if ((MODE & 16) && t == 0) r.cur = (r.cur == 2) ? 0 : r.cur + 1;
        if (t == 8) {
            if (MODE & 8) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (MODE & 4) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
            if (MODE & 64) {
                // saddr form: uniform 64-bit base in SGPRs + 32-bit per-lane offset
                const char* sbase = r.g + (size_t)r.next * 16384 + r.wave * 4096;
                const unsigned voff = r.lane * 16;
                const unsigned dst = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(r.smem) + ((r.cur == 0) ? 2 : r.cur - 1) * 16384 + r.wave * 4096;
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\t"
                             "global_load_lds_dwordx4 %1, %3 offset:1024\n\tglobal_load_lds_dwordx4 %1, %3 offset:2048\n\t"
                             "global_load_lds_dwordx4 %1, %3 offset:3072\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(dst), "s"(sbase) : "memory");
                r.next = (r.next + 1 == 664) ? 0 : r.next + 1;
            } else if (MODE & 8) {
                const char* src = r.g + (size_t)r.next * 16384 + r.wave * 4096 + r.lane * 16;
                const unsigned dst = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(r.smem) + ((r.cur == 0) ? 2 : r.cur - 1) * 16384 + r.wave * 4096;
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                             "global_load_lds_dwordx4 %1, off offset:1024\n\tglobal_load_lds_dwordx4 %1, off offset:2048\n\t"
                             "global_load_lds_dwordx4 %1, off offset:3072\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
                r.next = (r.next + 1 == 664) ? 0 : r.next + 1;
            }
        }
        a[i] = *(const f32x4*)(r.smem + r.cur * 16384 + t * 1024 + r.lane * 16);
    }
}

template <int GI, int MODE>
__device__ __forceinline__ void groups(const f32x4 (&xin)[32], f32x4 (&acc)[32], f32x4 (&ch)[2], f32x4 (&cur)[4],
                                       R& base, int lane, float slope, int c, int nc) {
    if constexpr (GI < 32) {
        f32x4 nxt[4];
        constexpr int TN = ((GI + 1) * 4) % 128;
        if (GI + 1 < 32 || c + 1 < nc) load_group<TN, MODE>(nxt, base);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (GI < 16) {
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int ci = 0; ci < 2; ++ci) ch[ci] = mfma4(cur[k2 * 2 + ci][s], xin[2 * GI + k2][s], ch[ci]);
            if constexpr (GI == 15 && (MODE & 32)) {
                // bias for the next chunk + mask store like the real kernel
                ((unsigned short*)(base.smem + 49152))[c * 256 + threadIdx.x] = (unsigned short)(ch[0][0] > 0);
            }
            if constexpr (GI == 15 && (MODE & 1) == 0) {
#pragma unroll
                for (int ci = 0; ci < 2; ++ci)
#pragma unroll
                    for (int r = 0; r < 4; ++r) ch[ci][r] = ch[ci][r] * fmaf(fminf(fmaxf(ch[ci][r] * 1e30f, 0.f), 1.f), 1.f - slope, slope);
            }
        } else {
            constexpr int nbp = (MODE & 2) ? 0 : GI - 16;
#pragma unroll
            for (int ci = 0; ci < 2; ++ci)
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int h = 0; h < 2; ++h) acc[2 * nbp + h] = mfma4(cur[ci * 2 + h][s], ch[ci][s], acc[2 * nbp + h]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) cur[i] = nxt[i];
        groups<GI + 1, MODE>(xin, acc, ch, cur, base, lane, slope, c, nc);
    }
}

template <int MODE>
__global__ void __launch_bounds__(256, 1) k(float* out, unsigned long long* cyc, int nc, float slope, const char* gbuf) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 49152 / 4; i += 256) ((float*)smem)[i] = 1e-3f * (1 + (i % 7));
    __syncthreads();
    R r; r.g = gbuf; r.smem = smem; r.cur = 0; r.next = 0; r.lane = lane; r.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f32x4 xin[32], acc[32];
    for (int i = 0; i < 32; ++i) { xin[i] = f32x4{1.f + lane + i, 2.f, 3.f, 4.f} * 1e-3f; acc[i] = f32x4{0, 0, 0, 0}; }
    f32x4 cur[4];
    load_group<0, 0>(cur, r);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int c = 0; c < nc; ++c) {
        f32x4 ch[2] = {f32x4{0.1f, 0.2f, 0.3f, 0.4f}, f32x4{0.1f, 0.2f, 0.3f, 0.4f}};
        if (MODE & 32) { ch[0] = *(const f32x4*)(smem + 60000 + (c & 31) * 32 + (lane >> 4) * 16); ch[1] = *(const f32x4*)(smem + 61024 + (c & 31) * 32 + (lane >> 4) * 16); }
        groups<0, MODE>(xin, acc, ch, cur, r, lane, slope, c, nc);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 32; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, float* out, unsigned long long* cyc) {
    const int nc = 320, grid = 256;
    static char* gbuf = nullptr;
    if (!gbuf) { hipMalloc(&gbuf, 664 * 16384); hipMemset(gbuf, 0, 664 * 16384); }
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160000);
    hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(256), 160000, 0, out, cyc, nc, 0.01f, gbuf);
    hipDeviceSynchronize();
    unsigned long long h[256];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double mean = 0;
    for (int i = 0; i < grid; ++i) mean += h[i];
    mean /= grid;
    printf("%-44s %9.1f cycles / chunk (ideal 16384)  %.3f cyc/MFMA\n", name, mean / nc, mean / nc / 512);
}

int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    run<0>("full chunk (epilogue, 32 acc tiles)", out, cyc);
    run<1>("no epilogue", out, cyc);
    run<2>("epilogue, part B into 2 accumulators only", out, cyc);
    run<3>("no epilogue, 2 accumulators", out, cyc);
    run<16>("full + rotating buffers", out, cyc);
    run<16 + 4>("full + rotating + barrier/slot", out, cyc);
    run<16 + 8>("full + rotating + DMA/slot", out, cyc);
    run<16 + 4 + 8>("full + rotating + barrier + DMA", out, cyc);
    run<16 + 4 + 8 + 32>("full + rot + bar + DMA + bias/mask LDS", out, cyc);
    run<16 + 8 + 64>("full + rotating + DMA(saddr)/slot", out, cyc);
    run<16 + 4 + 8 + 64>("full + rotating + barrier + DMA(saddr)", out, cyc);
    return 0;
}
