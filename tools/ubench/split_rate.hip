// Feasibility micro-benchmark for a split-precision (fp16 hi/lo, 3 MFMAs per product block, fp32 accumulate)
// version of the (lin2,lin3) chunk: v_mfma_f32_16x16x32_f16 fed from LDS (2 x ds_read_b128 per 3 MFMAs).
// Per chunk: part A 16 k-blocks x 2 tiles x 3 = 96 MFMAs, part B 32 tiles x 3 = 96 MFMAs, 128 KiB of LDS reads.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x4 mf(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

// a "group" = 4 tile-pairs (hi, lo) = 8 KiB of LDS = 12 MFMAs
struct R { const char* g; char* smem; int cur; int next; int wave; int lane; const char* src; unsigned dst; int n; };

__device__ __forceinline__ void piece(R& r) {
    if (r.n < 4) {
        unsigned keep;
        const char* src = r.src + r.n * 1024;
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(r.dst + r.n * 1024) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        r.n++;
    }
}

// T0 = pair index (8 pairs = 16 tiles per slot)
template <int TP, int MODE>
__device__ __forceinline__ void load_pair(f16x8& h, f16x8& l, R& r);

template <int T0, int MODE>
__device__ __forceinline__ void load_group(f16x8 (&h)[4], f16x8 (&l)[4], R& r) {
    load_pair<T0, MODE>(h[0], l[0], r);
    load_pair<T0 + 1, MODE>(h[1], l[1], r);
    load_pair<T0 + 2, MODE>(h[2], l[2], r);
    load_pair<T0 + 3, MODE>(h[3], l[3], r);
}

template <int TP, int MODE>
__device__ __forceinline__ void load_pair(f16x8& hh, f16x8& ll, R& r) {
    f16x8 h[1], l[1];
    {
        constexpr int i = 0;
        const int t = (2 * TP) % 16;
        if ((MODE & 16) && t == 0) r.cur = (r.cur == 4) ? 0 : r.cur + 1;
        if (t == 8) {
            if ((MODE & 8) && !(MODE & 128)) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            if ((MODE & 4) && (!(MODE & 64) || (r.next & 1))) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
            if (MODE & 8) {
                r.src = r.g + (size_t)r.next * 16384 + r.wave * 4096 + r.lane * 16;
                r.dst = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(r.smem) + ((r.cur == 0) ? 4 : r.cur - 1) * 16384 + r.wave * 4096;
                r.n = 0;
                r.next = (r.next + 1 == 664) ? 0 : r.next + 1;
            }
        }
        h[i] = *(const f16x8*)(r.smem + r.cur * 16384 + t * 1024 + r.lane * 16);
        l[i] = *(const f16x8*)(r.smem + r.cur * 16384 + (t + 1) * 1024 + r.lane * 16);
    }
    hh = h[0]; ll = l[0];
}

template <int GI, int MODE>
__device__ __forceinline__ void groups(const f16x8 (&xh)[16], const f16x8 (&xl)[16], f32x4 (&acc)[32], f32x4 (&ch)[2],
                                       f16x8 (&chh)[1], f16x8 (&chl)[1], f16x8 (&ch_)[4], f16x8 (&cl_)[4],
                                       R& smem, int lane, int c, int nc) {
    if constexpr (GI < 16) {
        f16x8 nh[4], nl[4];
        if (!(MODE & 256)) { if (GI + 1 < 16 || c + 1 < nc) load_group<((GI + 1) * 4) % 64, MODE>(nh, nl, smem); }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (GI < 8) {           // part A: 2 k-blocks x 2 tiles per group
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                for (int ci = 0; ci < 2; ++ci) {
                    ch[ci] = mf(ch_[k2 * 2 + ci], xh[2 * GI + k2], ch[ci]);
                    ch[ci] = mf(ch_[k2 * 2 + ci], xl[2 * GI + k2], ch[ci]);
                    ch[ci] = mf(cl_[k2 * 2 + ci], xh[2 * GI + k2], ch[ci]);
                    if (MODE & 256) { __builtin_amdgcn_sched_barrier(0); if (k2 * 2 + ci == 0) load_pair<((GI + 1) * 4) % 64, MODE>(nh[0], nl[0], smem); if (k2 * 2 + ci == 1) load_pair<((GI + 1) * 4) % 64 + 1, MODE>(nh[1], nl[1], smem); if (k2 * 2 + ci == 2) load_pair<((GI + 1) * 4) % 64 + 2, MODE>(nh[2], nl[2], smem); if (k2 * 2 + ci == 3) load_pair<((GI + 1) * 4) % 64 + 3, MODE>(nh[3], nl[3], smem); __builtin_amdgcn_sched_barrier(0); }
                    if ((MODE & 8) && (!(MODE & 32) || ci == 0)) piece(smem);
                }
            if constexpr (GI == 7 && (MODE & 1) == 0) {
                // epilogue: activation + split into fp16 hi / lo, packed as the next B operand
#pragma unroll
                for (int ci = 0; ci < 2; ++ci)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float z = ch[ci][r];
                        z = z * fmaf(fminf(fmaxf(z * 1e30f, 0.f), 1.f), 0.99f, 0.01f);
                        const _Float16 hi = (_Float16)z;
                        const _Float16 lo = (_Float16)(z - (float)hi);
                        chh[0][ci * 4 + r] = hi;
                        chl[0][ci * 4 + r] = lo;
                    }
            }
        } else {                          // part B: 4 output tiles per group
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                constexpr int dummy = 0; (void)dummy;
                const int nb = (MODE & 2) ? t : (GI - 8) * 4 + t;
                acc[nb] = mf(ch_[t], chh[0], acc[nb]);
                acc[nb] = mf(ch_[t], chl[0], acc[nb]);
                acc[nb] = mf(cl_[t], chh[0], acc[nb]);
                if (MODE & 256) { __builtin_amdgcn_sched_barrier(0); if (t == 0) load_pair<((GI + 1) * 4) % 64, MODE>(nh[0], nl[0], smem); if (t == 1) load_pair<((GI + 1) * 4) % 64 + 1, MODE>(nh[1], nl[1], smem); if (t == 2) load_pair<((GI + 1) * 4) % 64 + 2, MODE>(nh[2], nl[2], smem); if (t == 3) load_pair<((GI + 1) * 4) % 64 + 3, MODE>(nh[3], nl[3], smem); __builtin_amdgcn_sched_barrier(0); }
                if ((MODE & 8) && (!(MODE & 32) || (t & 1) == 0)) piece(smem);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) { ch_[i] = nh[i]; cl_[i] = nl[i]; }
        groups<GI + 1, MODE>(xh, xl, acc, ch, chh, chl, ch_, cl_, smem, lane, c, nc);
    }
}

template <int MODE>
__global__ void __launch_bounds__(256, 1) k(float* out, unsigned long long* cyc, int nc, const char* gbuf) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 81920 / 2; i += 256) ((_Float16*)smem)[i] = (_Float16)(1e-2f * (1 + (i % 7)));
    __syncthreads();
    R r; r.g = gbuf; r.smem = smem; r.cur = 0; r.next = 0; r.lane = lane; r.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); r.n = 4; r.src = gbuf; r.dst = 0;
    f16x8 xh[16], xl[16];
    f32x4 acc[32];
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 8; ++j) { xh[i][j] = (_Float16)(0.01f * (lane % 5 + i + j)); xl[i][j] = (_Float16)(1e-5f * (i + j)); }
    for (int i = 0; i < 32; ++i) acc[i] = f32x4{0, 0, 0, 0};
    f16x8 ch_[4], cl_[4], chh[1], chl[1];
    load_group<0, 0>(ch_, cl_, r);
    for (int j = 0; j < 8; ++j) { chh[0][j] = (_Float16)0.1f; chl[0][j] = (_Float16)1e-4f; }
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int c = 0; c < nc; ++c) {
        f32x4 ch[2] = {f32x4{0.1f, 0.2f, 0.3f, 0.4f}, f32x4{0.1f, 0.2f, 0.3f, 0.4f}};
        groups<0, MODE>(xh, xl, acc, ch, chh, chl, ch_, cl_, r, lane, c, nc);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 32; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s + (float)chh[0][0];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, float* out, unsigned long long* cyc) {
    const int nc = 640, grid = 256;
    static char* gbuf = nullptr;
    if (!gbuf) { hipMalloc(&gbuf, 664 * 16384); hipMemset(gbuf, 0, 664 * 16384); }
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160000);
    hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(256), 160000, 0, out, cyc, nc, gbuf);
    hipDeviceSynchronize();
    unsigned long long h[256];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double mean = 0;
    for (int i = 0; i < grid; ++i) mean += h[i];
    mean /= grid;
    printf("%-40s %8.1f cycles / chunk  (192 MFMAs: %.2f cyc/MFMA; fp32 kernel chunk = 18400)\n", name, mean / nc, mean / nc / 192);
}

int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    run<0>("split chunk, epilogue, 32 acc tiles", out, cyc);
    run<1>("split chunk, no epilogue", out, cyc);
    run<3>("no epilogue, 4 accumulators", out, cyc);
    run<1 + 16>("no epi + rotating buffers", out, cyc);
    run<1 + 16 + 4>("no epi + rot + barrier/slot", out, cyc);
    run<1 + 16 + 8>("no epi + rot + DMA/slot", out, cyc);
    run<1 + 16 + 4 + 8>("no epi + rot + barrier + DMA", out, cyc);
    run<16 + 4 + 8>("epilogue + rot + barrier + DMA", out, cyc);
    run<16 + 4 + 8 + 32>("  + spread pieces", out, cyc);
    run<16 + 4 + 8 + 64>("  + barrier every 2nd slot", out, cyc);
    run<16 + 4 + 8 + 128>("  + no vmcnt wait", out, cyc);
    run<16 + 4 + 8 + 32 + 64>("  + spread + barrier/2", out, cyc);
    run<16 + 4 + 8 + 32 + 64 + 128>("  + spread + barrier/2 + no vmcnt", out, cyc);
    run<16 + 4 + 8 + 256>("  + interleaved reads", out, cyc);
    run<16 + 4 + 8 + 256 + 32>("  + interleaved reads + spread", out, cyc);
    run<16 + 4 + 256>("epilogue + rot + barrier, interleaved", out, cyc);
    run<16 + 256>("epilogue + rot, interleaved", out, cyc);
    run<16>("epilogue + rot only", out, cyc);
    run<16 + 8>("epilogue + rot + DMA", out, cyc);
    run<16 + 8 + 32>("epilogue + rot + DMA spread", out, cyc);
    run<16 + 4>("epilogue + rot + barrier", out, cyc);
    return 0;
}
