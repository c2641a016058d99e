// Feasibility micro-benchmark: wave-PAIR version of the split-precision (lin2,lin3) chunk.
// Workgroup = 8 waves = 2 per SIMD (256 registers each); waves w and w+4 share one SIMD and 16 poses:
//   role A (waves 0..3): holds x2 (hi/lo), computes the 32-row chunk of lin2 (96 MFMAs), activation + split, hands the
//                        chunk's B operand to its partner through LDS;
//   role B (waves 4..7): holds the lin3 accumulators (32 tiles), 96 MFMAs per chunk.
// Every 16-KiB ring slot holds 4 tile pairs for role A and 4 for role B; one workgroup barrier per slot; each wave
// DMAs 2 KiB of every slot.  Question: how close to the MFMA floor (192 MFMAs x 16 cycles = 3072 cycles per chunk
// and SIMD) do two interleaving waves get, compared with 6.2 k for the single-wave kernel?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x4 mf(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

constexpr int SLOTS = 5, SLOT_B = 16384, NSLOT_G = 664;

struct R { const char* g; char* smem; int cur; int next; int wave; int lane; int role; const char* base; };
#define LDP(i) do { if (MODE & 4) { __builtin_amdgcn_sched_barrier(0); nh[i] = *(const f16x8*)(r.base + (2 * (i)) * 1024); nl[i] = *(const f16x8*)(r.base + (2 * (i) + 1) * 1024); __builtin_amdgcn_sched_barrier(0); } } while (0)

__device__ __forceinline__ void dma_piece(const char* src, unsigned dst) {
    unsigned keep;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// advance to the next slot: wait for own DMA of it, barrier, read own group (4 pairs) of it; returns DMA params
template <int MODE>
__device__ __forceinline__ void next_group(f16x8 (&h)[4], f16x8 (&l)[4], R& r, const char*& src, unsigned& dst) {
    if (MODE & 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    if (MODE & 1) { asm volatile("s_barrier" ::: "memory"); }
    const int prev = r.cur;
    r.cur = (r.cur == SLOTS - 1) ? 0 : r.cur + 1;
    const char* base = r.smem + r.cur * SLOT_B + r.role * 8192 + r.lane * 16;
    r.base = base;
    if (!(MODE & 4)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            h[i] = *(const f16x8*)(base + (2 * i) * 1024);
            l[i] = *(const f16x8*)(base + (2 * i + 1) * 1024);
        }
    }
    // DMA of slot (s+4) goes to the buffer of slot s-1 = (prev - 1)
    const int tgt = (prev == 0) ? SLOTS - 1 : prev - 1;
    src = r.g + (size_t)r.next * SLOT_B + r.wave * 2048 + r.lane * 16;
    dst = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(r.smem) + tgt * SLOT_B + r.wave * 2048;
    r.next = (r.next + 1 == NSLOT_G) ? 0 : r.next + 1;
}

template <int MODE>
__global__ void __launch_bounds__(512) k(float* out, unsigned long long* cyc, int nc, const char* gbuf) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < SLOTS * SLOT_B / 2; i += 512) ((_Float16*)smem)[i] = (_Float16)(1e-2f * (1 + (i % 7)));
    char* hand = smem + SLOTS * SLOT_B;                       // [4 pairs][2 buffers][2 KiB]
    for (int i = threadIdx.x; i < 4 * 2 * 2048 / 2; i += 512) ((_Float16*)hand)[i] = (_Float16)0.01f;
    __syncthreads();
    R r; r.g = gbuf; r.smem = smem; r.cur = 0; r.next = 0; r.lane = lane; r.wave = wave; r.role = wave >> 2;
    char* myhand = hand + (wave & 3) * 4096 + lane * 16;
    f16x8 ch_[4], cl_[4], nh[4], nl[4];
    const char* src = gbuf; unsigned dst = 0;
    next_group<0>(ch_, cl_, r, src, dst);
    for (int i = 0; i < 4; ++i) { nh[i] = ch_[i]; nl[i] = cl_[i]; }
    float sum = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (r.role == 0) {
        f16x8 xh[16], xl[16];
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 8; ++j) { xh[i][j] = (_Float16)(0.01f * (lane % 5 + i + j)); xl[i][j] = (_Float16)(1e-5f * (i + j)); }
        for (int c = 0; c < nc; ++c) {
            f32x4 ch[2][3];
#pragma unroll
            for (int ci = 0; ci < 2; ++ci)
#pragma unroll
                for (int p = 0; p < 3; ++p) ch[ci][p] = f32x4{0.1f, 0.2f, 0.3f, 0.4f};
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                next_group<MODE>(nh, nl, r, src, dst);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                    for (int ci = 0; ci < 2; ++ci) {
                        ch[ci][0] = mf(ch_[k2 * 2 + ci], xh[2 * g + k2], ch[ci][0]);
                        ch[ci][1] = mf(ch_[k2 * 2 + ci], xl[2 * g + k2], ch[ci][1]);
                        ch[ci][2] = mf(cl_[k2 * 2 + ci], xh[2 * g + k2], ch[ci][2]);
                        LDP(k2 * 2 + ci);
                        if ((MODE & 2) && ci == 0) dma_piece(src + k2 * 1024, dst + k2 * 1024);
                    }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i) { ch_[i] = nh[i]; cl_[i] = nl[i]; }
            }
            f16x8 oh, ol;
#pragma unroll
            for (int ci = 0; ci < 2; ++ci)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float z = ch[ci][0][q] + ch[ci][1][q] + ch[ci][2][q];
                    z = z * fmaf(fminf(fmaxf(z * 1e30f, 0.f), 1.f), 0.99f, 0.01f);
                    const _Float16 hi = (_Float16)z;
                    const _Float16 lo = (_Float16)(z - (float)hi);
                    oh[ci * 4 + q] = hi;
                    ol[ci * 4 + q] = lo;
                }
            *(f16x8*)(myhand + (c & 1) * 2048) = oh;
            *(f16x8*)(myhand + (c & 1) * 2048 + 1024) = ol;
        }
        sum = (float)xh[0][0];
    } else {
        f32x4 acc[32];
        for (int i = 0; i < 32; ++i) acc[i] = f32x4{0, 0, 0, 0};
        f16x8 chh, chl;
        for (int j = 0; j < 8; ++j) { chh[j] = (_Float16)0.1f; chl[j] = (_Float16)1e-4f; }
        for (int c = 0; c < nc; ++c) {
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                next_group<MODE>(nh, nl, r, src, dst);
                f16x8 th, tl;
                if (g == 0) {            // partner's chunk operand (written >= one barrier ago)
                    th = *(const f16x8*)(myhand + ((c + 1) & 1) * 2048);
                    tl = *(const f16x8*)(myhand + ((c + 1) & 1) * 2048 + 1024);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    acc[g * 4 + t] = mf(ch_[t], chh, acc[g * 4 + t]);
                    acc[g * 4 + t] = mf(ch_[t], chl, acc[g * 4 + t]);
                    acc[g * 4 + t] = mf(cl_[t], chh, acc[g * 4 + t]);
                    LDP(t);
                    if ((MODE & 2) && (t & 1) == 0) dma_piece(src + (t >> 1) * 1024, dst + (t >> 1) * 1024);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (g == 0) { chh = th; chl = tl; }
#pragma unroll
                for (int i = 0; i < 4; ++i) { ch_[i] = nh[i]; cl_[i] = nl[i]; }
            }
        }
        for (int i = 0; i < 32; ++i) sum += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 512 + threadIdx.x] = sum;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MODE>
void run(const char* name, float* out, unsigned long long* cyc) {
    const int nc = 640, grid = 256;
    static char* gbuf = nullptr;
    if (!gbuf) { (void)hipMalloc(&gbuf, NSLOT_G * SLOT_B); (void)hipMemset(gbuf, 0, NSLOT_G * SLOT_B); }
    (void)hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160000);
    hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(512), 160000, 0, out, cyc, nc, gbuf);
    (void)hipDeviceSynchronize();
    static unsigned long long h[256 * 8];
    (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double ma = 0, mb = 0;
    for (int i = 0; i < grid; ++i) for (int w = 0; w < 8; ++w) (w < 4 ? ma : mb) += h[i * 8 + w];
    ma /= grid * 4; mb /= grid * 4;
    printf("%-36s role A %8.1f  role B %8.1f cycles / chunk  (floor 3072; single-wave kernel 6200)\n", name, ma / nc, mb / nc);
    printf("    per-wave means:");
    for (int w = 0; w < 8; ++w) { double m = 0; for (int i = 0; i < grid; ++i) m += h[i * 8 + w]; printf(" %7.1f", m / grid / nc); }
    printf("\n");
}

int main() {
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 256 * 8 * 8);
    run<0>("pair, no ring sync", out, cyc);
    run<1>("pair + barrier/slot", out, cyc);
    run<2>("pair + DMA", out, cyc);
    run<3>("pair + barrier + DMA", out, cyc);
    run<4>("pair, interleaved reads", out, cyc);
    run<5>("pair + barrier, interleaved", out, cyc);
    run<7>("pair + barrier + DMA, interleaved", out, cyc);
    return 0;
}
