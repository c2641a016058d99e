// Feasibility micro-benchmark: "pair32" version of the split-precision (lin2,lin3) chunk.
// Workgroup = 4 waves (one per SIMD, 512 registers); waves (0,1) and (2,3) are pairs that share 32 poses.
// Both waves of a pair run the same code: part A with the contraction split between them (each holds half of x2's rows
// as B operands for 32 poses), partial chunk sums exchanged through LDS, part B with the output rows split (each holds
// half of the lin3 accumulators).  Each wave therefore reads only half of the weight tiles from LDS (its 4 tile pairs
// of every 16-KiB slot) and issues 6 MFMAs per tile pair: LDS read traffic per pose is half of the 16-pose kernel's.
// Per chunk and wave: 192 MFMAs (floor 3072 cycles), 64 KiB of LDS tile reads, 4+4 KiB exchange.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x4 mf(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

constexpr int SLOTS = 5, SLOT_B = 16384, NSLOT_G = 664;

struct R { const char* g; char* smem; int cur; int next; int wave; int lane; int half; const char* base; const char* src; unsigned dst; };

__device__ __forceinline__ void dma_piece(const char* src, unsigned dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
}

template <int MODE>
__device__ __forceinline__ void next_slot(R& r) {
    if (MODE & 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    if (MODE & 1) { asm volatile("s_barrier" ::: "memory"); }
    const int prev = r.cur;
    r.cur = (r.cur == SLOTS - 1) ? 0 : r.cur + 1;
    r.base = r.smem + r.cur * SLOT_B + r.half * 8192 + r.lane * 16;
    const int tgt = (prev == 0) ? SLOTS - 1 : prev - 1;
    r.src = r.g + (size_t)r.next * SLOT_B + r.wave * 4096 + r.lane * 16;
    r.dst = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(r.smem) + tgt * SLOT_B + r.wave * 4096;
    r.next = (r.next + 1 == NSLOT_G) ? 0 : r.next + 1;
}

#define SB() __builtin_amdgcn_sched_barrier(0)

template <int MODE>
__global__ void __launch_bounds__(256, 1) k(float* out, unsigned long long* cyc, int nc, const char* gbuf) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < SLOTS * SLOT_B / 2; i += 256) ((_Float16*)smem)[i] = (_Float16)(1e-2f * (1 + (i % 7)));
    char* xch = smem + SLOTS * SLOT_B;                        // [4 waves][4 KiB] partial chunk sums
    for (int i = threadIdx.x; i < 4 * 4096 / 4; i += 256) ((float*)xch)[i] = 0.01f;
    __syncthreads();
    R r; r.g = gbuf; r.smem = smem; r.cur = 0; r.next = 0; r.lane = lane; r.wave = wave; r.half = wave & 1;
    r.base = smem + r.half * 8192 + lane * 16; r.src = gbuf; r.dst = 0;
    char* mine = xch + wave * 4096 + lane * 16;
    const char* theirs = xch + (wave ^ 1) * 4096 + lane * 16;
    f16x8 th[4], tl[4], nh[4], nl[4];
    for (int i = 0; i < 4; ++i) { th[i] = *(const f16x8*)(r.base + 2 * i * 1024); tl[i] = *(const f16x8*)(r.base + (2 * i + 1) * 1024); nh[i] = th[i]; nl[i] = tl[i]; }
    // state: x2 half (8 k-blocks x 2 pose halves, hi/lo) and 16 x 2 accumulator tiles
    f16x8 xh[8][2], xl[8][2];
    for (int i = 0; i < 8; ++i) for (int p = 0; p < 2; ++p)
        for (int j = 0; j < 8; ++j) { xh[i][p][j] = (_Float16)(0.01f * (lane % 5 + i + j + p)); xl[i][p][j] = (_Float16)(1e-5f * (i + j + p)); }
    f32x4 acc[16][2];
    for (int i = 0; i < 16; ++i) for (int p = 0; p < 2; ++p) acc[i][p] = f32x4{0, 0, 0, 0};
    f16x8 chh[2], chl[2];
    for (int p = 0; p < 2; ++p) for (int j = 0; j < 8; ++j) { chh[p][j] = (_Float16)0.1f; chl[p][j] = (_Float16)1e-4f; }
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int c = 0; c < nc; ++c) {
        f32x4 ch[2][2][3];                       // [row tile][pose half][partial]
#pragma unroll
        for (int ci = 0; ci < 2; ++ci)
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int q = 0; q < 3; ++q) ch[ci][p][q] = f32x4{0.1f, 0.2f, 0.3f, 0.4f};
        // ---- part A of chunk c+1: 4 slots, per slot 2 k-blocks x 2 row tiles
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            next_slot<MODE>(r);
            SB();
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                for (int ci = 0; ci < 2; ++ci) {
                    const int i = k2 * 2 + ci;
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        ch[ci][p][0] = mf(th[i], xh[2 * g + k2][p], ch[ci][p][0]);
                        ch[ci][p][1] = mf(th[i], xl[2 * g + k2][p], ch[ci][p][1]);
                        ch[ci][p][2] = mf(tl[i], xh[2 * g + k2][p], ch[ci][p][2]);
                    }
                    SB();
                    nh[i] = *(const f16x8*)(r.base + (2 * i) * 1024);
                    nl[i] = *(const f16x8*)(r.base + (2 * i + 1) * 1024);
                    if (MODE & 2) dma_piece(r.src + i * 1024, r.dst + i * 1024);
                    SB();
                }
#pragma unroll
            for (int i = 0; i < 4; ++i) { th[i] = nh[i]; tl[i] = nl[i]; }
        }
        // partial sums of chunk c+1 to LDS
#pragma unroll
        for (int ci = 0; ci < 2; ++ci)
#pragma unroll
            for (int p = 0; p < 2; ++p)
                *(f32x4*)(mine + (ci * 2 + p) * 1024) = ch[ci][p][0] + ch[ci][p][1] + ch[ci][p][2];
        // ---- part B of chunk c: 4 slots, per slot 4 output tiles
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            next_slot<MODE>(r);
            SB();
#pragma unroll
            for (int t = 0; t < 4; ++t) {
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    acc[g * 4 + t][p] = mf(th[t], chh[p], acc[g * 4 + t][p]);
                    acc[g * 4 + t][p] = mf(th[t], chl[p], acc[g * 4 + t][p]);
                    acc[g * 4 + t][p] = mf(tl[t], chh[p], acc[g * 4 + t][p]);
                }
                SB();
                nh[t] = *(const f16x8*)(r.base + (2 * t) * 1024);
                nl[t] = *(const f16x8*)(r.base + (2 * t + 1) * 1024);
                if (MODE & 2) dma_piece(r.src + t * 1024, r.dst + t * 1024);
                SB();
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) { th[i] = nh[i]; tl[i] = nl[i]; }
        }
        // epilogue of chunk c+1: own + partner partial, activation, split -> B operand of the next part B
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int ci = 0; ci < 2; ++ci) {
                const f32x4 o = *(const f32x4*)(theirs + (ci * 2 + p) * 1024);
                const f32x4 m = ch[ci][p][0] + ch[ci][p][1] + ch[ci][p][2];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float z = m[q] + o[q];
                    z = z * fmaf(fminf(fmaxf(z * 1e30f, 0.f), 1.f), 0.99f, 0.01f);
                    const _Float16 hi = (_Float16)z;
                    const _Float16 lo = (_Float16)(z - (float)hi);
                    chh[p][ci * 4 + q] = hi;
                    chl[p][ci * 4 + q] = lo;
                }
            }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sum = (float)xh[0][0][0];
    for (int i = 0; i < 16; ++i) for (int p = 0; p < 2; ++p) sum += acc[i][p][0] + acc[i][p][1] + acc[i][p][2] + acc[i][p][3];
    out[blockIdx.x * 256 + threadIdx.x] = sum;
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int MODE>
void run(const char* name, float* out, unsigned long long* cyc) {
    const int nc = 640, grid = 256;
    static char* gbuf = nullptr;
    if (!gbuf) { (void)hipMalloc(&gbuf, NSLOT_G * SLOT_B); (void)hipMemset(gbuf, 0, NSLOT_G * SLOT_B); }
    (void)hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160000);
    hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(256), 160000, 0, out, cyc, nc, gbuf);
    (void)hipDeviceSynchronize();
    static unsigned long long h[256 * 4];
    (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0;
    for (int i = 0; i < grid * 4; ++i) m += h[i];
    m /= grid * 4;
    printf("%-36s %8.1f cycles / chunk of 64 poses per CU (floor 3072; 16-pose-wave kernel 6200)\n", name, m / nc);
}

int main() {
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 256 * 4 * 8);
    run<0>("pair32, no ring sync", out, cyc);
    run<1>("pair32 + barrier/slot", out, cyc);
    run<2>("pair32 + DMA", out, cyc);
    run<3>("pair32 + barrier + DMA", out, cyc);
    return 0;
}
