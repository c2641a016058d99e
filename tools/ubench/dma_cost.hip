// What does one 1-KiB LDS-DMA piece cost a wave that is otherwise issuing MFMAs back to back, and does the
// addressing form matter?  One wave per SIMD (4 waves), 24 MFMAs + NP pieces per iteration, pieces spread.
//   V=0 none   V=1 global_load_lds (64-bit VGPR address), M0 saved/restored   V=2 same, M0 set only
//   V=3 global_load_lds, SGPR base + 32-bit VGPR offset   V=4 buffer_load ... offen lds (32-bit VGPR offset)
//   V=5 buffer_load ... off lds with ADD_TID_ENABLE (stride 16, no VGPR)   V=6 plain global_load_dwordx4 to VGPRs
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x4 mf(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

template <int V>
__device__ __forceinline__ void piece(const char* gaddr, const char* gbase, unsigned goff, unsigned lds, i32x4 rsrc, unsigned soff, f32x4& sink) {
    if (V == 1) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gaddr), "s"(lds) : "memory");
    } else if (V == 2) {
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gaddr), "s"(lds) : "memory");
    } else if (V == 3) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(goff), "s"(gbase), "s"(lds) : "memory");
    } else if (V == 4) {
        asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" : : "v"(goff), "s"(rsrc), "s"(soff), "s"(lds) : "memory");
    } else if (V == 5) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 off, %0, %1 lds" : : "s"(rsrc), "s"(soff), "s"(lds) : "memory");
    } else if (V == 6) {
        f32x4 t;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(t) : "v"(gaddr) : "memory");
        sink = t;   // consumed after the loop only
    }
}

template <int V, int NP>
__global__ void __launch_bounds__(256, 1) k(float* out, unsigned long long* cyc, int iters, const char* gbuf, int* bad) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(0.01f * (lane + j)); b[j] = (_Float16)(0.02f * j); }
    f32x4 acc[8], sink = f32x4{0, 0, 0, 0};
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + wave * 16384;
    i32x4 rsrc;
    const unsigned long long ga = (unsigned long long)gbuf;
    rsrc[0] = (int)(ga & 0xffffffffu);
    rsrc[1] = (int)((ga >> 32) & 0xffffu) | (V == 5 ? (16 << 16) : 0);
    rsrc[2] = (int)0x7fffffff;
    rsrc[3] = (V == 5) ? 0x00800000 : 0x00020000;   // ADD_TID_ENABLE: DATA_FORMAT bits become stride[17:14], keep them 0
    rsrc[0] = __builtin_amdgcn_readfirstlane(rsrc[0]); rsrc[1] = __builtin_amdgcn_readfirstlane(rsrc[1]);
    rsrc[2] = __builtin_amdgcn_readfirstlane(rsrc[2]); rsrc[3] = __builtin_amdgcn_readfirstlane(rsrc[3]);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned pos = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 24; ++m) {
            acc[m % 8] = mf(a, b, acc[m % 8]);
            if (NP > 0 && (m % (24 / (NP > 0 ? NP : 1))) == 0 && m / (24 / (NP > 0 ? NP : 1)) < NP) {
                const int j = m / (24 / (NP > 0 ? NP : 1));
                const unsigned off = (pos + wave * 4u + j) * 1024u;              // byte offset of this piece in gbuf
                __builtin_amdgcn_sched_barrier(0);
                piece<V>(gbuf + off + lane * 16, gbuf, off + lane * 16, lds0 + ((pos + j) & 15) * 1024, rsrc, V == 4 ? 0u : off, sink);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        pos = (pos + 16) & 4095;
        if (V != 0 && V != 6) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    // data check of the last iteration's pieces (LDS variants)
    if (V >= 1 && V <= 5 && blockIdx.x == 0) {
        const unsigned lastpos = (pos + 4096 - 16) & 4095;
        for (int j = 0; j < NP; ++j) {
            const unsigned off = (lastpos + wave * 4u + j) * 1024u + lane * 16;
            const int* g = (const int*)(gbuf + off);
            const int* l = (const int*)(smem + wave * 16384 + ((lastpos + j) & 15) * 1024 + lane * 16);
            for (int q = 0; q < 4; ++q) if (g[q] != l[q]) atomicAdd(bad, 1);
        }
    }
    float s = sink[0];
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int V, int NP>
void run(const char* name, float* out, unsigned long long* cyc, const char* gbuf, int* bad) {
    const int iters = 2000, grid = 256;
    (void)hipMemset(bad, 0, 4);
    (void)hipFuncSetAttribute((const void*)k<V, NP>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipLaunchKernelGGL((k<V, NP>), dim3(grid), dim3(256), 65536, 0, out, cyc, iters, gbuf, bad);
    hipError_t e = hipDeviceSynchronize();
    static unsigned long long h[256 * 4];
    (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    int hb = 0; (void)hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
    double m = 0;
    for (int i = 0; i < grid * 4; ++i) m += h[i];
    m /= grid * 4;
    printf("%-58s NP=%d %7.1f cycles/iter (24 MFMAs; floor 384)  mismatches %d  %s\n", name, NP, m / iters, hb, e == hipSuccess ? "" : hipGetErrorString(e)); fflush(stdout);
}

int main(int argc, char** argv) {
    const int sel = argc > 1 ? atoi(argv[1]) : -1;
    float* out; unsigned long long* cyc; char* gbuf; int* bad;
    (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 256 * 4 * 8); (void)hipMalloc(&bad, 4);
    const size_t gb = 4352 * 1024;
    (void)hipMalloc(&gbuf, gb);
    int* hbuf = (int*)malloc(gb);
    for (size_t i = 0; i < gb / 4; ++i) hbuf[i] = (int)(i * 2654435761u);
    (void)hipMemcpy(gbuf, hbuf, gb, hipMemcpyHostToDevice);
    if (sel < 0 || sel == 0) run<0, 0>("no DMA", out, cyc, gbuf, bad);
    if (sel < 0 || sel == 1) run<1, 4>("global_load_lds vaddr64, M0 save/restore", out, cyc, gbuf, bad);
    if (sel < 0 || sel == 2) run<2, 4>("global_load_lds vaddr64, M0 set", out, cyc, gbuf, bad);
    if (sel < 0 || sel == 3) run<3, 4>("global_load_lds saddr + voffset32", out, cyc, gbuf, bad);
    if (sel < 0 || sel == 4) run<4, 4>("buffer_load offen lds", out, cyc, gbuf, bad);
    if (sel < 0 || sel == 5) run<5, 4>("buffer_load off lds, ADD_TID_ENABLE", out, cyc, gbuf, bad);
    if (sel < 0 || sel == 6) run<6, 4>("global_load_dwordx4 to VGPR", out, cyc, gbuf, bad);
    if (sel < 0 || sel == 7) run<2, 2>("global_load_lds vaddr64, M0 set", out, cyc, gbuf, bad);
    if (sel < 0 || sel == 8) run<2, 8>("global_load_lds vaddr64, M0 set", out, cyc, gbuf, bad);
    if (sel < 0 || sel == 9) run<5, 8>("buffer_load off lds, ADD_TID_ENABLE", out, cyc, gbuf, bad);
    return 0;
}
