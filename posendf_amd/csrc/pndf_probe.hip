// Memory-subsystem probe of the box the engine runs on (measurement aid behind bench.py's `box` block):
// the fused kernels stream their 11 MB of weights through L2 -> LDS every step, so two boxes can only be compared with the
// latencies of the levels that stream crosses beside the kernel time.
//   dependent-load latency (one lane, one 128-byte line per hop, full-period LCG walk) with the walked footprint
//     1 MiB   -> resident in the XCD's 4 MB L2          (what 31 of 32 compute units of an XCD see of the weight stream)
//     64 MiB  -> resident in the 256 MB Infinity Cache  (what the first one sees: an L2 miss served over the fabric)
//     1 GiB   -> HBM (incl. the translation misses of a random walk)
//   streaming read bandwidth of a 1 GiB buffer (all compute units, 16 B per lane)
// Everything is allocated, measured and freed inside the call; it synchronises the device.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "../../include/posendf_amd_debug.h"
#include "pndf_experiment.h"
#include "pndf_host.h"

namespace {

constexpr int LINE_WORDS = 32;                       // one hop per 128-byte line

// next(i) = (a i + c) mod n, n a power of two, a = 1 (mod 4), c odd: one cycle through all n lines
__global__ void probe_fill(uint32_t* buf, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) buf[(size_t)i * LINE_WORDS] = (i * 1664525u + 1013904223u) & (n - 1);
}

// pseudo-random words (the weights are random bits to the memory path: operand toggling costs energy)
__global__ void probe_fill_random(uint32_t* buf, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)i * 2654435761u + 12345u;
        x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
        buf[i] = x;
    }
}

__global__ void probe_chase(const uint32_t* buf, int hops, unsigned long long* ticks, uint32_t* sink) {
    if (threadIdx.x != 0) return;
    uint32_t i = 0;
    for (int k = 0; k < 64; ++k) {                   // settle (clock ramp, first translations)
        i = buf[(size_t)i * LINE_WORDS];
        asm volatile("" : "+v"(i));
    }
    const unsigned long long t0 = wall_clock64();
    for (int k = 0; k < hops; ++k) {
        i = buf[(size_t)i * LINE_WORDS];
        asm volatile("" : "+v"(i));                  // a dependent chain the compiler cannot collapse
    }
    const unsigned long long t1 = wall_clock64();
    *ticks = t1 - t0;
    *sink = i;
}

typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) probe_stream(const f4* src, size_t n, float* sink) {
    f4 acc = f4{0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        acc += src[i];
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) *sink = acc.x;      // (never true: keeps the loads)
}

// The weight ring alone: every workgroup (one per CU, 4 waves) streams the same 11 MB through LDS exactly as the fused kernels
// do -- 16-KiB slots, each wave moves 4 KiB with four global_load_lds_dwordx4, four slots in flight, a counted wait and a barrier
// per slot -- with no arithmetic behind it.  What it delivers per CU is the ceiling the kernels' stream lives under on this box.
constexpr int PR_SLOT = 16 * 1024, PR_SLOTS = 5, PR_STREAM_SLOTS = 670;
__global__ void __launch_bounds__(256, 1) probe_ring(const char* stream, int passes, unsigned long long* ticks) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem + wave * 4096;
    uint32_t off = wave * 4096 + lane * 16;
    auto fetch = [&](uint32_t buf) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024\n\t"
                     "global_load_lds_dwordx4 %0, %1 offset:2048\n\tglobal_load_lds_dwordx4 %0, %1 offset:3072"
                     : : "v"(off), "s"(stream), "s"(lds0 + buf * PR_SLOT) : "memory", "m0");
    };
    const unsigned long long t0 = wall_clock64();
    for (int p = 0; p < passes; ++p) {
        off = wave * 4096 + lane * 16;
        uint32_t buf = 0;
        for (int b = 0; b < PR_SLOTS - 1; ++b) {          // four slots in flight
            fetch(buf);
            off += PR_SLOT;
            buf = (buf + 1 == PR_SLOTS) ? 0 : buf + 1;
        }
        for (int s = 0; s < PR_STREAM_SLOTS; ++s) {
            asm volatile("s_waitcnt vmcnt(12)" ::: "memory");      // the oldest slot has landed (this wave's share)
            __builtin_amdgcn_s_barrier();                          // ... and every wave's: its buffer may be refilled
            if (s + PR_SLOTS - 1 < PR_STREAM_SLOTS) fetch(buf);
            else asm volatile("s_nop 0" ::: "memory");
            off += PR_SLOT;
            buf = (buf + 1 == PR_SLOTS) ? 0 : buf + 1;
            if (s + PR_SLOTS - 1 >= PR_STREAM_SLOTS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // tail: nothing new behind it
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    const unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

}  // namespace

// out[0..2] = dependent-load latency in ns at the three footprints, out[3] = streaming read GB/s, out[4] = wall-clock
// counter rate in MHz, out[5] = hops timed per footprint; with n_out >= 8 also out[6], out[7]: the weight ring alone (below).
// Returns 0 or a negative pndf_status.
extern "C" int pndf_debug_mem_probe(int device, double* out, int n_out) {
    if (!out || n_out < 6) return -1;
    DeviceGuard guard(device);
    if (!guard.ok) return -3;
    int rate_khz = 0;
    if (hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, device) != hipSuccess || rate_khz <= 0) rate_khz = 100000;
    const size_t bytes = (size_t)1 << 30;
    uint32_t* buf = nullptr;
    unsigned long long* ticks = nullptr;
    uint32_t* sink = nullptr;
    if (hipMalloc((void**)&buf, bytes) != hipSuccess) return -3;
    int rc = 0;
    if (hipMalloc((void**)&ticks, 64) != hipSuccess || hipMalloc((void**)&sink, 64) != hipSuccess) rc = -3;
    const int hops = 4096;
    const size_t foot[3] = {(size_t)1 << 20, (size_t)64 << 20, bytes};
    for (int f = 0; f < 3 && rc == 0; ++f) {
        const uint32_t n = (uint32_t)(foot[f] / (LINE_WORDS * 4));
        hipLaunchKernelGGL(probe_fill, dim3((n + 255) / 256), dim3(256), 0, 0, buf, n);
        // bring the footprint into the level it fits in: two streaming passes over it (the 1 GiB one fits in none)
        for (int pass = 0; pass < 2; ++pass)
            hipLaunchKernelGGL(probe_stream, dim3(1024), dim3(256), 0, 0, (const f4*)buf, foot[f] / 16, (float*)sink);
        hipLaunchKernelGGL(probe_chase, dim3(1), dim3(64), 0, 0, buf, hops, ticks, sink);
        if (hipGetLastError() != hipSuccess) { rc = -3; break; }
        unsigned long long t = 0;
        if (hipMemcpy(&t, ticks, sizeof(t), hipMemcpyDeviceToHost) != hipSuccess) { rc = -3; break; }
        out[f] = (double)t / hops * 1e6 / rate_khz;      // ticks per hop -> ns
    }
    if (rc == 0) {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(probe_stream, dim3(2048), dim3(256), 0, 0, (const f4*)buf, bytes / 16, (float*)sink);
        (void)hipEventRecord(e0, 0);
        for (int r = 0; r < 3; ++r)
            hipLaunchKernelGGL(probe_stream, dim3(2048), dim3(256), 0, 0, (const f4*)buf, bytes / 16, (float*)sink);
        (void)hipEventRecord(e1, 0);
        float ms = 0.f;
        if (hipGetLastError() != hipSuccess || hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess || ms <= 0.f) rc = -3;
        else out[3] = 3.0 * (double)bytes / (ms * 1e-3) / 1e9;
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
    }
    out[4] = rate_khz / 1e3;
    out[5] = hops;
    if (rc == 0 && n_out >= 8) {
        // out[6] = GB/s that ONE compute unit pulls through its LDS ring while all of them do (the fused f16x3 kernel needs ~51),
        // out[7] = ns per 16-KiB slot.  Mean over the workgroups, one per CU.
        hipDeviceProp_t prop;
        int cus = 256;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
        unsigned long long* tk = nullptr;
        const int passes = 8;
        if (hipMalloc((void**)&tk, (size_t)cus * sizeof(unsigned long long)) != hipSuccess) rc = -3;
        if (rc == 0 && hipFuncSetAttribute((const void*)probe_ring, hipFuncAttributeMaxDynamicSharedMemorySize, PR_SLOTS * PR_SLOT) != hipSuccess) rc = -3;
        if (rc == 0) {
            hipLaunchKernelGGL(probe_ring, dim3(cus), dim3(256), PR_SLOTS * PR_SLOT, 0, (const char*)buf, 1, tk);      // warm
            hipLaunchKernelGGL(probe_ring, dim3(cus), dim3(256), PR_SLOTS * PR_SLOT, 0, (const char*)buf, passes, tk);
            std::vector<unsigned long long> h(cus);
            if (hipGetLastError() != hipSuccess) rc = -3;
            else if (hipMemcpy(h.data(), tk, (size_t)cus * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) rc = -3;
            else {
                double sum = 0;
                for (int i = 0; i < cus; ++i) sum += (double)h[i];
                const double sec = sum / cus / (rate_khz * 1e3);
                out[6] = (double)passes * PR_STREAM_SLOTS * PR_SLOT / sec / 1e9;
                out[7] = sec / ((double)passes * PR_STREAM_SLOTS) * 1e9;
            }
        }
        if (tk) (void)hipFree(tk);
    }
    if (sink) (void)hipFree(sink);
    if (ticks) (void)hipFree(ticks);
    (void)hipFree(buf);
    return rc;
}

// The ring-only stream of pndf_debug_mem_probe as a load of its own: `passes` walks of an 11 MB buffer by one workgroup per
// compute unit, synchronous.  tools/power_window.py --ring-only reads the package power while it runs: what delivering the weight
// stream into LDS costs in energy with nothing else going on.  Returns seconds per pass in *sec_per_pass (may be NULL).
extern "C" int pndf_debug_ring_stream(int device, int passes, double* sec_per_pass) {
    if (passes <= 0) return -1;
    DeviceGuard guard(device);
    if (!guard.ok) return -3;
    int rate_khz = 0;
    if (hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, device) != hipSuccess || rate_khz <= 0) rate_khz = 100000;
    hipDeviceProp_t prop;
    int cus = 256;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
    // allocated and freed per call, on `device` (function-local statics were tied to the first device passed in, never
    // freed and not thread-safe); every launch is checked -- a failed launch must not come back as rc 0 with garbage timings
    char* buf = nullptr;
    unsigned long long* tk = nullptr;
    const size_t bytes = (size_t)(PR_STREAM_SLOTS + PR_SLOTS) * PR_SLOT;
    if (cus > 1024) cus = 1024;
    int rc = 0;
    if (hipMalloc((void**)&buf, bytes) != hipSuccess || hipMalloc((void**)&tk, 1024 * sizeof(unsigned long long)) != hipSuccess) rc = -3;
    if (rc == 0) {
        hipLaunchKernelGGL(probe_fill_random, dim3(1024), dim3(256), 0, 0, (uint32_t*)buf, bytes / 4);
        if (hipGetLastError() != hipSuccess) rc = -3;
    }
    if (rc == 0 && hipFuncSetAttribute((const void*)probe_ring, hipFuncAttributeMaxDynamicSharedMemorySize, PR_SLOTS * PR_SLOT) != hipSuccess) rc = -3;
    if (rc == 0) {
        hipLaunchKernelGGL(probe_ring, dim3(cus), dim3(256), PR_SLOTS * PR_SLOT, 0, (const char*)buf, passes, tk);
        if (hipGetLastError() != hipSuccess) rc = -3;
    }
    unsigned long long t = 0;
    if (rc == 0 && hipMemcpy(&t, tk, sizeof(t), hipMemcpyDeviceToHost) != hipSuccess) rc = -3;      // (synchronises: the kernel's own errors surface here)
    if (rc == 0 && sec_per_pass) *sec_per_pass = (double)t / (rate_khz * 1e3) / passes;
    if (tk) (void)hipFree(tk);
    if (buf) (void)hipFree(buf);
    return rc;
}

PNDF_EXPORT_EXPERIMENT_WORD(probe)
