// Probe of ds_read_b64_tr_b16 (gfx950): which (lane', element') of the per-lane 8-byte LDS reads ends up in (lane, element).
// LDS holds u16 value i at half-index i; lane l supplies byte address 8 l, i.e. it "owns" halfs 4l..4l+3.  The output
// value v at (lane, j) therefore came from lane' = v / 4, element' = v % 4.
// Build: hipcc --offload-arch=gfx950 -O2 tools/ubench/tr_probe.hip -o /tmp/tr_probe && /tmp/tr_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__global__ void probe(uint16_t* out, uint16_t* out8) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const uint32_t addr = (uint32_t)(size_t)(__attribute__((address_space(3))) uint16_t*)lds + threadIdx.x * 8;
    uint64_t r;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
    out[threadIdx.x * 4 + 0] = (uint16_t)(r & 0xffff);
    out[threadIdx.x * 4 + 1] = (uint16_t)((r >> 16) & 0xffff);
    out[threadIdx.x * 4 + 2] = (uint16_t)((r >> 32) & 0xffff);
    out[threadIdx.x * 4 + 3] = (uint16_t)((r >> 48) & 0xffff);
    // second pattern: lane l supplies byte address 32 (l & 15) + 8 (l >> 4)  (row-major [16 rows][16 halfs], lane group g
    // reads halfs 4g..4g+3 of row l & 15)
    const uint32_t addr2 = (uint32_t)(size_t)(__attribute__((address_space(3))) uint16_t*)lds + (threadIdx.x & 15) * 32 + (threadIdx.x >> 4) * 8;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr2) : "memory");
    out8[threadIdx.x * 4 + 0] = (uint16_t)(r & 0xffff);
    out8[threadIdx.x * 4 + 1] = (uint16_t)((r >> 16) & 0xffff);
    out8[threadIdx.x * 4 + 2] = (uint16_t)((r >> 32) & 0xffff);
    out8[threadIdx.x * 4 + 3] = (uint16_t)((r >> 48) & 0xffff);
}

int main() {
    uint16_t *d, *d2, h[256], h2[256];
    hipMalloc(&d, sizeof(h));
    hipMalloc(&d2, sizeof(h2));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, d2);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    hipMemcpy(h2, d2, sizeof(h2), hipMemcpyDeviceToHost);
    printf("pattern A (lane l owns halfs 4l..4l+3): lane: (src lane, src elem) x4\n");
    for (int l = 0; l < 64; ++l) {
        printf("%2d:", l);
        for (int j = 0; j < 4; ++j) printf(" (%2d,%d)", h[l * 4 + j] / 4, h[l * 4 + j] % 4);
        printf("\n");
    }
    printf("pattern B (row-major 16x16 halfs, lane (g,p) -> row p, halfs 4g..4g+3): lane: (row, col) x4\n");
    for (int l = 0; l < 64; ++l) {
        printf("%2d:", l);
        for (int j = 0; j < 4; ++j) printf(" (%2d,%2d)", h2[l * 4 + j] / 16, h2[l * 4 + j] % 16);
        printf("\n");
    }
    return 0;
}
