// Probe: operand and result layout of v_mfma_f32_32x32x16_f16 on gfx950 (the layout tests/test_p32_layout.py assumes).
// Hypothesis: A (32 x 16): lane l holds A[l & 31][8 (l >> 5) + j], j = 0..7;  B (16 x 32): lane l holds B[8 (l >> 5) + j][l & 31];
// D (32 x 32): register r of lane l holds D[(r & 3) + 8 (r >> 2) + 4 (l >> 5)][l & 31].  Random small-integer operands
// (exact in fp16 / fp32), asymmetric, compared with a host product.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma32_probe.hip -o gpurun_ab/mfma32_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k(const float* A, const float* B, float* D) {      // A [32][16], B [16][32], D [32][32] row-major
    const int l = threadIdx.x, m = l & 31, h = l >> 5;
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)A[m * 16 + 8 * h + j]; b[j] = (_Float16)B[(8 * h + j) * 32 + m]; }
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + m] = c[r];
}

int main() {
    float hA[32 * 16], hB[16 * 32], hD[32 * 32], ref[32 * 32];
    unsigned s = 12345u;
    for (auto& v : hA) { s = s * 1664525u + 1013904223u; v = (float)((int)(s >> 24) % 15 - 7); }
    for (auto& v : hB) { s = s * 1664525u + 1013904223u; v = (float)((int)(s >> 24) % 13 - 6); }
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { float t = 0; for (int kk = 0; kk < 16; ++kk) t += hA[i * 16 + kk] * hB[kk * 32 + j]; ref[i * 32 + j] = t; }
    float *dA, *dB, *dD;
    hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dD, sizeof(hD));
    hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 1024; ++i) bad += (hD[i] != ref[i]);
    printf("v_mfma_f32_32x32x16_f16 layout hypothesis: %s (%d of 1024 elements differ)\n", bad ? "WRONG" : "confirmed", bad);
    return bad != 0;
}
