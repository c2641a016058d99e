// Probe (round 4): is the operand split's lo half the same bit pattern when the remainder and its rounding to fp16 are ONE
// instruction each (v_fma_mixlo_f16 / v_fma_mixhi_f16: fma in fp32, result rounded to fp16 into one half of the register)
// instead of v_fma_mix_f32 + v_cvt_pk_f16_f32 (three instead of four instructions per pair of values)?
// Inputs: random fp32 over 2^-30 .. 2^17 of both signs, values next to fp16 rounding boundaries, fp16 subnormal range,
// zeros, the largest operands the kernels can meet (|x| < 2^14 after scaling) and beyond (overflow of hi), inf, NaN.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-inline-asm tools/ubench/split_probe.hip -o gpurun_ab/split_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split_four(float a, float b, unsigned& hi, unsigned& lo) {
    const unsigned hp = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(a, b));
    hi = hp;
    float ra, rb;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(ra) : "v"(hp), "v"(a));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(rb) : "v"(hp), "v"(b));
    f16x2 l;
    l[0] = (_Float16)ra;
    l[1] = (_Float16)rb;
    lo = __builtin_bit_cast(unsigned, l);
}
__device__ __forceinline__ void split_three(float a, float b, unsigned& hi, unsigned& lo) {
    const unsigned hp = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(a, b));
    hi = hp;
    unsigned l;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=&v"(l) : "v"(hp), "v"(a), "v"(b));
    lo = l;
}

__global__ void probe(const float* x, int n, unsigned* h4, unsigned* l4, unsigned* h3, unsigned* l3) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    split_four(x[2 * i], x[2 * i + 1], h4[i], l4[i]);
    split_three(x[2 * i], x[2 * i + 1], h3[i], l3[i]);
}

int main() {
    const int n = 1 << 22;
    std::vector<float> x(n);
    uint64_t s = 0x9e3779b97f4a7c15ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (int i = 0; i < n; ++i) {
        const uint64_t r = rnd();
        const int e = (int)(r % 48) - 30;                                   // 2^-30 .. 2^17
        const double m = 1.0 + (double)((r >> 8) & 0xffffff) / 16777216.0;
        x[i] = (float)(((r >> 40) & 1) ? -ldexp(m, e) : ldexp(m, e));
    }
    // edges: halfway points of the fp16 grid at several exponents +- one fp32 ulp, fp16 subnormals, zeros, large, inf, NaN
    int k = 0;
    for (int e = -26; e <= 16; ++e)
        for (int q = 0; q < 8; ++q) {
            const float base = ldexpf(1.0f + (float)(q * 131 % 1024) / 1024.0f + 1.0f / 2048.0f, e);      // exactly between two fp16 values
            x[k++] = base; x[k++] = nextafterf(base, 0.f); x[k++] = nextafterf(base, 1e30f); x[k++] = -base;
        }
    const float special[] = {0.f, -0.f, 65504.f, 65519.f, 65520.f, 1e5f, -1e5f, INFINITY, -INFINITY, NAN, 5.96e-8f, 2.98e-8f, 6.1e-5f, 1e-38f, 1e-45f};
    for (float v : special) x[k++] = v;
    float* dx; unsigned* o[4];
    hipMalloc(&dx, n * 4);
    for (auto& p : o) hipMalloc(&p, n * 2);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(n / 2 / 256), dim3(256), 0, 0, dx, n, o[0], o[1], o[2], o[3]);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    std::vector<unsigned> r[4];
    for (int j = 0; j < 4; ++j) { r[j].resize(n / 2); hipMemcpy(r[j].data(), o[j], n * 2, hipMemcpyDeviceToHost); }
    long bad_hi = 0, bad_lo = 0, nan_only = 0;
    for (int i = 0; i < n / 2; ++i) {
        if (r[0][i] != r[2][i]) ++bad_hi;
        if (r[1][i] != r[3][i]) {
            // NaN payloads may differ: count separately
            auto is_nan16 = [](unsigned h) { return (h & 0x7c00u) == 0x7c00u && (h & 0x3ffu); };
            const bool n0 = is_nan16(r[1][i] & 0xffff) == is_nan16(r[3][i] & 0xffff) && is_nan16(r[1][i] >> 16) == is_nan16(r[3][i] >> 16);
            const bool same_non_nan = ((is_nan16(r[1][i] & 0xffff) || (r[1][i] & 0xffff) == (r[3][i] & 0xffff)) &&
                                       (is_nan16(r[1][i] >> 16) || (r[1][i] >> 16) == (r[3][i] >> 16)));
            if (n0 && same_non_nan) ++nan_only; else {
                if (bad_lo < 8) printf("  differs: x = (%.9g, %.9g)  lo4 = %08x  lo3 = %08x\n", x[2 * i], x[2 * i + 1], r[1][i], r[3][i]);
                ++bad_lo;
            }
        }
    }
    printf("split probe: %d pairs, hi differs %ld, lo differs %ld (NaN payload only: %ld)\n", n / 2, bad_hi, bad_lo, nan_only);
    return (bad_hi || bad_lo) ? 2 : 0;
}
