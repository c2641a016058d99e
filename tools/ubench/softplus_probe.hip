// Probe of the softplus device functions of pndf_device.h on the hardware (round 4): sp_consts, act_softplus (scalar),
// act_softplus2 (pair, packed fp32), act_softplus4 against a double-precision host evaluation of nn.Softplus(beta,
// threshold=20) and its derivative, plus two ISA questions the packed forms raise on gfx950:
//   * does a packed fp32 instruction read BOTH registers of an SGPR pair source (v_pk_mul_f32 v[..], v[..], s[n:n+1])?
//   * does an inline constant reach both halves (v_pk_add_f32 ..., 1.0 op_sel_hi:[1,0])?
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-inline-asm -I posendf_amd/csrc tools/ubench/softplus_probe.hip -o gpurun_ab/softplus_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <vector>
#include "pndf_device.h"

__global__ void probe(const float* z, int n, float beta, float* out_s, float* out_d, float* out2, float* out2d, float* out4,
                      float* out4d, float* consts, float* isa) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const SpK k = sp_consts(beta);
    if (i == 0) {
        consts[0] = k.beta; consts[1] = k.b2; consts[2] = k.c; consts[3] = k.invb;
        // SGPR pair source with two different values
        float lo, hi;
        asm volatile("s_mov_b32 s8, 0x40000000\n\ts_mov_b32 s9, 0x40400000\n\tv_mov_b32 v10, 1.0\n\tv_mov_b32 v11, 1.0\n\t"
                     "s_nop 4\n\tv_pk_mul_f32 v[12:13], v[10:11], s[8:9]\n\ts_nop 4\n\tv_mov_b32 %0, v12\n\tv_mov_b32 %1, v13"
                     : "=v"(lo), "=v"(hi) : : "s8", "s9", "v10", "v11", "v12", "v13");
        isa[0] = lo; isa[1] = hi;                  // expected 2, 3 when both SGPRs are read; 2, 2 when the low one is replicated
        asm volatile("v_mov_b32 v10, 2.0\n\tv_mov_b32 v11, 4.0\n\ts_nop 4\n\tv_pk_add_f32 v[12:13], v[10:11], 1.0 op_sel_hi:[1,0]\n\t"
                     "s_nop 4\n\tv_mov_b32 %0, v12\n\tv_mov_b32 %1, v13"
                     : "=v"(lo), "=v"(hi) : : "v10", "v11", "v12", "v13");
        isa[2] = lo; isa[3] = hi;                  // expected 3, 5
    }
    if (i >= n) return;
    float d;
    out_s[i] = act_softplus(z[i], k, d);
    out_d[i] = d;
    if ((i & 1) == 0 && i + 1 < n) {
        f32x2 dd;
        const f32x2 y = act_softplus2(f32x2{z[i], z[i + 1]}, k, dd);
        out2[i] = y[0]; out2[i + 1] = y[1]; out2d[i] = dd[0]; out2d[i + 1] = dd[1];
    }
    if ((i & 3) == 0 && i + 3 < n) {
        f32x4 v = f32x4{z[i], z[i + 1], z[i + 2], z[i + 3]}, dv;
        act_softplus4(v, k, dv);
        for (int r = 0; r < 4; ++r) { out4[i + r] = v[r]; out4d[i + r] = dv[r]; }
    }
}

int main() {
    const int n = 4096;
    for (float beta : {100.0f, 1.0f, 10.0f, 1000.0f}) {
        std::vector<float> z(n);
        for (int i = 0; i < n; ++i) {
            const double t = (i / (double)(n - 1)) * 2.0 - 1.0;                 // [-1, 1]
            z[i] = (float)(t * t * t * 60.0 / beta);                              // beta z in [-60, 60], dense near 0
        }
        z[7] = 0.0f; z[8] = 20.0f / beta; z[9] = 20.0001f / beta; z[10] = 19.9999f / beta; z[11] = -200.0f / beta;
        float *dz, *o[6], *dc, *di;
        hipMalloc(&dz, n * 4); hipMalloc(&dc, 16); hipMalloc(&di, 16);
        for (auto& p : o) { hipMalloc(&p, n * 4); hipMemset(p, 0, n * 4); }
        hipMemcpy(dz, z.data(), n * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(n / 256), dim3(256), 0, 0, dz, n, beta, o[0], o[1], o[2], o[3], o[4], o[5], dc, di);
        if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
        std::vector<float> r[6];
        for (int j = 0; j < 6; ++j) { r[j].resize(n); hipMemcpy(r[j].data(), o[j], n * 4, hipMemcpyDeviceToHost); }
        float c[4], isa[4];
        hipMemcpy(c, dc, 16, hipMemcpyDeviceToHost); hipMemcpy(isa, di, 16, hipMemcpyDeviceToHost);
        printf("beta %g: consts beta %.9g b2 %.9g (want %.9g) c %.9g (want %.9g) invb %.9g | SGPR-pair pk_mul -> (%g, %g) want (2, 3) | "
               "inline-const pk_add -> (%g, %g) want (3, 5)\n", beta, c[0], c[1], beta * 1.4426950408889634, c[2], 0.6931471805599453 / beta,
               c[3], isa[0], isa[1], isa[2], isa[3]);
        const char* names[3] = {"act_softplus ", "act_softplus2", "act_softplus4"};
        for (int f = 0; f < 3; ++f) {
            double ey = 0, ed = 0; int wy = -1, wd = -1;
            for (int i = 0; i < n - 4; ++i) {
                const double bz = (double)beta * z[i];
                const double y = bz > 20.0 ? (double)z[i] : log1p(exp(bz)) / beta;
                const double d = bz > 20.0 ? 1.0 : exp(bz) / (1.0 + exp(bz));
                const double scale = fmax(fabs(y), 1e-30);
                const double e1 = fabs(r[2 * f][i] - y) / scale, e2 = fabs(r[2 * f + 1][i] - d) / fmax(d, 1e-30);
                if (e1 > ey) { ey = e1; wy = i; }
                if (e2 > ed) { ed = e2; wd = i; }
            }
            printf("  %s: max rel err value %.3e (z = %g, got %g)  derivative %.3e (z = %g, got %g)\n", names[f], ey, z[wy], r[2 * f][wy], ed,
                   z[wd], r[2 * f + 1][wd]);
        }
    }
    return 0;
}
