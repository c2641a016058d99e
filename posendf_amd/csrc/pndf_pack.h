// Host-side packing helpers shared by the C-ABI translation units (pndf_capi.hip: the amass.yaml streams; pndf_generic.hip:
// runtime-planned networks): logical matrix views and the lane-linear 16 x 16 tile both kernel families read.
#pragma once
#include <stddef.h>

#include "pndf_layout.h"

namespace pndf_pack {
using namespace pndf;

struct Mat {          // logical matrix view M[r][c] of dfnet.lin{l}.weight, optionally transposed, zero padded
    const float* w;   // (out, in) row-major
    int out, in;
    bool transposed;
    float at(int r, int c) const {
        const int o = transposed ? c : r, i = transposed ? r : c;
        return (o < out && i < in) ? w[(size_t)o * in + i] : 0.f;
    }
};

// tile(M, nt, kt)[lane*4 + s] = M[16 nt + (lane & 15)][16 kt + 4 (lane >> 4) + s]   (pndf_layout.h)
inline void emit_tile(const Mat& m, int nt, int kt, float* dst) {
    for (int lane = 0; lane < 64; ++lane)
        for (int s = 0; s < 4; ++s) dst[lane * 4 + s] = m.at(16 * nt + (lane & 15), 16 * kt + 4 * (lane >> 4) + s);
}

// split-precision block (pndf_layout.h "split-precision stream"): hi tile then lo tile of (16 rows x 32 k), 8 halfs per lane each,
//   block(M, nt, kb)[lane * 8 + jj] = scale * M[16 nt + (lane & 15)][16 (2 kb + (jj >> 2)) + 4 (lane >> 4) + (jj & 3)]
// hi = the value rounded to nearest even, lo = the remainder rounded to nearest (pndf_capi.hip emit_pair: the amass.yaml stream)
inline void emit_pair_f16(const Mat& m, int nt, int kb, float scale, float* dst) {
    _Float16* hi = (_Float16*)dst;
    _Float16* lo = (_Float16*)(dst + TILE_FLOATS);
    for (int lane = 0; lane < 64; ++lane)
        for (int jj = 0; jj < 8; ++jj) {
            const float w = scale * m.at(16 * nt + (lane & 15), 16 * (2 * kb + (jj >> 2)) + 4 * (lane >> 4) + (jj & 3));
            const _Float16 h = (_Float16)w;
            hi[lane * 8 + jj] = h;
            lo[lane * 8 + jj] = (_Float16)(w - (float)h);
        }
}

// 16x16 logical matrices of one encoder joint (zero padded), see pndf_layout.h "encoder on the MFMA pipe"
struct EncMat {
    const float* w1;   // [10][in]
    const float* w2;   // [6][10]
    int in;
    int kind;          // 0: W1 (rows = hidden, k = input)   1: W2 (rows 4..9 = feature, k = hidden)
                       // 2: W2^T (rows = hidden, k = feature row 4..9)   3: W1^T (rows = input, k = hidden)
    float at(int r, int c) const {
        switch (kind) {
            case 0: return (r < HID && c < in) ? w1[r * in + c] : 0.f;
            case 1: return (r >= ENC_FEAT_ROW && r < ENC_FEAT_ROW + FEAT && c < HID) ? w2[(r - ENC_FEAT_ROW) * HID + c] : 0.f;
            case 2: return (r < HID && c >= ENC_FEAT_ROW && c < ENC_FEAT_ROW + FEAT) ? w2[(c - ENC_FEAT_ROW) * HID + r] : 0.f;
            default: return (r < in && c < HID) ? w1[c * in + r] : 0.f;
        }
    }
};
inline void emit_enc_tile(const EncMat& m, float* dst) {
    for (int lane = 0; lane < 64; ++lane)
        for (int s = 0; s < 4; ++s) dst[lane * 4 + s] = m.at(lane & 15, 4 * (lane >> 4) + s);
}

// the encoder's two stream sections: forward = joint order, W1 then W2; backward = reverse joint order, W2^T then W1^T.
// `tensors`: the 84 encoder tensors in state-dict order; each destination holds ENC_TILES (42) tiles.
inline void emit_encoder_sections(const float* const* tensors, float* fwd, float* bwd) {
    for (int j = 0; j < NJ; ++j)
        for (int kind = 0; kind < 2; ++kind, fwd += TILE_FLOATS)
            emit_enc_tile(EncMat{tensors[4 * j], tensors[4 * j + 2], enc_in(j), kind}, fwd);
    for (int j = NJ - 1; j >= 0; --j)
        for (int kind = 2; kind < 4; ++kind, bwd += TILE_FLOATS)
            emit_enc_tile(EncMat{tensors[4 * j], tensors[4 * j + 2], enc_in(j), kind}, bwd);
}

}  // namespace pndf_pack
