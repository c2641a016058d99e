// C ABI of the Pose-NDF engine (include/posendf_amd.h): handle management, weight packing, launches.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <algorithm>
#include <cmath>
#include <vector>

#include "../../include/posendf_amd.h"
#include "pndf_layout.h"
#include "pndf_args.h"
#include "pndf_host.h"
#include "pndf_pack.h"
#include "pndf_generic.h"

using namespace pndf;

extern "C" __global__ void pndf_fused_relu_kernel(PndfKernelArgs args);
extern "C" __global__ void pndf_fused_softplus_kernel(PndfKernelArgs args);
extern "C" __global__ void pndf_fused_split_relu_kernel(PndfKernelArgs args);
extern "C" __global__ void pndf_fused_split_softplus_kernel(PndfKernelArgs args);
extern "C" __global__ void pndf_fused_split2_relu_kernel(PndfKernelArgs args);        // pndf_kernel_split_x2.hip
extern "C" __global__ void pndf_fused_split2_softplus_kernel(PndfKernelArgs args);
extern "C" __global__ void pndf_fused_half_relu_kernel(PndfKernelArgs args);
extern "C" __global__ void pndf_fused_bf16_relu_kernel(PndfKernelArgs args);     // pndf_kernel_bf16.hip
extern "C" long long pndf_kernel_softplus_scratch_floats_per_wg();
extern "C" int pndf_kernel_lds_bytes();


struct pndf_engine {
    pndf_config cfg;
    int device = 0;
    bool have_weights = false;
    // f16x3 only: every lo tile of the packed trunk is zero (weights exactly representable in fp16 at their layer scale),
    // so the lo hi term vanishes identically and the two-term kernels give bit-identical results with 2/3 of the MFMAs
    bool lo_all_zero = false;
    char* d_stream = nullptr;   // STEP_TILES KiB + a replica of the first STREAM_PAD_SLOTS slots (the ring never wraps)
    float* d_bias = nullptr;
    // softplus: fp32 derivative scratch, one block per RESIDENT workgroup (= per CU: the kernels take a whole CU each
    // and walk the 64-pose blocks with a grid-stride loop), allocated once in pndf_create.  Launches of one handle share
    // it, so launches on different streams are ordered with an event (sp_done recorded after each softplus launch).
    float* d_scratch = nullptr;
    int resident_wgs = 0;
    hipEvent_t sp_done = nullptr;
    void* sp_stream = nullptr;
    bool sp_pending = false;
    // any DFNet that is not shaped like configs/amass.yaml (another depth, wider layers): the runtime-planned path, pndf_generic.hip
    PndfGeneric* generic = nullptr;
    std::string err;
};

constexpr int STREAM_PAD_SLOTS = 5;           // >= RING_SLOTS - 1 of pndf_device.h (the ring's prefetch distance; 5 covers the six-buffer experiment arm)

static thread_local std::string g_create_err;

static int fail(pndf_engine* h, int code, const std::string& msg) {
    if (h) h->err = msg; else g_create_err = msg;
    return code;
}

#define HIP_TRY(h, expr)                                                                     \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return fail(h, PNDF_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

// The lab is quarantined from the product (pndf_experiment.h): every translation unit exports one word with a bit per tuning /
// ablation macro that differs from its product default; a product library reports 0.
PNDF_EXPORT_EXPERIMENT_WORD(capi)
extern "C" {
extern const unsigned pndf_experiment_word_fp32, pndf_experiment_word_split, pndf_experiment_word_split_x2, pndf_experiment_word_lbs,
    pndf_experiment_word_generic, pndf_experiment_word_bf16;
}
extern "C" unsigned pndf_experiment_word(void) {
    return pndf_experiment_word_capi | pndf_experiment_word_fp32 | pndf_experiment_word_split | pndf_experiment_word_split_x2 |
           pndf_experiment_word_lbs | pndf_experiment_word_generic | pndf_experiment_word_bf16;
}
extern "C" const char* pndf_version(void) {
    static const std::string v = [] {
        char w[16];
        snprintf(w, sizeof w, "0x%08x", pndf_experiment_word());
        return std::string("posendf_amd 0.4 (gfx950; fp32 MFMA 16x16x4 and split-fp16 MFMA 16x16x32; relu, lrelu, softplus; experiments=") + w + ")";
    }();
    return v.c_str();
}

extern "C" const char* pndf_last_error(pndf_handle h) { return h ? h->err.c_str() : g_create_err.c_str(); }

extern "C" void pndf_default_config(pndf_config* cfg, int32_t act, float beta) {
    memset(cfg, 0, sizeof(*cfg));
    cfg->act = act;
    cfg->beta = beta;
    cfg->num_joints = NJ;
    cfg->n_dims = NLIN + 1;
    for (int i = 0; i <= NLIN; ++i) cfg->dims[i] = DIMS[i];
    for (int i = 0; i < NJ; ++i) cfg->parent[i] = PARENT[i];
    cfg->precision = PNDF_PREC_FP32;
    cfg->enc_act = -1;      // the same activation as the trunk (every config of the reference)
    cfg->enc_beta = 0.f;
}

static int check_config(pndf_engine* h, const pndf_config* cfg) {
    if (!cfg) return fail(h, PNDF_ERR_BAD_ARG, "cfg is null");
    if (cfg->num_joints != NJ)
        return fail(h, PNDF_ERR_UNSUPPORTED, "only the 21-joint structure of get_parent_mapping('smpl') is implemented");
    // The fused kernels are laid out for configs/amass.yaml (126 | 84, 256, 512, 1024, 512, 256, 64, 1).  A DFNet of the same
    // depth whose hidden layers are NARROWER runs on them zero-padded (padded units have zero outgoing weights, so they
    // reach neither the distance nor its gradient, whatever the activation); any other `dims` list the reference can build
    // (net_modules.py:14-28) -- 2 .. 8 linear layers, hidden widths up to 1024 -- runs on the runtime-planned kernels of
    // pndf_generic.hip (exact fp32 whatever precision was asked for).  Only beyond that is a configuration refused.
    if (cfg->n_dims < 3 || cfg->n_dims > MAXLIN + 1)
        return fail(h, PNDF_ERR_UNSUPPORTED, "DFNet depth: n_dims must be 3 .. 9 (1 .. 7 hidden layers + the output layer)");
    if ((cfg->dims[0] != DIMS[0] && cfg->dims[0] != NOENC_IN) || cfg->dims[cfg->n_dims - 1] != 1)
        return fail(h, PNDF_ERR_UNSUPPORTED, "DFNet in_dim must be 126 (StrEnc.use=True) or 84 (False), its output 1");
    for (int i = 1; i < cfg->n_dims - 1; ++i)
        if (cfg->dims[i] < 1 || cfg->dims[i] > MAX_WIDTH)
            return fail(h, PNDF_ERR_UNSUPPORTED, "DFNet hidden widths must be 1 .. 1024");
    for (int i = 0; i < NJ; ++i)
        if (cfg->parent[i] != PARENT[i]) return fail(h, PNDF_ERR_UNSUPPORTED, "parent table must be get_parent_mapping('smpl')");
    if (cfg->act != PNDF_ACT_RELU && cfg->act != PNDF_ACT_LRELU && cfg->act != PNDF_ACT_SOFTPLUS)
        return fail(h, PNDF_ERR_UNSUPPORTED, "unknown activation (relu, lrelu and softplus are implemented)");
    if (cfg->act == PNDF_ACT_SOFTPLUS && !(cfg->beta > 0.f))
        return fail(h, PNDF_ERR_BAD_ARG, "softplus beta must be positive");
    if (cfg->enc_act != -1 && cfg->enc_act != PNDF_ACT_RELU && cfg->enc_act != PNDF_ACT_LRELU && cfg->enc_act != PNDF_ACT_SOFTPLUS)
        return fail(h, PNDF_ERR_UNSUPPORTED, "unknown encoder activation (relu, lrelu and softplus are implemented; -1 = the trunk's)");
    if (cfg->enc_act == PNDF_ACT_SOFTPLUS && !(cfg->enc_beta > 0.f) && !(cfg->beta > 0.f))
        return fail(h, PNDF_ERR_BAD_ARG, "softplus beta of the encoder must be positive");
    if (cfg->precision != PNDF_PREC_FP32 && cfg->precision != PNDF_PREC_F16X3 && cfg->precision != PNDF_PREC_F16 &&
        cfg->precision != PNDF_PREC_BF16)
        return fail(h, PNDF_ERR_UNSUPPORTED, "unknown precision (fp32, f16x3, f16 and bf16 are implemented)");
    if ((cfg->precision == PNDF_PREC_F16 || cfg->precision == PNDF_PREC_BF16) && cfg->act == PNDF_ACT_SOFTPLUS)
        return fail(h, PNDF_ERR_UNSUPPORTED, "the plain-f16 / plain-bf16 comparison kernels implement relu / lrelu only");
    return PNDF_OK;
}

extern "C" int pndf_create(pndf_handle* out, const pndf_config* cfg, int device) {
    if (!out) return fail(nullptr, PNDF_ERR_BAD_ARG, "out is null");
    *out = nullptr;
    int rc = check_config(nullptr, cfg);
    if (rc) return rc;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev)
        return fail(nullptr, PNDF_ERR_NO_DEVICE, "no HIP device " + std::to_string(device) + " (the engine has no CPU fallback)");
    hipDeviceProp_t prop;
    HIP_TRY(nullptr, hipGetDeviceProperties(&prop, device));
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
        return fail(nullptr, PNDF_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950 only");
    DeviceGuard guard(device);
    if (!guard.ok) return fail(nullptr, PNDF_ERR_HIP, "hipSetDevice failed");
    pndf_engine* h = new pndf_engine();
    h->cfg = *cfg;
    h->device = device;
    if (prop.multiProcessorCount <= 0) {
        delete h;
        return fail(nullptr, PNDF_ERR_HIP, "the device reports no compute units (multiProcessorCount <= 0)");
    }
    h->resident_wgs = prop.multiProcessorCount;
    if (pndf_generic_needed(*cfg)) {
        if (cfg->precision == PNDF_PREC_BF16) {
            delete h;
            return fail(nullptr, PNDF_ERR_UNSUPPORTED, "the plain-bf16 comparison kernel exists for amass.yaml-shaped networks only "
                                                       "(the runtime-planned kernels run fp32 or split-precision fp16)");
        }
        std::string why;
        rc = pndf_generic_create(&h->generic, *cfg, h->resident_wgs, why);
        if (rc != PNDF_OK) {
            delete h;
            return fail(nullptr, rc, why);
        }
        *out = h;
        return PNDF_OK;
    }
    hipError_t e = hipMalloc((void**)&h->d_stream, (size_t)(STEP_TILES + STREAM_PAD_SLOTS * SLOT_TILES) * TILE_BYTES);
    if (e == hipSuccess) e = hipMalloc((void**)&h->d_bias, BIAS_FLOATS * sizeof(float));
    if (e == hipSuccess && cfg->act == PNDF_ACT_SOFTPLUS) {
        const size_t sbytes = (size_t)h->resident_wgs * pndf_kernel_softplus_scratch_floats_per_wg() * sizeof(float);
        // PNDF_SP_SCRATCH=uncached|finegrained: memory-type experiments on the derivative scratch (profiles/r04/sp_forward_diag.txt)
        const char* mt = getenv("PNDF_SP_SCRATCH");
        if (mt && mt[0] == 'u') e = hipExtMallocWithFlags((void**)&h->d_scratch, sbytes, hipDeviceMallocUncached);
        else if (mt && mt[0] == 'f') e = hipExtMallocWithFlags((void**)&h->d_scratch, sbytes, hipDeviceMallocFinegrained);
        else e = hipMalloc((void**)&h->d_scratch, sbytes);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&h->sp_done, hipEventDisableTiming);
    }
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)pndf_fused_relu_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, pndf_kernel_lds_bytes());
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)pndf_fused_split_relu_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, pndf_kernel_lds_bytes());
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)pndf_fused_split_softplus_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, pndf_kernel_lds_bytes());
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)pndf_fused_split2_relu_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, pndf_kernel_lds_bytes());
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)pndf_fused_split2_softplus_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, pndf_kernel_lds_bytes());
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)pndf_fused_half_relu_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, pndf_kernel_lds_bytes());
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)pndf_fused_bf16_relu_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, pndf_kernel_lds_bytes());
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)pndf_fused_softplus_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, pndf_kernel_lds_bytes());
    if (e != hipSuccess) {
        std::string m = std::string("pndf_create: ") + hipGetErrorString(e);
        pndf_destroy(h);
        return fail(nullptr, PNDF_ERR_HIP, m);
    }
    *out = h;
    return PNDF_OK;
}

extern "C" int pndf_destroy(pndf_handle h) {
    if (!h) return PNDF_OK;
    DeviceGuard guard(h->device);
    pndf_generic_destroy(h->generic);
    if (h->sp_done) (void)hipEventDestroy(h->sp_done);
    if (h->d_stream) (void)hipFree(h->d_stream);
    if (h->d_bias) (void)hipFree(h->d_bias);
    if (h->d_scratch) (void)hipFree(h->d_scratch);
    delete h;
    return PNDF_OK;
}

// ------------------------------------------------------------------------------------------ packing
// (Mat / emit_tile / EncMat / emit_enc_tile: pndf_pack.h, shared with pndf_generic.hip)
using pndf_pack::Mat;
using pndf_pack::emit_tile;
using pndf_pack::EncMat;
using pndf_pack::emit_enc_tile;

extern "C" void pndf_packed_sizes(int64_t* stream_floats, int64_t* bias_floats) {
    if (stream_floats) *stream_floats = (int64_t)STEP_TILES * TILE_FLOATS;
    if (bias_floats) *bias_floats = BIAS_FLOATS;
}

// in_dim of the trunk: 126 with the structure encoder, 84 = 21 x 4 without it (model.StrEnc.use = False, reference
// model/posendf.py:40-42,73-74: DFNet sees the normalised quaternions, `p.reshape(len(p), -1)` net_modules.py:49)
static bool table_has_encoder(int n) { return n == 4 * NJ + 2 * NLIN; }

// widths of the DFNet in a state-dict table: dims[0] = in_dim (126 | 84), dims[l + 1] = rows of dfnet.lin{l}
struct NetDims {
    int d[NLIN + 1];
    int in(int l) const { return d[l]; }
    int out(int l) const { return d[l + 1]; }
};

static const char* check_tensors(const float* const* tensors, const int64_t* numel, int n, NetDims* dims_out = nullptr) {
    if (!tensors || !numel) return "tensors / numel is null";
    if (n != 4 * NJ + 2 * NLIN && n != 2 * NLIN)
        return "expected 98 tensors (encoder + dfnet) or 14 (dfnet only, StrEnc.use = False) in state-dict order";
    const bool enc = table_has_encoder(n);
    int t = 0;
    for (int j = 0; enc && j < NJ; ++j) {
        const int64_t want[4] = {HID * enc_in(j), HID, FEAT * HID, FEAT};
        for (int k = 0; k < 4; ++k, ++t)
            if (!tensors[t] || numel[t] != want[k]) return "encoder tensor missing or of the wrong size";
    }
    NetDims nd;
    nd.d[0] = enc ? DIMS[0] : NOENC_IN;
    for (int l = 0; l < NLIN; ++l) {             // the bias lengths are the layer widths
        const int64_t out = numel[t + 2 * l + 1];
        if (!tensors[t + 2 * l] || !tensors[t + 2 * l + 1] || out < 1 || out > DIMS[l + 1] || (l == NLIN - 1 && out != 1))
            return "dfnet tensor missing, or a layer wider than the configs/amass.yaml architecture (256,512,1024,512,256,64,1)";
        nd.d[l + 1] = (int)out;
    }
    for (int l = 0; l < NLIN; ++l)
        if (numel[t + 2 * l] != (int64_t)nd.out(l) * nd.in(l)) return "dfnet weight of the wrong size";
    if (dims_out) *dims_out = nd;
    return nullptr;
}

extern "C" int pndf_pack_host(const float* const* tensors, const int64_t* numel, int n_tensors, float* stream,
                              float* bias) {
    NetDims nd;
    if (check_tensors(tensors, numel, n_tensors, &nd) || !stream || !bias) return PNDF_ERR_BAD_SHAPE;
    // ---- bias block: b0..b5 | w6 | b6 | per joint: b1 padded to 16, b2 on rows 4..9 of 16
    memset(bias, 0, BIAS_FLOATS * sizeof(float));
    for (int l = 0; l < 8; ++l) bias[SCALE_OFF + l] = 1.0f;
    const bool enc = table_has_encoder(n_tensors);     // without the encoder its tiles / biases stay zero (skipped on chip)
    const float* const* lin = tensors + (enc ? 4 * NJ : 0);
    for (int l = 0; l < NLIN - 1; ++l) memcpy(bias + BIAS_OFF[l], lin[2 * l + 1], sizeof(float) * nd.out(l));   // narrower layers: zero padded
    memcpy(bias + W6_OFF, lin[2 * (NLIN - 1)], sizeof(float) * nd.in(NLIN - 1));
    bias[BIAS_OFF[NLIN - 1]] = lin[2 * (NLIN - 1) + 1][0];
    for (int j = 0; enc && j < NJ; ++j) {
        memcpy(bias + ENCB_OFF + 32 * j, tensors[4 * j + 1], sizeof(float) * HID);
        memcpy(bias + ENCB_OFF + 32 * j + 16 + ENC_FEAT_ROW, tensors[4 * j + 3], sizeof(float) * FEAT);
    }
    // ---- stream: encoder forward tiles | trunk phases | encoder backward tiles, in consumption order
    float* dst = stream;
    memset(stream, 0, (size_t)STEP_TILES * TILE_FLOATS * sizeof(float));
    for (int j = 0; enc && j < NJ; ++j) {                           // forward: joint order, W1 then W2
        for (int kind = 0; kind < 2; ++kind, dst += TILE_FLOATS)
            emit_enc_tile(EncMat{tensors[4 * j], tensors[4 * j + 2], enc_in(j), kind}, dst);
    }
    dst = stream + (size_t)ENC_TILES_PADDED * TILE_FLOATS;
    for (int ph = 0; ph < 6; ++ph) {
        const Phase& P = PHASES[ph];
        const Mat A{lin[2 * P.a_lin], nd.out(P.a_lin), nd.in(P.a_lin), P.transposed};
        const Mat B{lin[2 * P.b_lin], nd.out(P.b_lin), nd.in(P.b_lin), P.transposed};
        for (int c = 0; c < P.NC; ++c) {
            for (int kt = 0; kt < P.KA; ++kt)                       // part A: (kt, ci)
                for (int ci = 0; ci < P.CT; ++ci, dst += TILE_FLOATS) emit_tile(A, c * P.CT + ci, kt, dst);
            for (int nbp = 0; nbp < P.NB / 2; ++nbp)                // part B: (nbp, ci, h)
                for (int ci = 0; ci < P.CT; ++ci)
                    for (int hh = 0; hh < 2; ++hh, dst += TILE_FLOATS) emit_tile(B, 2 * nbp + hh, c * P.CT + ci, dst);
        }
    }
    if (dst - stream != (ptrdiff_t)(ENC_TILES_PADDED + TRUNK_FWD_TILES + TRUNK_BWD_TILES) * TILE_FLOATS) return PNDF_ERR_BAD_SHAPE;
    for (int j = NJ - 1; enc && j >= 0; --j) {                      // backward: reverse joint order, W2^T then W1^T
        for (int kind = 2; kind < 4; ++kind, dst += TILE_FLOATS)
            emit_enc_tile(EncMat{tensors[4 * j], tensors[4 * j + 2], enc_in(j), kind}, dst);
    }
    return PNDF_OK;
}

// ---- split-precision stream
// Exact power-of-two scaling (pndf_kernel_split.hip, "operand scaling"): the weights of layer l travel as s_l W with
// s_l = the power of two that brings the layer's largest |weight| into [2^12, 2^13) -- the hi halves cannot overflow and
// the lo halves of all weights down to 2^-14 of the largest stay in fp16's normal range; 1 / s_l goes to the bias block
// (SCALE_OFF + l).  Biases stay unscaled: activations and gradients are scaled PER POSE on the chip, and the packer only
// supplies the norms (NORM_OFF) from which the kernel derives guaranteed bounds for the layers it cannot measure.
namespace {
inline void split_f16(float w, _Float16& hi, _Float16& lo) {
    hi = (_Float16)w;                       // round to nearest even
    lo = (_Float16)(w - (float)hi);
#ifdef PNDF_EXP_LO_BITS                     // (energy experiment, profiles/r05/energy_breakdown.txt: the lo half keeps this many
    {                                       // explicit mantissa bits -- does the matrix pipe pay for operand bits that toggle?)
        uint16_t u = __builtin_bit_cast(uint16_t, lo);
        const int drop = 10 - PNDF_EXP_LO_BITS;
        u = (uint16_t)((u + (1u << (drop - 1))) & ~((1u << drop) - 1));
        lo = __builtin_bit_cast(_Float16, u);
    }
#endif
}
// fp32 -> bfloat16 bits, round to nearest even (the packer's inputs are finite: pack_host_split refuses a layer that is not)
inline uint16_t bf16_bits(float w) {
    const uint32_t u = __builtin_bit_cast(uint32_t, w);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
// block(M, nt, kb): hi tile then lo tile, 8 halfs per lane each; returns whether any lo half is non-zero.
// `bf16` (precision bf16, the one-term comparison kernel): the hi tile holds bfloat16 bits, the lo tile zeros.
bool emit_pair(const Mat& m, int nt, int kb, float* dst, float scale, bool bf16) {
    _Float16* hi = (_Float16*)dst;
    _Float16* lo = (_Float16*)(dst + TILE_FLOATS);
    bool any_lo = false;
    for (int lane = 0; lane < 64; ++lane)
        for (int jj = 0; jj < 8; ++jj) {
            const float w = m.at(16 * nt + (lane & 15), 16 * (2 * kb + (jj >> 2)) + 4 * (lane >> 4) + (jj & 3));
            if (bf16) {
                hi[lane * 8 + jj] = __builtin_bit_cast(_Float16, bf16_bits(w * scale));
                lo[lane * 8 + jj] = (_Float16)0;
                continue;
            }
            split_f16(w * scale, hi[lane * 8 + jj], lo[lane * 8 + jj]);
            any_lo |= (lo[lane * 8 + jj] != (_Float16)0);
        }
    return any_lo;
}
}  // namespace

static int pack_host_split(const float* const* tensors, const int64_t* numel, int n_tensors, float* stream, float* bias,
                           bool* lo_all_zero, bool bf16 = false);
extern "C" int pndf_pack_host_split(const float* const* tensors, const int64_t* numel, int n_tensors, float* stream,
                                    float* bias) {
    return pack_host_split(tensors, numel, n_tensors, stream, bias, nullptr);
}
static int pack_host_split(const float* const* tensors, const int64_t* numel, int n_tensors, float* stream, float* bias,
                           bool* lo_all_zero, bool bf16) {
    // biases and encoder tiles are identical to the fp32 stream (the encoder stays on fp32 MFMA)
    int rc = pndf_pack_host(tensors, numel, n_tensors, stream, bias);
    if (rc != PNDF_OK) return rc;
    NetDims nd;
    (void)check_tensors(tensors, numel, n_tensors, &nd);
    const bool enc = table_has_encoder(n_tensors);
    const float* const* lin = tensors + (enc ? 4 * NJ : 0);
    // per-layer weight scale; a layer without a finite non-zero weight cannot be scaled: refused (-> fp32 kernel)
    float wscale[6];
    for (int l = 0; l < 6; ++l) {
        float mx = 0.f;
        bool nan = false;
        const int64_t n = (int64_t)nd.out(l) * nd.in(l);
        for (int64_t i = 0; i < n; ++i) {
            const float a = std::fabs(lin[2 * l][i]);
            nan |= (a != a);
            if (a > mx) mx = a;
        }
        if (nan || !(mx > 0x1p-100f && mx < 0x1p100f)) return PNDF_ERR_UNSUPPORTED;
        int e;
        (void)std::frexp(mx, &e);                       // mx = f * 2^e, f in [0.5, 1)
        wscale[l] = std::ldexp(1.0f, 13 - e);           // s_l * mx in [2^12, 2^13)
        bias[SCALE_OFF + l] = 1.0f / wscale[l];
    }
    // norms for the a-priori bounds of the chunked layers, rounded UP (they must stay upper bounds in fp32)
    auto up = [](double v) { return std::nextafter((float)v, INFINITY); };
    for (int k = 0; k < 3; ++k) {
        const int lf = 2 * k, lb = 5 - 2 * k;           // forward chunk layers 0, 2, 4; backward chunk layers 5, 3, 1
        const int inf = nd.in(lf), outf = nd.out(lf);
        double rowmax = 0.0, bmax = 0.0;
        for (int o = 0; o < outf; ++o) {
            double sum = 0.0;
            for (int i = 0; i < inf; ++i) sum += std::fabs((double)lin[2 * lf][(size_t)o * inf + i]);
            rowmax = std::max(rowmax, sum);
            bmax = std::max(bmax, std::fabs((double)lin[2 * lf + 1][o]));
        }
        const int inb = nd.in(lb), outb = nd.out(lb);
        double colmax = 0.0;
        for (int i = 0; i < inb; ++i) {
            double sum = 0.0;
            for (int o = 0; o < outb; ++o) sum += std::fabs((double)lin[2 * lb][(size_t)o * inb + i]);
            colmax = std::max(colmax, sum);
        }
        bias[NORM_OFF + k] = up(rowmax);
        bias[NORM_OFF + 3 + k] = up(bmax);
        bias[NORM_OFF + 6 + k] = up(colmax);
    }
    float* dst = stream + (size_t)ENC_TILES_PADDED * TILE_FLOATS;
    bool any_lo = false;
    for (int ph = 0; ph < 6; ++ph) {
        Phase P = PHASES[ph];
        if (PNDF_BIG_CT != 2 && P.NC == 32) {      // (experiment: the two big phases in chunks of PNDF_BIG_CT tiles, pndf_layout.h)
            P.NC = P.NC * P.CT / PNDF_BIG_CT;
            P.CT = PNDF_BIG_CT;
        }
        const Mat A{lin[2 * P.a_lin], nd.out(P.a_lin), nd.in(P.a_lin), P.transposed};
        const Mat B{lin[2 * P.b_lin], nd.out(P.b_lin), nd.in(P.b_lin), P.transposed};
        auto partA = [&](int c) {
            for (int kb = 0; kb < P.KA / 2; ++kb)
                for (int ci = 0; ci < P.CT; ++ci, dst += 2 * TILE_FLOATS) any_lo |= emit_pair(A, c * P.CT + ci, kb, dst, wscale[P.a_lin], bf16);
        };
        auto partB = [&](int c) {
            for (int nb = 0; nb < P.NB; ++nb)
                for (int b = 0; b < P.CT / 2; ++b, dst += 2 * TILE_FLOATS) any_lo |= emit_pair(B, nb, (c * P.CT) / 2 + b, dst, wscale[P.b_lin], bf16);
        };
        partA(0);
        for (int c = 0; c < P.NC; ++c) {
            if (c + 1 < P.NC) partA(c + 1);
            partB(c);
        }
    }
    if (dst - stream != (ptrdiff_t)(ENC_TILES_PADDED + TRUNK_FWD_TILES + TRUNK_BWD_TILES) * TILE_FLOATS) return PNDF_ERR_BAD_SHAPE;
    if (lo_all_zero) *lo_all_zero = !any_lo;
    return PNDF_OK;
}

extern "C" int pndf_load_weights(pndf_handle h, const float* const* tensors, const int64_t* numel, int n_tensors) {
    if (!h) return PNDF_ERR_BAD_ARG;
    if (h->generic) {
        DeviceGuard guard(h->device);
        if (!guard.ok) return fail(h, PNDF_ERR_HIP, "hipSetDevice failed");
        std::string why;
        const int grc = pndf_generic_load(h->generic, tensors, numel, n_tensors, why);
        if (grc != PNDF_OK) return fail(h, grc, why);
        h->have_weights = true;
        return PNDF_OK;
    }
    NetDims nd;
    if (const char* why = check_tensors(tensors, numel, n_tensors, &nd)) return fail(h, PNDF_ERR_BAD_SHAPE, why);
    if (table_has_encoder(n_tensors) != (h->cfg.dims[0] == DIMS[0]))
        return fail(h, PNDF_ERR_BAD_SHAPE, "tensor table does not match the configured in_dim (98 tensors for 126, 14 for 84)");
    for (int i = 0; i <= NLIN; ++i)
        if (nd.d[i] != h->cfg.dims[i]) return fail(h, PNDF_ERR_BAD_SHAPE, "dfnet tensor shapes do not match the configured dims");
    std::vector<float> stream((size_t)STEP_TILES * TILE_FLOATS), bias(BIAS_FLOATS);
    bool lo_zero = false;
    const int prc = (h->cfg.precision != PNDF_PREC_FP32)
                        ? pack_host_split(tensors, numel, n_tensors, stream.data(), bias.data(), &lo_zero, h->cfg.precision == PNDF_PREC_BF16)
                        : pndf_pack_host(tensors, numel, n_tensors, stream.data(), bias.data());
    if (prc == PNDF_ERR_UNSUPPORTED)
        return fail(h, PNDF_ERR_UNSUPPORTED, "a trunk layer has no finite non-zero weight: outside the operating "
                                             "range of the fp16 hi/lo split -- use precision fp32 for this network");
    if (prc != PNDF_OK)
        return fail(h, PNDF_ERR_BAD_SHAPE, "internal: packed stream length mismatch");
    if (h->cfg.act == PNDF_ACT_SOFTPLUS) {
        // zero-padded units of a narrower network: softplus(0) = ln 2 / beta with derivative 1/2 is harmless for the
        // result (zero outgoing weights) but would enter the per-pose operand bounds the split kernels measure (largest
        // activation, largest derivative of a layer).  A bias of -1e6 makes both exactly zero; relu-family units are
        // zero at 0 anyway.
        for (int l = 0; l < NLIN - 1; ++l)
            for (int j = nd.out(l); j < DIMS[l + 1]; ++j) bias[BIAS_OFF[l] + j] = -1.0e6f;
    }
    DeviceGuard guard(h->device);
    if (!guard.ok) return fail(h, PNDF_ERR_HIP, "hipSetDevice failed");
    HIP_TRY(h, hipDeviceSynchronize());   // no launch may still be reading the old weights
    HIP_TRY(h, hipMemcpy(h->d_stream, stream.data(), stream.size() * sizeof(float), hipMemcpyHostToDevice));
    // replica of the first slots behind the stream: the ring's fetch offset never wraps inside a step
    HIP_TRY(h, hipMemcpy(h->d_stream + (size_t)STEP_TILES * TILE_BYTES, stream.data(),
                         (size_t)STREAM_PAD_SLOTS * SLOT_TILES * TILE_BYTES, hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->d_bias, bias.data(), bias.size() * sizeof(float), hipMemcpyHostToDevice));
    h->have_weights = true;
    // PNDF_THREE_TERMS=1 keeps the three-term kernels (same-box A/B runs and the bit-identity test)
    const char* keep3 = getenv("PNDF_THREE_TERMS");
    h->lo_all_zero = lo_zero && h->cfg.precision == PNDF_PREC_F16X3 && !(keep3 && keep3[0] == '1');
    return PNDF_OK;
}

// name of the kernel the compute calls of this handle launch (after pndf_load_weights), for logs, benches and tests
extern "C" const char* pndf_kernel_name(pndf_handle h) {
    if (!h) return "";
    if (h->generic) return pndf_generic_kernel_name(h->generic);
    const bool sp = h->cfg.act == PNDF_ACT_SOFTPLUS;
    switch (h->cfg.precision) {
        case PNDF_PREC_F16X3:
            if (h->lo_all_zero) return sp ? "pndf_fused_split2_softplus_kernel" : "pndf_fused_split2_relu_kernel";
            return sp ? "pndf_fused_split_softplus_kernel" : "pndf_fused_split_relu_kernel";
        case PNDF_PREC_F16: return "pndf_fused_half_relu_kernel";
        case PNDF_PREC_BF16: return "pndf_fused_bf16_relu_kernel";
        default: return sp ? "pndf_fused_softplus_kernel" : "pndf_fused_relu_kernel";
    }
}

// ------------------------------------------------------------------------------------------ launches
// `instrumented`: a kernel of the DEBUG library (libposendf_amd_debug.so: the s_memtime / stage-dump builds of the fused kernels) to
// launch instead of the handle's own, with the same arguments, grid and LDS -- pndf_internal_launch below; nullptr on every
// product path.  The debug library decides which of its kernels matches the handle (pndf_internal_describe).
static int launch(pndf_engine* h, int mode, const float* q, const float* gout, float* qo, float* d, int64_t B,
                  int steps, float* dbg, void* stream, const void* instrumented = nullptr) {
    const bool timing = instrumented != nullptr;
    if (!h) return PNDF_ERR_BAD_ARG;
    if (!h->have_weights) return fail(h, PNDF_ERR_NO_WEIGHTS, "pndf_load_weights has not been called");
    if (B < 0 || steps < 0) return fail(h, PNDF_ERR_BAD_ARG, "negative batch or step count");
    if (B == 0) return PNDF_OK;
    if (!q || (mode != MODE_FORWARD && !qo) || (mode == MODE_FORWARD && !d))
        return fail(h, PNDF_ERR_BAD_ARG, "null pose / output pointer");
    if (((uintptr_t)q | (uintptr_t)qo) & 15) return fail(h, PNDF_ERR_BAD_ARG, "pose buffers must be 16-byte aligned");
    if (((uintptr_t)d | (uintptr_t)gout) & 3) return fail(h, PNDF_ERR_BAD_ARG, "misaligned distance buffer");
    PndfKernelArgs a;
    if (h->generic) {
        if (dbg || timing) return fail(h, PNDF_ERR_UNSUPPORTED, "stage dumps and region timing exist for the amass.yaml-shaped kernels only");
        DeviceGuard gguard(h->device);
        if (!gguard.ok) return fail(h, PNDF_ERR_HIP, "hipSetDevice failed");
        if (mode == MODE_PROJECT && steps == 0) {      // zero iterations: the loop body never runs (sample_poses.py:70)
            if (qo != q) HIP_TRY(h, hipMemcpyAsync(qo, q, (size_t)B * NQ * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
            if (d) HIP_TRY(h, hipMemsetAsync(d, 0, (size_t)B * sizeof(float), (hipStream_t)stream));
            return PNDF_OK;
        }
        std::string why;
        const int grc = pndf_generic_launch(h->generic, mode, q, gout, qo, d, B, steps, stream, why);
        return grc == PNDF_OK ? PNDF_OK : fail(h, grc, why);
    }
    a.q_in = q; a.q_out = qo; a.d_out = d; a.grad_out = gout;
    a.stream = h->d_stream; a.bias = h->d_bias; a.dbg = dbg;
    a.B = B; a.steps = steps; a.mode = mode;
    a.slope = (h->cfg.act == PNDF_ACT_LRELU) ? 0.01f : 0.0f;   // nn.LeakyReLU() default slope, net_modules.py:31
    a.beta = h->cfg.beta;
    a.scratch = nullptr;
    a.reserved0 = 0;
    a.noenc = (h->cfg.dims[0] == NOENC_IN) ? 1 : 0;
    DeviceGuard guard(h->device);
    if (!guard.ok) return fail(h, PNDF_ERR_HIP, "hipSetDevice failed");
    const bool softplus = h->cfg.act == PNDF_ACT_SOFTPLUS;
    if (dbg && !timing) return fail(h, PNDF_ERR_BAD_ARG, "a dump buffer needs an instrumented kernel (libposendf_amd_debug.so)");
    if (timing && softplus && B > (int64_t)WG_POSES * h->resident_wgs)
        return fail(h, PNDF_ERR_UNSUPPORTED, "instrumented softplus kernel: at most one 64-pose block per compute unit");
    if (mode == MODE_PROJECT && steps == 0) {
        // zero iterations: the loop body never runs (sample_poses.py:70); poses pass through
        if (qo != q) HIP_TRY(h, hipMemcpyAsync(qo, q, (size_t)B * NQ * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
        if (d) HIP_TRY(h, hipMemsetAsync(d, 0, (size_t)B * sizeof(float), (hipStream_t)stream));
        return PNDF_OK;
    }
    const int64_t nblocks = (B + WG_POSES - 1) / WG_POSES;
    const dim3 grid((unsigned)nblocks), block(WG_THREADS);
    if (softplus) {
        // Persistent grid: at most one workgroup per CU (a workgroup takes a whole CU), each walking its blocks; the
        // derivative scratch is indexed by workgroup and was allocated in pndf_create -- nothing is allocated, freed or
        // synchronised here.  The scratch is shared by all launches of the handle: a launch on another stream than the
        // previous one first waits (on the device) for that one's completion event.
        const dim3 pgrid((unsigned)(nblocks < h->resident_wgs ? nblocks : h->resident_wgs));
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing((hipStream_t)stream, &cap);
        const bool capturing = cap != hipStreamCaptureStatusNone;
        if (h->sp_pending && h->sp_stream != stream && !capturing)
            HIP_TRY(h, hipStreamWaitEvent((hipStream_t)stream, h->sp_done, 0));
        a.scratch = h->d_scratch;
        if (timing) {   // instrumented: one workgroup per block like the relu timing kernel needs B <= 64 * resident_wgs
            void* kargs[] = {&a};
            HIP_TRY(h, hipLaunchKernel(instrumented, pgrid, block, kargs, pndf_kernel_lds_bytes(), (hipStream_t)stream));
        } else if (h->cfg.precision == PNDF_PREC_F16X3 && h->lo_all_zero)
            hipLaunchKernelGGL(pndf_fused_split2_softplus_kernel, pgrid, block, pndf_kernel_lds_bytes(), (hipStream_t)stream, a);
        else if (h->cfg.precision == PNDF_PREC_F16X3)
            hipLaunchKernelGGL(pndf_fused_split_softplus_kernel, pgrid, block, pndf_kernel_lds_bytes(), (hipStream_t)stream, a);
        else
            hipLaunchKernelGGL(pndf_fused_softplus_kernel, pgrid, block, pndf_kernel_lds_bytes(), (hipStream_t)stream, a);
        HIP_TRY(h, hipGetLastError());
        if (!capturing) {
            HIP_TRY(h, hipEventRecord(h->sp_done, (hipStream_t)stream));
            h->sp_stream = stream;
            h->sp_pending = true;
        }
        return PNDF_OK;
    }
    const bool split = h->cfg.precision == PNDF_PREC_F16X3, half = h->cfg.precision == PNDF_PREC_F16;
    if (timing) {
        void* kargs[] = {&a};
        HIP_TRY(h, hipLaunchKernel(instrumented, grid, block, kargs, pndf_kernel_lds_bytes(), (hipStream_t)stream));
    } else if (half) hipLaunchKernelGGL(pndf_fused_half_relu_kernel, grid, block, pndf_kernel_lds_bytes(), (hipStream_t)stream, a);
    else if (h->cfg.precision == PNDF_PREC_BF16) hipLaunchKernelGGL(pndf_fused_bf16_relu_kernel, grid, block, pndf_kernel_lds_bytes(), (hipStream_t)stream, a);
    else if (split && h->lo_all_zero) hipLaunchKernelGGL(pndf_fused_split2_relu_kernel, grid, block, pndf_kernel_lds_bytes(), (hipStream_t)stream, a);
    else if (split) hipLaunchKernelGGL(pndf_fused_split_relu_kernel, grid, block, pndf_kernel_lds_bytes(), (hipStream_t)stream, a);
    else hipLaunchKernelGGL(pndf_fused_relu_kernel, grid, block, pndf_kernel_lds_bytes(), (hipStream_t)stream, a);
    HIP_TRY(h, hipGetLastError());
    return PNDF_OK;
}

extern "C" int pndf_forward(pndf_handle h, const float* q, float* d, int64_t B, void* stream) {
    PndfRange range("pndf_forward");
    return launch(h, MODE_FORWARD, q, nullptr, nullptr, d, B, 1, nullptr, stream);
}

extern "C" int pndf_forward_grad(pndf_handle h, const float* q, const float* grad_out, float* d, float* dq,
                                 int64_t B, void* stream) {
    PndfRange range("pndf_forward_grad");
    return launch(h, MODE_FORWARD_GRAD, q, grad_out, dq, d, B, 1, nullptr, stream);
}

extern "C" int pndf_project(pndf_handle h, const float* q_in, float* q_out, float* d_last, int64_t B, int steps,
                            void* stream) {
    PndfRange range("pndf_project");
    return launch(h, MODE_PROJECT, q_in, nullptr, q_out, d_last, B, steps, nullptr, stream);
}

// ---- hooks for the debug library (libposendf_amd_debug.so; include/posendf_amd_debug.h).  Not declared in any installed header and
// not part of the boundary: the instrumented builds of the fused kernels, the stage-dump kernel and the memory probes live in a
// library of their own, which reaches the engine through these three entry points -- bound at run time by
// posendf_amd.engine (function addresses handed to pndf_debug_bind), so that a variant product library (PNDF_LIBRARY) is
// instrumented by its own debug library and never by the default one.
extern "C" int pndf_internal_launch(pndf_handle h, int mode, const float* q, const float* gout, float* qo, float* d, int64_t B, int steps,
                                    float* dbg, void* stream, const void* kernel) {
    if (!kernel) return fail(h, PNDF_ERR_BAD_ARG, "pndf_internal_launch: no kernel");
    return launch(h, mode, q, gout, qo, d, B, steps, dbg, stream, kernel);
}
// what[0..6] = precision, act, lo_all_zero, noenc, runtime-planned, resident workgroups, LDS bytes of a fused kernel
extern "C" int pndf_internal_describe(pndf_handle h, int* what, int n) {
    if (!h || !what || n < 7) return PNDF_ERR_BAD_ARG;
    what[0] = h->cfg.precision;
    what[1] = h->cfg.act;
    what[2] = h->lo_all_zero ? 1 : 0;
    what[3] = (h->cfg.dims[0] == NOENC_IN) ? 1 : 0;
    what[4] = h->generic ? 1 : 0;
    what[5] = h->resident_wgs;
    what[6] = pndf_kernel_lds_bytes();
    return PNDF_OK;
}
extern "C" int pndf_internal_fail(pndf_handle h, int code, const char* msg) { return fail(h, code, msg ? msg : ""); }
