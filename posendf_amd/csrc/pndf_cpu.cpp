// Host twins of the three compute entry points (`pndf_*_cpu`): the same contract as pndf_forward /
// pndf_forward_grad / pndf_project (include/posendf_amd.h) on HOST pointers, for a caller whose `train.device` is "cpu"
// (reference model/posendf.py:35,64 moves the pose to whatever device the config names).  Written from scratch in plain
// C++ for the host cores -- NOT the oracle (oracle/ is test infrastructure and is imported by nothing here) and NOT a
// fallback: the device entry points never route here, a missing GPU still fails loudly there.
//
// Layout: poses are processed in blocks of PB = 32; inside a block every activation tensor is [feature][pose] so that the
// inner loop of a layer runs over the 32 poses of the block (contiguous floats: one or two AVX-512 / four AVX2 registers)
// with the weight as a broadcast scalar, four output rows at a time.  The backward pass runs the same loop on transposed
// copies of the weights made once at load time.  Blocks are dealt to std::threads (PNDF_CPU_THREADS, default: the
// hardware concurrency, at most one thread per block).  Arithmetic is fp32 throughout; the activation conventions are
// PyTorch's (nn.LeakyReLU slope 0.01 with derivative `x > 0 ? 1 : slope`, nn.ReLU on the output of the relu family,
// nn.Softplus(beta, threshold 20) and softplus_backward's e / (e + 1)); F.normalize(dim=1) with eps 1e-12; the update
// q - d * grad keeps the reference's two roundings (sample_poses.py:74).
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/posendf_amd.h"
#include "pndf_layout.h"

using namespace pndf;

namespace {

constexpr int PB = 32;

struct Layer {
    int in = 0, out = 0;
    std::vector<float> w, wt, b;      // w [out][in], wt [in][out], b [out]
};

}  // namespace

struct pndf_cpu_engine {
    pndf_config cfg;
    bool encoder = true;
    bool have_weights = false;
    Layer enc[NJ][2];
    Layer lin[MAXLIN];      // any DFNet depth the reference can build (net_modules.py:14-28: `dims` is a free list): 2 .. 8 linear layers
    int dims[MAXLIN + 1];
    int nlin = NLIN;
    std::string err;
};

namespace {

thread_local std::string g_cpu_create_err;      // (per thread, like pndf_lbs_last_error(nullptr))

int cpu_fail(pndf_cpu_engine* h, int code, const std::string& msg) {
    (h ? h->err : g_cpu_create_err) = msg;
    return code;
}

// y[o][p] = b[o] + sum_i w[o][i] x[i][p]   (four rows of y at a time; the compiler vectorises the p loops)
__attribute__((target_clones("avx512f", "arch=haswell", "default")))
void dense(const float* w, const float* b, int out, int in, const float* x, float* y) {
    int o = 0;
    for (; o + 4 <= out; o += 4) {
        float a0[PB], a1[PB], a2[PB], a3[PB];
        for (int p = 0; p < PB; ++p) {
            a0[p] = b ? b[o] : 0.f; a1[p] = b ? b[o + 1] : 0.f; a2[p] = b ? b[o + 2] : 0.f; a3[p] = b ? b[o + 3] : 0.f;
        }
        const float *w0 = w + (size_t)o * in, *w1 = w0 + in, *w2 = w1 + in, *w3 = w2 + in;
        for (int i = 0; i < in; ++i) {
            const float* xi = x + (size_t)i * PB;
            const float c0 = w0[i], c1 = w1[i], c2 = w2[i], c3 = w3[i];
            for (int p = 0; p < PB; ++p) {
                a0[p] += c0 * xi[p]; a1[p] += c1 * xi[p]; a2[p] += c2 * xi[p]; a3[p] += c3 * xi[p];
            }
        }
        memcpy(y + (size_t)o * PB, a0, sizeof(a0)); memcpy(y + (size_t)(o + 1) * PB, a1, sizeof(a1));
        memcpy(y + (size_t)(o + 2) * PB, a2, sizeof(a2)); memcpy(y + (size_t)(o + 3) * PB, a3, sizeof(a3));
    }
    for (; o < out; ++o) {
        float a0[PB];
        for (int p = 0; p < PB; ++p) a0[p] = b ? b[o] : 0.f;
        const float* w0 = w + (size_t)o * in;
        for (int i = 0; i < in; ++i) {
            const float* xi = x + (size_t)i * PB;
            const float c0 = w0[i];
            for (int p = 0; p < PB; ++p) a0[p] += c0 * xi[p];
        }
        memcpy(y + (size_t)o * PB, a0, sizeof(a0));
    }
}

struct Act {
    int kind;      // pndf_act
    float beta;
};

// in place: z -> act(z); der <- act'(z)   (n values)
void activate(const Act& a, float* z, float* der, int n, bool output_layer) {
    if (a.kind == PNDF_ACT_SOFTPLUS) {
        for (int i = 0; i < n; ++i) {
            const float bz = z[i] * a.beta;
            if (bz > 20.0f) {
                der[i] = 1.0f;
            } else {
                const float e = expf(bz);
                der[i] = e / (e + 1.0f);
                z[i] = log1pf(e) / a.beta;
            }
        }
        return;
    }
    const float slope = (a.kind == PNDF_ACT_LRELU && !output_layer) ? 0.01f : 0.0f;      // net_modules.py:30-37
    for (int i = 0; i < n; ++i) {
        const bool pos = z[i] > 0.0f;
        der[i] = pos ? 1.0f : slope;
        // relu(NaN) = NaN, lrelu(NaN) = NaN as in PyTorch
        z[i] = (z[i] != z[i]) ? z[i] : (pos ? z[i] : z[i] * slope);
    }
}

struct Scratch {      // per thread
    std::vector<float> n, x[MAXLIN + 1], dx[MAXLIN + 1], g, g2, eh[NJ], ehd[NJ], ef[NJ], efd[NJ], ein[NJ], gf, gh, gin;
    float inv[4][PB], nrm[4][PB];
};

void forward_grad_block(const pndf_cpu_engine& E, const float* q /*[nb][84]*/, int nb, const float* gout, float* d, float* dq,
                        bool want_grad, Scratch& S) {
    const Act act{E.cfg.act, E.cfg.beta};
    // model.StrEnc.act / beta are read on their own (net_modules.py:128); -1 / <= 0: the trunk's
    const Act eact{E.cfg.enc_act == -1 ? E.cfg.act : E.cfg.enc_act, E.cfg.enc_beta > 0.f ? E.cfg.enc_beta : E.cfg.beta};
    // ---- normalise over joints per component (posendf.py:71), poses of the block as the inner index
    S.n.assign((size_t)NQ * PB, 0.f);
    for (int c = 0; c < 4; ++c)
        for (int p = 0; p < PB; ++p) {
            float ss = 0.f;
            if (p < nb)
                for (int j = 0; j < NJ; ++j) ss += q[(size_t)p * NQ + 4 * j + c] * q[(size_t)p * NQ + 4 * j + c];
            const float norm = sqrtf(ss);
            S.nrm[c][p] = norm;
            S.inv[c][p] = 1.0f / std::max(norm, 1e-12f);
        }
    for (int j = 0; j < NJ; ++j)
        for (int c = 0; c < 4; ++c)
            for (int p = 0; p < nb; ++p) S.n[(size_t)(4 * j + c) * PB + p] = q[(size_t)p * NQ + 4 * j + c] * S.inv[c][p];
    // ---- structure encoder (net_modules.py:140-170) or the normalised pose itself
    const int d0 = E.dims[0];
    S.x[0].assign((size_t)d0 * PB, 0.f);
    if (E.encoder) {
        for (int j = 0; j < NJ; ++j) {
            const int in = enc_in(j);
            S.ein[j].assign((size_t)in * PB, 0.f);
            memcpy(S.ein[j].data(), &S.n[(size_t)4 * j * PB], sizeof(float) * 4 * PB);
            if (PARENT[j] >= 0) memcpy(S.ein[j].data() + 4 * PB, S.ef[PARENT[j]].data(), sizeof(float) * FEAT * PB);
            S.eh[j].resize((size_t)HID * PB); S.ehd[j].resize((size_t)HID * PB);
            S.ef[j].resize((size_t)FEAT * PB); S.efd[j].resize((size_t)FEAT * PB);
            dense(E.enc[j][0].w.data(), E.enc[j][0].b.data(), HID, in, S.ein[j].data(), S.eh[j].data());
            activate(eact, S.eh[j].data(), S.ehd[j].data(), HID * PB, false);
            dense(E.enc[j][1].w.data(), E.enc[j][1].b.data(), FEAT, HID, S.eh[j].data(), S.ef[j].data());
            activate(eact, S.ef[j].data(), S.efd[j].data(), FEAT * PB, false);
            memcpy(&S.x[0][(size_t)FEAT * j * PB], S.ef[j].data(), sizeof(float) * FEAT * PB);      // cat(f_0 .. f_20), :169
        }
    } else {
        memcpy(S.x[0].data(), S.n.data(), sizeof(float) * NQ * PB);
    }
    // ---- DFNet (net_modules.py:46-72)
    const int NL = E.nlin;
    for (int l = 0; l < NL; ++l) {
        const Layer& L = E.lin[l];
        S.x[l + 1].resize((size_t)L.out * PB);
        S.dx[l + 1].resize((size_t)L.out * PB);
        dense(L.w.data(), L.b.data(), L.out, L.in, S.x[l].data(), S.x[l + 1].data());
        activate(act, S.x[l + 1].data(), S.dx[l + 1].data(), L.out * PB, l == NL - 1);
    }
    for (int p = 0; p < nb; ++p) d[p] = S.x[NL][p];
    if (!want_grad) return;
    // ---- d (sum_b grad_out_b d_b) / d q: reverse pass
    S.g.assign((size_t)PB, 0.f);
    for (int p = 0; p < nb; ++p) S.g[p] = (gout ? gout[p] : 1.0f) * S.dx[NL][p];
    for (int l = NL - 1; l >= 0; --l) {
        const Layer& L = E.lin[l];
        S.g2.resize((size_t)L.in * PB);
        dense(L.wt.data(), nullptr, L.in, L.out, S.g.data(), S.g2.data());
        if (l > 0)
            for (size_t i = 0; i < (size_t)L.in * PB; ++i) S.g2[i] *= S.dx[l][i];
        S.g.swap(S.g2);
    }
    // S.g = d / d x0  [d0][PB]
    std::vector<float>& gn = S.g2;
    gn.assign((size_t)NQ * PB, 0.f);
    if (E.encoder) {
        // children before parents: joints in decreasing index order (every parent has a smaller index)
        std::vector<float>& gfeat = S.gf;
        gfeat.assign(S.g.begin(), S.g.begin() + (size_t)NFEAT * PB);      // accumulates the children's contributions
        for (int j = NJ - 1; j >= 0; --j) {
            const int in = enc_in(j);
            float* gfj = &gfeat[(size_t)FEAT * j * PB];
            for (int i = 0; i < FEAT * PB; ++i) gfj[i] *= S.efd[j][i];
            S.gh.resize((size_t)HID * PB);
            dense(E.enc[j][1].wt.data(), nullptr, HID, FEAT, gfj, S.gh.data());
            for (int i = 0; i < HID * PB; ++i) S.gh[i] *= S.ehd[j][i];
            S.gin.resize((size_t)in * PB);
            dense(E.enc[j][0].wt.data(), nullptr, in, HID, S.gh.data(), S.gin.data());
            memcpy(&gn[(size_t)4 * j * PB], S.gin.data(), sizeof(float) * 4 * PB);
            if (PARENT[j] >= 0) {
                float* gp = &gfeat[(size_t)FEAT * PARENT[j] * PB];
                for (int i = 0; i < FEAT * PB; ++i) gp[i] += S.gin[(size_t)4 * PB + i];
            }
        }
    } else {
        memcpy(gn.data(), S.g.data(), sizeof(float) * NQ * PB);
    }
    // ---- F.normalize backward: n = q / max(||q_c||, eps) per component column
    for (int c = 0; c < 4; ++c)
        for (int p = 0; p < nb; ++p) {
            float dot = 0.f;
            for (int j = 0; j < NJ; ++j) dot += gn[(size_t)(4 * j + c) * PB + p] * q[(size_t)p * NQ + 4 * j + c];
            const float norm = S.nrm[c][p], den = std::max(norm, 1e-12f);
            const float kk = (norm > 1e-12f) ? dot / (den * den * norm) : 0.f;      // the clamped branch has no norm term
            for (int j = 0; j < NJ; ++j)
                dq[(size_t)p * NQ + 4 * j + c] = gn[(size_t)(4 * j + c) * PB + p] / den - q[(size_t)p * NQ + 4 * j + c] * kk;
        }
}

int threads_for(int64_t blocks) {
    int n = (int)std::thread::hardware_concurrency();
    if (const char* e = getenv("PNDF_CPU_THREADS")) n = atoi(e);
    if (n < 1) n = 1;
    return (int)std::min<int64_t>(n, blocks);
}

// (throws std::runtime_error when a worker or a thread start failed: the entry points turn it into PNDF_ERR_HOST)
template <class F>
void parallel_blocks(int64_t B, F&& body) {
    const int64_t blocks = (B + PB - 1) / PB;
    const int nt = threads_for(blocks);
    std::atomic<bool> failed{false};
    auto worker = [&](int t) {
        try {
            Scratch S;
            for (int64_t blk = t; blk < blocks && !failed.load(std::memory_order_relaxed); blk += nt)
                body(blk * PB, (int)std::min<int64_t>(PB, B - blk * PB), S);
        } catch (...) {      // (an allocation of the scratch: nothing else in a block throws)
            failed.store(true);
        }
    };
    if (nt == 1) {
        worker(0);
    } else {
        std::vector<std::thread> pool;
        try {
            for (int t = 1; t < nt; ++t) pool.emplace_back(worker, t);
        } catch (...) {
            failed.store(true);      // the blocks of the threads that did not start stay undone: the call fails as a whole
        }
        if (!failed.load()) worker(0);
        for (auto& th : pool) th.join();
    }
    if (failed.load()) throw std::runtime_error("a worker thread could not be started or ran out of memory");
}

int check(pndf_cpu_engine* h, const void* q, int64_t B) {
    if (!h) return PNDF_ERR_BAD_ARG;
    if (!h->have_weights) return cpu_fail(h, PNDF_ERR_NO_WEIGHTS, "pndf_cpu_load_weights has not been called");
    if (B < 0) return cpu_fail(h, PNDF_ERR_BAD_ARG, "negative batch");
    if (B > 0 && !q) return cpu_fail(h, PNDF_ERR_BAD_ARG, "null pose pointer");
    return PNDF_OK;
}

// No C++ exception may cross the C ABI: a failed thread start or allocation becomes a status code with its text.
template <class F>
int guarded(pndf_cpu_engine* h, F&& body) {
    try {
        body();
        return PNDF_OK;
    } catch (const std::exception& e) {
        return cpu_fail(h, PNDF_ERR_HOST, std::string("host resource failure: ") + e.what());
    } catch (...) {
        return cpu_fail(h, PNDF_ERR_HOST, "host resource failure");
    }
}

}  // namespace

extern "C" int pndf_cpu_create(pndf_cpu_handle* out, const pndf_config* cfg) {
    if (!out || !cfg) return cpu_fail(nullptr, PNDF_ERR_BAD_ARG, "null argument");
    *out = nullptr;
    if (cfg->act < PNDF_ACT_RELU || cfg->act > PNDF_ACT_SOFTPLUS) return cpu_fail(nullptr, PNDF_ERR_UNSUPPORTED, "unknown activation");
    if (cfg->num_joints != NJ || cfg->n_dims < 3 || cfg->n_dims > MAXLIN + 1)
        return cpu_fail(nullptr, PNDF_ERR_UNSUPPORTED, "21 joints and a DFNet of 2 .. 8 linear layers (n_dims 3 .. 9)");
    const int nlin = cfg->n_dims - 1;
    for (int j = 0; j < NJ; ++j)
        if (cfg->parent[j] != PARENT[j]) return cpu_fail(nullptr, PNDF_ERR_UNSUPPORTED, "parent table other than net_utils.py:46");
    if (cfg->dims[0] != NFEAT && cfg->dims[0] != NOENC_IN) return cpu_fail(nullptr, PNDF_ERR_UNSUPPORTED, "DFNet in_dim must be 126 (encoder) or 84");
    if (cfg->dims[nlin] != 1) return cpu_fail(nullptr, PNDF_ERR_UNSUPPORTED, "DFNet must end in one output");
    for (int l = 1; l < nlin; ++l)
        if (cfg->dims[l] < 1 || cfg->dims[l] > MAX_WIDTH)      // the same architectures the device engine accepts
            return cpu_fail(nullptr, PNDF_ERR_UNSUPPORTED, "hidden widths 1 .. 1024");
    if (cfg->act == PNDF_ACT_SOFTPLUS && !(cfg->beta > 0.f)) return cpu_fail(nullptr, PNDF_ERR_BAD_ARG, "Softplus beta must be positive");
    if (cfg->enc_act < -1 || cfg->enc_act > PNDF_ACT_SOFTPLUS) return cpu_fail(nullptr, PNDF_ERR_UNSUPPORTED, "unknown encoder activation");
    if (cfg->enc_act == PNDF_ACT_SOFTPLUS && !(cfg->enc_beta > 0.f) && !(cfg->beta > 0.f)) return cpu_fail(nullptr, PNDF_ERR_BAD_ARG, "Softplus beta of the encoder must be positive");
    return guarded(nullptr, [&] {
        pndf_cpu_engine* h = new pndf_cpu_engine();
        h->cfg = *cfg;
        h->encoder = cfg->dims[0] == NFEAT;
        h->nlin = nlin;
        for (int l = 0; l <= nlin; ++l) h->dims[l] = cfg->dims[l];
        *out = h;
    });
}

extern "C" int pndf_cpu_destroy(pndf_cpu_handle h) {
    delete h;
    return PNDF_OK;
}

extern "C" const char* pndf_cpu_last_error(pndf_cpu_handle h) { return h ? h->err.c_str() : g_cpu_create_err.c_str(); }

extern "C" int pndf_cpu_load_weights(pndf_cpu_handle h, const float* const* tensors, const int64_t* numel, int n_tensors) {
    if (!h || !tensors || !numel) return PNDF_ERR_BAD_ARG;
    const int want = (h->encoder ? 4 * NJ : 0) + 2 * h->nlin;
    if (n_tensors != want) return cpu_fail(h, PNDF_ERR_BAD_SHAPE, "tensor count: " + std::to_string(n_tensors) + ", expected " + std::to_string(want));
    h->have_weights = false;      // a failed load leaves no half-loaded engine behind
    int rc = PNDF_OK;
    const int grc = guarded(h, [&] {
        int t = 0;
        auto take = [&](Layer& L, int out, int in) -> bool {
            if (!tensors[t] || !tensors[t + 1] || numel[t] != (int64_t)out * in || numel[t + 1] != out) return false;
            L.in = in; L.out = out;
            L.w.assign(tensors[t], tensors[t] + (size_t)out * in);
            L.b.assign(tensors[t + 1], tensors[t + 1] + out);
            L.wt.resize((size_t)in * out);
            for (int o = 0; o < out; ++o)
                for (int i = 0; i < in; ++i) L.wt[(size_t)i * out + o] = L.w[(size_t)o * in + i];
            t += 2;
            return true;
        };
        if (h->encoder)
            for (int j = 0; j < NJ && rc == PNDF_OK; ++j)
                if (!take(h->enc[j][0], HID, enc_in(j)) || !take(h->enc[j][1], FEAT, HID))
                    rc = cpu_fail(h, PNDF_ERR_BAD_SHAPE, "encoder tensor " + std::to_string(t) + " has the wrong size");
        for (int l = 0; l < h->nlin && rc == PNDF_OK; ++l)
            if (!take(h->lin[l], h->dims[l + 1], h->dims[l]))
                rc = cpu_fail(h, PNDF_ERR_BAD_SHAPE, "dfnet.lin" + std::to_string(l) + " has the wrong size");
    });
    if (grc != PNDF_OK) return grc;
    if (rc != PNDF_OK) return rc;
    h->have_weights = true;
    return PNDF_OK;
}

extern "C" int pndf_forward_cpu(pndf_cpu_handle h, const float* q, float* d, int64_t B) {
    if (int rc = check(h, q, B)) return rc;
    if (B > 0 && !d) return cpu_fail(h, PNDF_ERR_BAD_ARG, "null output pointer");
    return guarded(h, [&] {
        parallel_blocks(B, [&](int64_t p0, int nb, Scratch& S) { forward_grad_block(*h, q + p0 * NQ, nb, nullptr, d + p0, nullptr, false, S); });
    });
}

extern "C" int pndf_forward_grad_cpu(pndf_cpu_handle h, const float* q, const float* grad_out, float* d, float* dq, int64_t B) {
    if (int rc = check(h, q, B)) return rc;
    if (B > 0 && !dq) return cpu_fail(h, PNDF_ERR_BAD_ARG, "null output pointer");
    return guarded(h, [&] {
        parallel_blocks(B, [&](int64_t p0, int nb, Scratch& S) {
            float dd[PB];
            forward_grad_block(*h, q + p0 * NQ, nb, grad_out ? grad_out + p0 : nullptr, dd, dq + p0 * NQ, true, S);
            if (d) memcpy(d + p0, dd, sizeof(float) * nb);
        });
    });
}

extern "C" int pndf_project_cpu(pndf_cpu_handle h, const float* q_in, float* q_out, float* d_last, int64_t B, int steps) {
    if (int rc = check(h, q_in, B)) return rc;
    if (steps < 0 || (B > 0 && !q_out)) return cpu_fail(h, PNDF_ERR_BAD_ARG, "negative step count or null output pointer");
    return guarded(h, [&] {
    parallel_blocks(B, [&](int64_t p0, int nb, Scratch& S) {
        float qb[PB * NQ], dqb[PB * NQ], dd[PB];
        memcpy(qb, q_in + p0 * NQ, sizeof(float) * nb * NQ);
        for (int p = 0; p < nb; ++p) dd[p] = 0.f;
        for (int s = 0; s < steps; ++s) {
            forward_grad_block(*h, qb, nb, nullptr, dd, dqb, true, S);
            for (int p = 0; p < nb; ++p)
                for (int i = 0; i < NQ; ++i) {
                    volatile float prod = dd[p] * dqb[p * NQ + i];      // two roundings, as the reference evaluates q - d * grad
                    qb[p * NQ + i] = qb[p * NQ + i] - prod;
                }
        }
        memcpy(q_out + p0 * NQ, qb, sizeof(float) * nb * NQ);
        if (d_last) memcpy(d_last + p0, dd, sizeof(float) * nb);
    });
    });
}
