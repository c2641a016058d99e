// Fused Pose-NDF distance / gradient / projection kernel for gfx950 (MI355X, CDNA4), exact fp32.
//
// One workgroup = 4 waves = 64 poses; one wave = 16 poses, one wave per SIMD, whole 512-register file.
// Per projection step (reference experiments/sample_poses.py:70-74) a wave runs, entirely on chip:
//   normalise over joints (model/posendf.py:71) -> 21 BoneMLPs along the kinematic tree
//   (model/network/net_modules.py:162-169; VALU, weights via scalar loads) -> trunk forward
//   (net_modules.py:46-72; v_mfma_f32_16x16x4_f32, activations resident in registers, weights streamed
//   global -> LDS by DMA) -> d -> trunk backward (same MFMA, transposed weight tiles) -> encoder backward
//   -> normalise backward -> q <- q - d * grad.
// See pndf_layout.h for the register/tile layout and DESIGN.md for the roofline.
#include "pndf_device.h"
#include <utility>

namespace {

template <int... I, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

// One fused layer pair.  xin: KA input tiles (B operands); acc: NB output tiles (accumulators).
// Forward: chunk accumulators start from the A-layer bias, get the activation, and their sign bits are
// parked in LDS; backward: chunk accumulators start at 0 and are multiplied by the parked derivative.
// Weight tiles are consumed in groups of GT = 2 CT tiles; the group after the current one is read from
// LDS before the current group's MFMAs are issued (hipcc does not software-pipeline this by itself).
template <int KA, int CT, int NC, int NB, bool BWD, bool SP, bool GTIME = false>
struct PhaseBody {
    static constexpr int GT = 2 * CT;                 // tiles per group
    static constexpr int NGA = KA / 2;                // part-A groups (two k-tiles each)
    static constexpr int NGB = NB / 2;                // part-B groups (two output tiles each)
    static constexpr int NG = NGA + NGB;
    static constexpr int CHUNK_TILES = CT * KA + NB * CT;
    static_assert(CHUNK_TILES % SLOT_TILES == 0, "chunk body must be whole slots");
    static_assert(KA % 2 == 0 && NB % 2 == 0, "tiles are consumed in pairs");

    // groups GI .. NG-1 of one chunk; `cur` holds group GI's tiles on entry and the next chunk's group 0
    // (or garbage after the very last group) on exit.
    // MORE: another chunk follows (the last chunk of a phase is a separate instantiation, so that `loaded` is a
    // compile-time constant: as a run-time condition it put a branch behind every MFMA of the chunk's last group)
    template <int GI, bool MORE>
    static __device__ __forceinline__ void groups(const f32x4 (&xin)[KA], f32x4 (&acc)[NB], f32x4 (&ch)[CT],
                                                  f32x4 (&cur)[GT], Ring& ring, uint8_t* mask, const ActP& ap,
                                                  int spslot, int c, RegionClock* rc, const f32x4 (&dpre)[CT]) {
        if constexpr (GI < NG) {
            f32x4 nxt[GT];
            constexpr int TNEXT = ((GI + 1) * GT) % CHUNK_TILES;
            constexpr bool MID = group_has_mid<GT, TNEXT>();
            constexpr bool loaded = (GI + 1 < NG) || MORE;
            // The next group's tiles are read ONE PER MFMA STEP of this group (not as a burst: right after the
            // workgroup barrier all four waves would queue their reads at once with an empty MFMA pipe), the ring
            // events ride on the read they belong to (always read 0 of the group), and the slot fetch that follows
            // a mid-slot barrier is issued as four 1-KiB DMA pieces behind the group's last four MFMA steps.
            DmaSrc dsrc{nullptr, 0u};
            uint32_t ddst = 0;
            auto feed = [&](auto pc, auto npc) {
                constexpr int P = decltype(pc)::value, NP = decltype(npc)::value;
                __builtin_amdgcn_sched_barrier(0);
                if (loaded) {
                    if constexpr (P < GT) {
                        constexpr int t = (TNEXT + P) % SLOT_TILES;
                        if constexpr (t == 0) ring_boundary(ring);
                        if constexpr (t == SLOT_TILES / 2) {
                            ring_midslot_sync(ring);
                            ring_dma_begin(ring, dsrc, ddst);
                        }
                        nxt[P] = ring_tile(ring, t);
                    }
                    if constexpr (MID && P >= NP - 4) ring_dma_piece(dsrc, ddst, P - (NP - 4));
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            if constexpr (GI < NGA) {
                // ---- part A: two k-tiles of the chunk rows; tile order (kt, ci)
                static_for<8>([&](auto pc) {
                    constexpr int P = decltype(pc)::value, k2 = P / 4, s = P % 4;
#pragma unroll
                    for (int ci = 0; ci < CT; ++ci)
                        ch[ci] = mfma4(cur[k2 * CT + ci][s], xin[2 * GI + k2][s], ch[ci]);
                    feed(pc, std::integral_constant<int, 8>{});
                });
                if constexpr (GI == NGA - 1) {
                    // ---- chunk epilogue
                    if constexpr (SP) {
#pragma unroll
                        for (int ci = 0; ci < CT; ++ci) {
                            if (!BWD) {
                                f32x4 dv;
                                act_softplus4<PNDF_SP_FORM_CHUNK, true>(ch[ci], ap.k, dv);
                                ap.sp.put<1>(spslot + c * CT + ci, dv);
                            } else {
                                if (ci == 0) wait_staged_derivatives<(KA * CT / 4 < 12) ? KA * CT / 4 : 12>();
                                ch[ci] = ch[ci] * staged_derivative_tile(ap.stage, ci, ap.lane);
                            }
                        }
                    } else if (!BWD) {
                        // sign bits are summed as st * 2^k in ONE fp32 chain (exact below 2^24): a float chain cannot
                        // be reassociated or deferred piecewise, so every step value dies at once
                        float bitsum = 0.f;
#pragma unroll
                        for (int ci = 0; ci < CT; ++ci) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const float st = step01(ch[ci][r]);
                                ch[ci][r] = ch[ci][r] * relu_factor(st, ap.slope);
                                bitsum = fmaf(st, (float)(1u << (ci * 4 + r)), bitsum);
                            }
                        }
                        store_chunk_bits<CT>(mask, c, (uint32_t)bitsum);
                    } else {
                        const uint32_t bits = load_chunk_bits<CT>(mask, c);
#pragma unroll
                        for (int ci = 0; ci < CT; ++ci) {
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                ch[ci][r] = ch[ci][r] * relu_factor((float)((bits >> (ci * 4 + r)) & 1u), ap.slope);
                        }
                    }
                }
            } else {
                // ---- part B: two output tiles get this chunk's contribution; tile order (ci, h)
                constexpr int nbp = GI - NGA;
                static_for<CT * 4>([&](auto pc) {
                    constexpr int P = decltype(pc)::value, ci = P / 4, s = P % 4;
#pragma unroll
                    for (int h = 0; h < 2; ++h)
                        acc[2 * nbp + h] = mfma4(cur[ci * 2 + h][s], ch[ci][s], acc[2 * nbp + h]);
                    feed(pc, std::integral_constant<int, CT * 4>{});
                });
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < GT; ++i) cur[i] = nxt[i];
            if constexpr (GTIME) {
                const unsigned long long now = __builtin_amdgcn_s_memtime();
                rc->grp[GI] += now - rc->last;
                rc->last = now;
            }
            groups<GI + 1, MORE>(xin, acc, ch, cur, ring, mask, ap, spslot, c, rc, dpre);
        }
    }
};

template <int KA, int CT, int NC, int NB, bool BWD, bool SP, bool GTIME = false>
__device__ __forceinline__ void run_phase(const f32x4 (&xin)[KA], f32x4 (&acc)[NB], Ring& ring,
                                          const float* biasA, uint8_t* mask, const ActP& ap, int spslot, int g,
                                          RegionClock* rc = nullptr) {
    using Body = PhaseBody<KA, CT, NC, NB, BWD, SP, GTIME>;
    f32x4 cur[Body::GT];
    load_group<Body::GT, 0>(cur, ring);
    auto chunk = [&](int c, auto more) {
        f32x4 ch[CT];
#pragma unroll
        for (int ci = 0; ci < CT; ++ci) {
            if (!BWD) ch[ci] = *(const f32x4*)(biasA + 16 * (c * CT + ci) + 4 * g);
            else ch[ci] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        // backward softplus: the chunk's parked derivatives are fetched here, a whole part A before the epilogue needs
        // them, by DMA into the wave's LDS staging window (stage_derivative_tile); part A issues KA * CT / 4 ring pieces
        f32x4 dpre[CT];
#pragma unroll
        for (int ci = 0; ci < CT; ++ci) {
            dpre[ci] = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (SP && BWD) stage_derivative_tile(ap.sp, spslot + c * CT + ci, ap.stage + ci * 1024);
        }
        if constexpr (GTIME) rc->last = __builtin_amdgcn_s_memtime();
        Body::template groups<0, decltype(more)::value>(xin, acc, ch, cur, ring, mask, ap, spslot, c, rc, dpre);
    };
    for (int c = 0; c + 1 < NC; ++c) chunk(c, std::true_type{});
    chunk(NC - 1, std::false_type{});
}

template <int NT, bool SP>
__device__ __forceinline__ void act_tiles(f32x4 (&x)[NT], uint32_t (&m)[(NT * 4 + 31) / 32], const ActP& ap, int spslot) {
    if constexpr (SP) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            f32x4 dv;
            act_softplus4<PNDF_SP_FORM_TILES, true>(x[t], ap.k, dv);
            ap.sp.put<2>(spslot + t, dv);
        }
    } else {
        constexpr int NW = (NT * 4 + 31) / 32;
        float lo[NW], hi[NW];      // 16 sign bits each, summed as st * 2^k in fp32 chains (see run_phase)
#pragma unroll
        for (int w = 0; w < NW; ++w) lo[w] = hi[w] = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float st = step01(x[t][r]);
                x[t][r] = x[t][r] * relu_factor(st, ap.slope);
                const int w = (t * 4 + r) / 32, b = (t * 4 + r) % 32;
                if (b < 16) lo[w] = fmaf(st, (float)(1u << b), lo[w]);
                else hi[w] = fmaf(st, (float)(1u << (b - 16)), hi[w]);
            }
        }
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            m[w] = (uint32_t)lo[w] | ((uint32_t)hi[w] << 16);
            // pin the packing HERE: LLVM otherwise sinks the whole chain to its first use in the backward pass
            // and keeps (spills) all 128 step values until then (each reload = a memory round trip)
            asm volatile("" : "+v"(m[w]));
        }
    }
}

template <int NT, bool SP>
__device__ __forceinline__ void dact_tiles(f32x4 (&gx)[NT], const uint32_t (&m)[(NT * 4 + 31) / 32], const ActP& ap, int spslot) {
    // softplus: an opaque copy of the lane offset keeps hipcc from carrying the forward pass's NT slot addresses (64 bits each)
    // across the whole step to here and spilling them (see dact_split_tiles in pndf_kernel_split.hip)
    SpRef sp = ap.sp;
    if constexpr (SP) asm volatile("" : "+v"(sp.off));
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if constexpr (SP) {
            gx[t] = gx[t] * sp.get(spslot + t);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                gx[t][r] = gx[t][r] * relu_factor((float)((m[(t * 4 + r) / 32] >> ((t * 4 + r) % 32)) & 1u), ap.slope);
        }
    }
}

}  // namespace

template <bool DBG, bool SP, bool TIMING = false>
__device__ __forceinline__ void pndf_fused_body(const PndfKernelArgs& args) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4;          // lane group = k_local of the MFMA
    const int p = lane & 15;          // pose within the wave
    const int wp = wave * 16 + p;     // pose within the workgroup
    ActP ap;
    ap.slope = args.slope;
    ap.k = sp_consts(args.beta);
    ap.sp = SpRef{SP ? (const char*)(args.scratch + (size_t)blockIdx.x * SP_WG_FLOATS) : nullptr, (uint32_t)tid * SP_LANE_BYTES};
    ap.stage = (char*)(smem + LDS_F) + wave * (16 * FSTRIDE * 4);
    ap.lane = lane;
    float* const lds_bias = (float*)(smem + LDS_BIAS);
    uint8_t* const lds_mask = (uint8_t*)(smem + LDS_MASK) + tid;
    float* const lds_q = (float*)(smem + LDS_Q);
    float* const my_q = lds_q + wp * NQ;
    float* const my_f = (float*)(smem + LDS_F) + wp * FSTRIDE;
    float* const my_gn = (float*)(smem + LDS_GN) + wp * FSTRIDE;   // aliases the feature row of the pose
    float* const dbg = (DBG && blockIdx.x == 0) ? args.dbg : nullptr;

    Ring ring;
    ring.gstream = args.stream;
    ring.smem = smem;
    ring.lane = lane;
    ring.st_wait = ring.st_bar = 0;
    ring.st_n = 0;
    // ---- stage the biases in LDS once (coalesced)
    for (int i = tid; i < BIAS_FLOATS / 4; i += WG_THREADS)
        ((f32x4*)lds_bias)[i] = ((const f32x4*)args.bias)[i];

    // A workgroup owns the 64-pose blocks blockIdx.x, blockIdx.x + gridDim.x, ...: the relu-family kernels are launched
    // with one workgroup per block, the softplus kernels with at most one workgroup per CU so that the derivative scratch
    // is bounded by the resident workgroups (pndf_capi.hip).  Every block restarts the ring (the forward-only mode leaves
    // it in the middle of the stream); the drain + barrier at the end of a block make that safe.
    const long long nblocks = (args.B + WG_POSES - 1) / WG_POSES;
    for (long long blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
    const long long pose0 = blk * WG_POSES;
    ring_start(ring, wave);   // slots 0..3 in flight; the __syncthreads() below makes them visible
    {
        long long nvalid = args.B - pose0;
        if (nvalid > WG_POSES) nvalid = WG_POSES;
        const f32x4* src = (const f32x4*)(args.q_in + pose0 * NQ);
        const int nvec = (int)nvalid * (NQ / 4);
        for (int i = tid; i < WG_POSES * (NQ / 4); i += WG_THREADS) {
            // poses past the end of the batch replicate the last valid pose (never written back)
            const int src_i = i < nvec ? i : (nvec - (NQ / 4) + (i % (NQ / 4)));
            ((f32x4*)lds_q)[i] = src[src_i];
        }
    }
    ring_wait_dma();
    __syncthreads();

    const int nsteps = (args.mode == MODE_PROJECT) ? args.steps : 1;
    float dval = 0.f;
    RegionClock rc;
    if constexpr (TIMING) {
#pragma unroll
        for (int i = 0; i < TIMING_REGIONS; ++i) rc.acc[i] = 0;
#pragma unroll
        for (int i = 0; i < TIMING_GROUPS; ++i) rc.grp[i] = 0;
        rc.last = __builtin_amdgcn_s_memtime();
    }
    const int g_launch = g;
    for (int step = 0; step < nsteps; ++step) {
        // softplus kernels: LICM hoists ~220 LDS addresses of the form constant + 16 g out of this loop and, under their
        // higher register pressure, spills them -- and every spill reload is a VMEM load whose vmcnt(0) drains the
        // ring's DMA.  An opaque copy of g per step keeps those one-instruction address computations inside the step.
        // The same goes for the ~200 64-bit addresses of the derivative scratch slots (ap.sp + slot * 4 KiB).
        int g = g_launch;
        if (step) ring_next_step(ring);
        if constexpr (SP) {
            asm volatile("" : "+v"(g));
            asm volatile("" : "+v"(ap.sp.off));
        }
        uint32_t eb[6];
        float poison = 0.f;      // softplus kernels: NaN for a pose that holds a NaN / infinity (joint_axis_norms), else +0
        uint32_t m2[4], m4[4], m6[1];
        f32x4 x6[4];
        f32x4 x4[32];
        {
            f32x4 x2[32];
            {
                // ---------------- normalise + encoder forward (posendf.py:71, net_modules.py:162-169)
                if (args.noenc) {
                    poison = noenc_forward<SP>(my_q, my_f, g);
                    ring_skip_encoder_section(ring);
                } else {
                    poison = encoder_forward<SP>(my_q, my_f, lds_bias + ENCB_OFF, eb, ring, ap, g);
                }
                if (DBG && dbg && step == 0) {
                    for (int i = 0; i < NFEAT; ++i) dbg[(size_t)(DBG_FEAT + i) * WG_THREADS + tid] = my_f[i];
                }
                // B operands of lin0: lane (g, p) holds features 16 kt + 4 g + s of pose p
                f32x4 x0[8];
#pragma unroll
                for (int kt = 0; kt < 8; ++kt) x0[kt] = *(const f32x4*)(my_f + 16 * kt + 4 * g);
                tick<TIMING>(rc, 0);
                // ---------------- trunk forward
                load_bias<32>(x2, lds_bias + BIAS_OFF[1], g);
                run_phase<8, 2, 8, 32, false, SP>(x0, x2, ring, lds_bias + BIAS_OFF[0],
                                                  lds_mask + MASK_BASE[0] * WG_THREADS, ap, SP_SLOT_CHUNK[0], g);
            }
            tick<TIMING>(rc, 1);
            act_tiles<32, SP>(x2, m2, ap, SP_SLOT_X2);
            tick<TIMING>(rc, 2);
            if (DBG && dbg && step == 0) dump_tiles<32>(dbg, DBG_X2, x2, tid);
            load_bias<32>(x4, lds_bias + BIAS_OFF[3], g);
            run_phase<32, 2, 32, 32, false, SP, TIMING>(x2, x4, ring, lds_bias + BIAS_OFF[2],
                                                        lds_mask + MASK_BASE[1] * WG_THREADS, ap, SP_SLOT_CHUNK[1], g, &rc);
        }
        tick<TIMING>(rc, 3);
        act_tiles<32, SP>(x4, m4, ap, SP_SLOT_X4);
        tick<TIMING>(rc, 4);
        if (DBG && dbg && step == 0) dump_tiles<32>(dbg, DBG_X4, x4, tid);
        load_bias<4>(x6, lds_bias + BIAS_OFF[5], g);
        run_phase<32, 4, 4, 4, false, SP>(x4, x6, ring, lds_bias + BIAS_OFF[4],
                                          lds_mask + MASK_BASE[2] * WG_THREADS, ap, SP_SLOT_CHUNK[2], g);
        tick<TIMING>(rc, 5);
        act_tiles<4, SP>(x6, m6, ap, SP_SLOT_X6);
        if (DBG && dbg && step == 0) dump_tiles<4>(dbg, DBG_X6, x6, tid);

        // ---------------- lin6 (64 -> 1) + output ReLU  (net_modules.py:64-69)
        f32x4 w6[4];
        load_bias<4>(w6, lds_bias + W6_OFF, g);
        float part = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) part = fmaf(w6[t][r], x6[t][r], part);
        }
        part += __shfl_xor(part, 16);
        part += __shfl_xor(part, 32);
        const float z7 = part + lds_bias[BIAS_OFF[6]];
        float gz7;
        if constexpr (SP) {
            dval = act_softplus<PNDF_SP_FORM_OUT>(z7, ap.k, gz7) + poison;      // output Softplus, net_modules.py:39-41,69; NaN / inf poses: joint_axis_norms
            gz7 += poison;
        } else {
            dval = (z7 != z7) ? z7 : fmaxf(z7, 0.f);   // (relu(NaN) = NaN as in PyTorch; v_max alone returns 0)                       // output ReLU for relu AND lrelu, net_modules.py:30-37
            gz7 = (z7 > 0.f) ? 1.f : 0.f;
        }
        if (DBG && dbg && step == 0) dbg[(size_t)DBG_D * WG_THREADS + tid] = dval;
        if (args.mode == MODE_FORWARD) break;

        // ---------------- trunk backward: d d / d x, masks from the forward pass
        if (args.mode == MODE_FORWARD_GRAD && args.grad_out) {
            long long pidx = pose0 + wp;
            if (pidx >= args.B) pidx = args.B - 1;
            gz7 *= args.grad_out[pidx];
        }
        f32x4 g6[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) g6[t] = w6[t] * gz7;
        dact_tiles<4, SP>(g6, m6, ap, SP_SLOT_X6);
        tick<TIMING>(rc, 6);
        {
            f32x4 g0[8];
            {
                f32x4 g2[32];
                {
                    f32x4 g4[32];
#pragma unroll
                    for (int t = 0; t < 32; ++t) g4[t] = f32x4{0.f, 0.f, 0.f, 0.f};
                    run_phase<4, 4, 4, 32, true, SP>(g6, g4, ring, nullptr, lds_mask + MASK_BASE[2] * WG_THREADS, ap,
                                                     SP_SLOT_CHUNK[2], g);
                    dact_tiles<32, SP>(g4, m4, ap, SP_SLOT_X4);
                    tick<TIMING>(rc, 7);
                    if (DBG && dbg && step == 0) dump_tiles<32>(dbg, DBG_G4, g4, tid);
#pragma unroll
                    for (int t = 0; t < 32; ++t) g2[t] = f32x4{0.f, 0.f, 0.f, 0.f};
                    run_phase<32, 2, 32, 32, true, SP>(g4, g2, ring, nullptr, lds_mask + MASK_BASE[1] * WG_THREADS, ap,
                                                       SP_SLOT_CHUNK[1], g);
                }
                dact_tiles<32, SP>(g2, m2, ap, SP_SLOT_X2);
                tick<TIMING>(rc, 8);
                if (DBG && dbg && step == 0) dump_tiles<32>(dbg, DBG_G2, g2, tid);
#pragma unroll
                for (int t = 0; t < 8; ++t) g0[t] = f32x4{0.f, 0.f, 0.f, 0.f};
                run_phase<32, 2, 8, 8, true, SP>(g2, g0, ring, nullptr, lds_mask + MASK_BASE[0] * WG_THREADS, ap,
                                                 SP_SLOT_CHUNK[0], g);
            }
            if (DBG && dbg && step == 0) dump_tiles<8>(dbg, DBG_G0, g0, tid);
            // d d / d feature back to the per-pose buffer: lane (g, p) owns features 16 t + 4 g + r
#pragma unroll
            for (int t = 0; t < 8; ++t) *(f32x4*)(my_f + 16 * t + 4 * g) = g0[t];
        }
        // The chunk masks (aliased by GN) are dead for every wave only after all waves have left the last
        // backward phase; GN/F of a pose are touched by its own wave only.
        __syncthreads();

        tick<TIMING>(rc, 9);
        // ---------------- encoder backward + normalise backward + update
        if (args.noenc) ring_skip_encoder_section(ring);     // d d / d n is already where my_gn expects it (my_gn aliases my_f)
        else encoder_backward<SP>(my_f, my_gn, eb, ring, ap, g);
        if (DBG && dbg && step == 0) {
            for (int i = 0; i < NQ; ++i) dbg[(size_t)(DBG_GN + i) * WG_THREADS + tid] = my_gn[i];
        }
        tick<TIMING>(rc, 10);
        {
            // backward of x / clamp_min(||x||_joints, eps)  (model/posendf.py:71)
            float ss[4], dot[4], denom[4], kk[4];
            ss[0] = ss[1] = ss[2] = ss[3] = 0.f;
            dot[0] = dot[1] = dot[2] = dot[3] = 0.f;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const f32x4 qv = *(const f32x4*)(my_q + 4 * j);
                const f32x4 gv = *(const f32x4*)(my_gn + 4 * j);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    ss[c] = fmaf(qv[c], qv[c], ss[c]);
                    dot[c] = fmaf(gv[c], qv[c], dot[c]);
                }
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float norm = sqrtf(ss[c]);
                denom[c] = fmaxf(norm, 1e-12f);
                kk[c] = (norm > 1e-12f) ? dot[c] / (denom[c] * denom[c] * norm) : 0.f;
            }
            // lane group g handles joints g, g+4, ...; results replace the pose tile in LDS
            for (int j = g; j < NJ; j += 4) {
                const f32x4 qv = *(const f32x4*)(my_q + 4 * j);
                const f32x4 gv = *(const f32x4*)(my_gn + 4 * j);
                f32x4 o;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float dq = gv[c] / denom[c] - qv[c] * kk[c];
                    if (DBG && dbg && step == 0) dbg[(size_t)(DBG_DQ + 4 * j + c) * WG_THREADS + tid] = dq;
                    // q <- q - d * grad  (experiments/sample_poses.py:74: product rounded, then subtracted)
                    o[c] = (args.mode == MODE_PROJECT) ? project_update(qv[c], dval, dq) : dq;
                }
                *(f32x4*)(my_q + 4 * j) = o;
            }
        }
        // GN (aliasing the chunk masks) must be consumed by every wave before the next step's masks land
        __syncthreads();
        tick<TIMING>(rc, 11);
    }
    if constexpr (TIMING) {
        if (lane == 0 && args.dbg) {
            unsigned long long* out = (unsigned long long*)args.dbg + ((size_t)blockIdx.x * 4 + wave) * (TIMING_REGIONS + TIMING_GROUPS + TIMING_RING);
#pragma unroll
            for (int i = 0; i < TIMING_REGIONS; ++i) out[i] = rc.acc[i];
#pragma unroll
            for (int i = 0; i < TIMING_GROUPS; ++i) out[TIMING_REGIONS + i] = rc.grp[i];
            ring_stamps_out(ring, out + TIMING_REGIONS + TIMING_GROUPS);
        }
    }

    // ---------------- write back (coalesced)
    {
        long long pidx = pose0 + wp;
        if (g == 0 && pidx < args.B && args.d_out) args.d_out[pidx] = dval;
    }
    if (args.mode != MODE_FORWARD) {
        __syncthreads();
        long long nvalid = args.B - pose0;
        if (nvalid > WG_POSES) nvalid = WG_POSES;
        f32x4* dst = (f32x4*)(args.q_out + pose0 * NQ);
        const int nvec = (int)nvalid * (NQ / 4);
        for (int i = tid; i < nvec; i += WG_THREADS) dst[i] = ((const f32x4*)lds_q)[i];
    }
    // drain the DMA prefetch that is still in flight before the workgroup's LDS is released
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    }   // block loop
}

#if defined(PNDF_TIMING_TU)
// relu-family kernel with s_memtime region stamps (performance analysis only; its own translation unit, pndf_kernel_timing.hip,
// with the ring's sampled event stamps compiled in -- pndf_device.h PNDF_RING_STAMPS -- and part of the DEBUG library only)
extern "C" __global__ void __launch_bounds__(WG_THREADS, 1)
pndf_fused_relu_kernel_timing(PndfKernelArgs args) {
    pndf_fused_body<false, false, true>(args);
}
extern "C" int pndf_kernel_timing_regions() { return TIMING_REGIONS + TIMING_GROUPS + TIMING_RING; }
extern "C" int pndf_kernel_timing_layout(int what) { return what == 0 ? TIMING_REGIONS : what == 1 ? TIMING_GROUPS : what == 2 ? TIMING_RING : what == 3 ? (int)RING_STAMP_PERIOD : RING_SLOTS; }
#elif defined(PNDF_DBG_TU)
// relu-family kernel with per-stage register dumps from workgroup 0 (tests / bring-up only; pndf_kernel_dbg.hip, DEBUG library)
extern "C" __global__ void __launch_bounds__(WG_THREADS, 1)
pndf_fused_relu_kernel_dbg(PndfKernelArgs args) {
    pndf_fused_body<true, false>(args);
}
extern "C" int pndf_kernel_dbg_floats() { return DBG_TOTAL * WG_THREADS; }
#else
extern "C" __global__ void __launch_bounds__(WG_THREADS, 1)
pndf_fused_relu_kernel(PndfKernelArgs args) {
    pndf_fused_body<false, false>(args);
}

// Softplus(beta) variant (the reference's published checkpoints): fp32 derivatives go through `scratch`.
extern "C" __global__ void __launch_bounds__(WG_THREADS, 1)
pndf_fused_softplus_kernel(PndfKernelArgs args) {
    pndf_fused_body<false, true>(args);
}

extern "C" int pndf_kernel_lds_bytes() { return LDS_TOTAL; }
extern "C" long long pndf_kernel_softplus_scratch_floats_per_wg() { return SP_WG_FLOATS; }
#endif

#ifndef PNDF_TU_TAG
#define PNDF_TU_TAG fp32
#endif
PNDF_EXPORT_EXPERIMENT_WORD(PNDF_TU_TAG)
