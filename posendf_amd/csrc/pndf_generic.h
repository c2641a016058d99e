// Host interface of the runtime-planned DFNet path (pndf_generic.hip) for pndf_capi.hip.
//
// The fused kernels of pndf_kernel*.hip are laid out at compile time for the one architecture the reference ships
// (configs/amass.yaml: 126 | 84 -> 256 -> 512 -> 1024 -> 512 -> 256 -> 64 -> 1).  The reference's DFNet takes its hidden
// widths as a free list (`dims`, model/network/net_modules.py:14-28), and its README points users at checkpoints of other
// configs; every such network -- 2 .. 8 linear layers, hidden widths 1 .. 1024 -- runs here: one persistent launch per call
// as well, the same encoder, normalisation and update code, the trunk layer by layer from a plan built at pndf_create.
#pragma once
#include <stdint.h>

#include <string>

#include "../../include/posendf_amd.h"

struct PndfGeneric;      // plan + device buffers of one engine

// does `cfg` need this path?  (false: the amass.yaml-shaped kernels take it; unsupported configurations are refused by
// pndf_create before this is asked)
bool pndf_generic_needed(const pndf_config& cfg);
// 0 or a negative pndf_status with `err` set
int pndf_generic_create(PndfGeneric** out, const pndf_config& cfg, int resident_wgs, std::string& err);
void pndf_generic_destroy(PndfGeneric* g);
// tensors in state-dict order: 84 encoder tensors (with the encoder) + 2 per linear layer
int pndf_generic_load(PndfGeneric* g, const float* const* tensors, const int64_t* numel, int n_tensors, std::string& err);
// mode: MODE_FORWARD / MODE_FORWARD_GRAD / MODE_PROJECT (pndf_args.h); enqueues ONE kernel on `stream`
int pndf_generic_launch(PndfGeneric* g, int mode, const float* q, const float* gout, float* qo, float* d, int64_t B, int steps,
                        void* stream, std::string& err);
const char* pndf_generic_kernel_name(const PndfGeneric* g);
