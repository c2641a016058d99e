"""The measurement aids behind bench.py's `regions` / `box` blocks (include/posendf_amd.h: pndf_debug_project_timing,
pndf_debug_timing_layout, pndf_debug_mem_probe): the instrumented kernels must compute what the product kernels compute,
and their stamps must add up."""
import ctypes

import numpy as np
import pytest

from posendf_amd import engine, synth


def test_timing_layout_is_consistent():
    lib = engine.load_library()
    nreg, ngrp, nring, period, slots = (lib.pndf_debug_timing_layout(i) for i in range(5))
    assert nreg == len(engine.Engine.REGION_NAMES) and nring == 4 and period > 0
    assert lib.pndf_debug_timing_regions() == nreg + ngrp + nring
    assert slots == 5      # the product ring: a slot is fetched four slots ahead (csrc/pndf_device.h)


@pytest.mark.gpu
@pytest.mark.parametrize("precision,act", [("f16x3", "lrelu"), ("fp32", "lrelu"), ("f16x3", "softplus")])
def test_instrumented_kernel_projects_like_the_product_kernel(precision, act):
    import torch
    eng = engine.Engine(act, device=0, precision=precision)
    eng.load_weights(synth.make_weights(0, 2.0, 0.1))
    q = torch.from_numpy(synth.make_poses(640, seed=3)).cuda()
    ref = torch.empty_like(q)
    eng.project(q.data_ptr(), ref.data_ptr(), None, 640, 4, torch.cuda.current_stream().cuda_stream)
    out = torch.empty_like(q)
    r = eng.project_timing(q, steps=4, out=out)
    torch.cuda.synchronize()
    assert torch.equal(out, ref)                                   # same arithmetic, bit for bit
    assert set(r["regions"]) == set(engine.Engine.REGION_NAMES)
    assert all(c > 0 for c in r["regions"].values())
    assert abs(sum(r["regions"].values()) - r["cycles_per_wave_step"]) < 1e-6 * r["cycles_per_wave_step"]
    ring = r["ring"]
    # 670 slots per step, every 16th sampled: 4 steps -> 167 or 168 samples per wave
    assert 4 * 670 / ring["sampled_every"] - 2 <= ring["sampled_slots_per_wave"] <= 4 * 670 / ring["sampled_every"] + 2
    assert ring["look_ahead_slots"] == 4
    assert 0 < ring["stamp_floor_cycles"] < 400
    assert ring["wait_cycles_per_slot"] >= 0.5 * ring["stamp_floor_cycles"]
    assert 0.5 < r["effective_sclk_ghz"] < 3.0


@pytest.mark.gpu
def test_mem_probe_orders_the_levels():
    lib = engine.load_library()
    out = (ctypes.c_double * 8)()
    assert lib.pndf_debug_mem_probe(0, out, 8) == 0
    l2, mall, hbm, gbps, mhz, hops, ring_gbps, ring_ns = list(out)
    assert 50 < l2 < mall * 1.05 and mall < hbm * 1.05 and hbm < 20000, (l2, mall, hbm)
    assert 500 < gbps < 9000 and mhz > 0 and hops == 4096
    # the ring alone must deliver more than the f16x3 kernel consumes (~51 GB/s per CU), or that kernel is fetch-bound here
    assert 51 < ring_gbps < 1000 and abs(ring_ns * ring_gbps - 16384) < 1.0, (ring_gbps, ring_ns)
    assert lib.pndf_debug_mem_probe(0, out, 5) < 0                 # too small an output buffer is refused
