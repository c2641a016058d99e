"""bench.py prints ONE JSON line with the fields the driver and the judge read (small sizes here)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import REPO

REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline"}
ROOFLINE = {"bound", "achieved", "peak", "unit", "frac", "traffic"}


def run_bench(*extra):
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "4096",
           "--proj-steps", "5", *extra]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=REPO)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_json_line_default_precision():
    d = run_bench("--cpu-budget", "2")
    assert REQUIRED <= set(d) and ROOFLINE <= set(d["roofline"])
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert d["value"] > 0 and d["unit"] == "poses/s" and "workload" in d["config"]
    assert d["roofline"]["bound"] == "mfma" and 0 < d["roofline"]["frac"] < 1
    assert abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 1e-9
    cb = d["cpu_baseline"]
    assert {"value", "unit", "cores", "kind", "sample"} <= set(cb) and cb["kind"] == "port" and cb["value"] > 0
    assert {"cpu", "runs_s", "config0_forward_only"} <= set(cb) and cb["config0_forward_only"]["poses_per_s"] > 0
    assert "fp32_exact" in d and "f16_single" in d and "forward_grad_single_launch" in d
    assert d["softplus"]["kernel"] == "pndf_fused_split_softplus_kernel" and d["softplus"]["kernel_ms"] > 0
    h16 = d["fp16_checkpoint"]                    # half-precision checkpoint: two-term kernels, a side block, never `value`
    assert h16["kernel"] == "pndf_fused_split2_relu_kernel" and 0 < h16["kernel_ms"] < 1.25 * d["roofline"]["kernel_ms"]      # (1 ms launches: jitter)
    assert d["roofline"]["kernel"] == "pndf_fused_split_relu_kernel"
    hb = d["host_boundary"]                       # PCIe-inclusive rate of a host-tensor caller: reported, never `value`
    # (at this test's tiny size -- 4,096 poses x 5 steps, ~1 ms -- launch jitter is of the order of the PCIe copies: the
    # full-size relation hb.ms > kernel_ms holds in profiles/*/bench_head.json, here only its order of magnitude is checked)
    assert hb["ms"] > 0.5 * d["roofline"]["kernel_ms"] and 0 < hb["poses_per_s"] < d["value"] * 2
    assert d["roofline"]["kernel_ms_median"] > 0
    md = d["motion_denoise_config4"]              # configs[4] on one GPU's share with the reference's objective: a side block
    assert md["finite"] and md["fused_adam_step_ms"] > 0 and 0 < md["body_model_pass"]["frac"] < 1
    ps = d["parity_sample"]                       # the line checks what it timed (5 steps here)
    assert ps["median"] < 1e-5 and ps["within_tolerance_frac"] > 0.9
    assert d["roofline"]["traffic"] is None and d["roofline"]["traffic_stale"] is None      # only quoted for the profiled workload
    gt = d["gpu_torch_baseline"]                  # the denominator of north_star's ">= 10x", measured in the same run
    assert gt["value"] > 0 and abs(gt["speedup_of_value"] - d["value"] / gt["value"]) < 1e-9


@pytest.mark.gpu
def test_bench_json_line_fp32_and_softplus():
    d = run_bench("--precision", "fp32", "--no-cpu-baseline", "--no-fp32-ref", "--no-gpu-torch-baseline")
    assert d["dtype"] == "f32" and d["roofline"]["peak"] == 157.3 and "cpu_baseline" not in d
    d = run_bench("--act", "softplus", "--no-cpu-baseline", "--no-fp32-ref", "--no-gpu-torch-baseline")
    assert d["roofline"]["kernel"] == "pndf_fused_split_softplus_kernel"


@pytest.mark.gpu
def test_bench_collective_path_single_rank():
    """The N > 1 code path (process group on RCCL, barriers, all_gather_into_tensor into the preallocated buffer,
    max-over-ranks reduction) with world size 1 -- all a one-GPU box can exercise of it."""
    env = dict(os.environ, PNDF_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29571", RANK="0",
               LOCAL_RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "4096",
           "--proj-steps", "5", "--no-cpu-baseline", "--no-fp32-ref", "--no-gpu-torch-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=REPO, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.strip()][-1])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["scaling"] == "weak"


def test_bench_refuses_more_gpus_than_visible():
    """`python bench.py --gpus N` launches its own N ranks (torch.distributed.run); with fewer visible devices it must
    fail loudly instead of printing an n_gpus: 1 line for a job that was asked to be N = 8."""
    import torch
    n = torch.cuda.device_count() + 1 if torch.cuda.is_available() else 2
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300, cwd=REPO,
                         env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    assert out.returncode != 0 and "visible" in out.stderr and not out.stdout.strip()
