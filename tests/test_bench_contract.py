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


DIAGNOSTIC_BLOCKS = {"box", "regions", "f16_single", "fp16_checkpoint", "host_twin", "host_boundary", "motion_denoise_config4"}


@pytest.mark.gpu
def test_bench_json_line_default_precision():
    """The DEFAULT line (what the driver runs): warm-up + timed loop + fp32_exact + softplus (with its own roofline block) +
    gpu_torch_baseline + cpu_baseline + parity_sample -- and none of the analysis blocks."""
    d = run_bench("--cpu-budget", "2")
    assert REQUIRED <= set(d) and ROOFLINE <= set(d["roofline"])
    assert not (DIAGNOSTIC_BLOCKS & set(d)), DIAGNOSTIC_BLOCKS & set(d)
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert d["value"] > 0 and d["unit"] == "poses/s" and "workload" in d["config"]
    # the line names its precisions honestly, within what the driver's record keeps of a string
    head = d["config"]["workload"][:120]
    assert "f16x3 (fp32-split) timed" in head and "fp32 (exact), bf16 (not parity grade) reported" in head
    assert len("cfg[2] B=65536/GPU x100 steps lrelu: f16x3 (fp32-split) timed; fp32 (exact), bf16 (not parity grade) reported") <= 120
    # BASELINE.json configs[2] "fp32 vs bf16": both halves in the default line, the bf16 one with its error against the fp64 oracle
    bf = d["bf16"]
    assert bf["kernel"] == "pndf_fused_bf16_relu_kernel" and bf["kernel_ms"] > 0
    assert bf["parity_sample"]["median"] > 10 * d["parity_sample"]["median"] and bf["parity_sample"]["within_tolerance_frac"] < 0.5
    assert d["roofline"]["bf16_poses_per_s"] == bf["poses_per_s_per_gpu"] and d["roofline"]["bf16_median_rel_err_vs_fp64"] == bf["parity_sample"]["median"]
    assert d["roofline"]["bound"] == "mfma" and 0 < d["roofline"]["frac"] < 1
    assert abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 1e-9
    cb = d["cpu_baseline"]
    assert {"value", "unit", "cores", "kind", "sample"} <= set(cb) and cb["kind"] == "port" and cb["value"] > 0
    assert {"cpu", "runs_s", "config0_forward_only"} <= set(cb) and cb["config0_forward_only"]["poses_per_s"] > 0
    # B = 4,096 on ALL physical cores; the calibration of smaller thread counts sits beside the figure, it does
    # not choose it
    assert cb["batch"] == 4096 and "B=4096" in cb["sample"] and str(cb["cores"]) in cb["thread_calibration_pose_steps_per_s"]
    assert f"{cb['cores']} threads = all physical cores" in cb["sample"] and len(cb["runs_s"]) in (1, 3)      # (1: a host too contended for three)
    assert cb["cores"] == max(int(k) for k in cb["thread_calibration_pose_steps_per_s"])
    if cb["cgroup_cpu_quota"]:              # the container's CFS quota caps the thread count (threads beyond it are throttled, not run)
        assert cb["cores"] <= cb["cgroup_cpu_quota"]
    assert cb["pinned_to"].startswith(f"{cb['cores']} distinct physical cores") and len(cb["host_loadavg_before_after"]) == 2
    assert {"sclk_mhz", "package_w", "telemetry_source"} <= set(d["roofline"])      # clock / power beside the time
    if d["roofline"]["sclk_mhz"] is not None:
        assert 300 < d["roofline"]["sclk_mhz"] < 3000 and 50 < d["roofline"]["package_w"] < 2000
    assert "fp32_exact" in d and "forward_grad_single_launch" in d
    # the exact-fp32 figure is first-class AND a scalar of `roofline` (the driver's record keeps the scalars of that object)
    assert d["roofline"]["fp32_exact_poses_per_s"] == d["fp32_exact"]["poses_per_s_per_gpu"] > 0
    assert 0 < d["roofline"]["fp32_exact_frac_of_fp32_mfma_peak"] < 1
    assert d["softplus"]["kernel"] == "pndf_fused_split_softplus_kernel" and d["softplus"]["kernel_ms"] > 0
    gts = d["softplus"]["gpu_torch_baseline"]     # the >= 10x denominator for the activation the reference's scripts load
    assert gts["value"] > 0 and abs(gts["speedup_of_softplus_kernel"] - d["softplus"]["poses_per_s_per_gpu"] / gts["value"]) < 1e-9
    rs = d["roofline_softplus"]                   # softplus is a first-class line
    assert ROOFLINE <= set(rs) and rs["kernel"] == "pndf_fused_split_softplus_kernel" and rs["kernel_ms"] == d["softplus"]["kernel_ms"]
    assert abs(rs["frac"] - rs["achieved"] / rs["peak"]) < 1e-12 and rs["algorithmic_bytes_per_launch"] == 4096 * 676 + 10720 * 1024
    assert d["roofline"]["softplus_kernel_ms"] == rs["kernel_ms"] and d["roofline"]["softplus_frac"] == rs["frac"]
    assert d["roofline"]["kernel"] == "pndf_fused_split_relu_kernel"
    assert d["roofline"]["kernel_ms_median"] > 0
    ps = d["parity_sample"]                       # the line checks what it timed (5 steps here)
    assert ps["median"] < 1e-5 and ps["within_tolerance_frac"] > 0.9
    assert d["roofline"]["traffic"] is None and d["roofline"]["traffic_stale"] is None      # only quoted for the profiled workload
    gt = d["gpu_torch_baseline"]                  # the denominator of north_star's ">= 10x", measured in the same run
    assert gt["value"] > 0 and abs(gt["speedup_of_value"] - d["value"] / gt["value"]) < 1e-9
    assert d["roofline"]["gpu_torch_poses_per_s"] == gt["value"] and d["roofline"]["speedup_vs_gpu_torch"] == gt["speedup_of_value"]


@pytest.mark.gpu
def test_bench_diagnostics_blocks():
    """`--diagnostics` adds the analysis blocks of rounds 3 - 5 to the same line (they are not part of the default run)."""
    d = run_bench("--diagnostics", "--no-cpu-baseline")
    assert DIAGNOSTIC_BLOCKS <= set(d), DIAGNOSTIC_BLOCKS - set(d)
    assert "f16_single" in d
    h16 = d["fp16_checkpoint"]                    # half-precision checkpoint: two-term kernels, a side block, never `value`
    # (which kernel ran is the contract here; that it is FASTER is a full-size statement, checked on the committed full-size
    # line below -- at this test's 1 ms launches jitter decides the order)
    assert h16["kernel"] == "pndf_fused_split2_relu_kernel" and h16["kernel_ms"] > 0
    hb = d["host_boundary"]                       # PCIe-inclusive rate of a host-tensor caller: reported, never `value`
    assert hb["ms"] > 0 and hb["poses_per_s"] > 0 and d["host_twin"]["value"] > 0
    md = d["motion_denoise_config4"]              # configs[4] on one GPU's share with the reference's objective: a side block
    assert md["finite"] and md["fused_adam_step_ms"] > 0 and 0 < md["body_model_pass"]["frac"] < 1
    # the line describes its box and says where a step's cycles went on it
    box, reg = d["box"], d["regions"]
    assert box["compute_units"] > 0 and {"device", "arch", "pci", "host_loadavg", "mem_probe", "power_window"} <= set(box)
    mp = box["mem_probe"]
    assert 50 < mp["l2_hit_latency_ns"] < mp["hbm_latency_ns"] * 1.05 and 500 < mp["stream_read_gbps"] < 9000
    pw = box["power_window"]                      # firmware accumulators; a box without amd-smi reports the error instead
    assert "error" in pw or (pw["mean_package_w"] > 100 and pw["energy_j_per_launch"] > 0 and 0 <= pw["ppt_limited_frac"] <= 1)
    assert "error" in pw or pw["device_bdf"] == box["pci"]      # the readings come from the device that ran the kernel
    assert reg["kernel"] == "pndf_fused_split_relu_kernel_timing" and len(reg["regions"]) == 12
    assert reg["cycles_per_wave_step"] > 1e5 and reg["ring"]["look_ahead_slots"] == 4
    assert reg["ring"]["wait_cycles_per_slot"] > 0 and reg["ring"]["barrier_cycles_per_slot"] > 0
    assert reg["fp32_kernel"]["cycles_per_wave_step"] > reg["cycles_per_wave_step"]


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


def _two_rank_env(port):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(PNDF_BENCH_BACKEND="gloo", PNDF_BENCH_SHARE_DEVICE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return env


def _one_json_line(out):
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines                      # ONE line although two processes ran
    return json.loads(lines[0])


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4, 8])
def test_bench_real_ranks_on_one_device(world):
    """The N > 1 path with REAL processes and the HIP kernel, for every world size of the driver's scaling run (N = 2, 4, 8;
    VERDICT r3 item 3, r5 item 7): `bench.py --gpus N` launches its own N ranks under torch.distributed.run; all share the
    one visible device (PNDF_BENCH_SHARE_DEVICE=1) and meet over gloo (RCCL refuses two ranks on one device).  Exercised:
    self-launch, rank -> shard offset, ONE final collective of 85 floats per pose into the preallocated buffer with every
    rank checking its own block in it, max-over-ranks timing, whole-job value, one JSON line, and the per-rank records that
    prove the rank count."""
    B = 1024
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1", "--batch", str(B),
           "--proj-steps", "5"]
    d = _one_json_line(subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=REPO, env=_two_rank_env(29591 + world)))
    assert d["n_gpus"] == world and d["scaling"] == "weak" and d["config"]["global_batch"] == world * B
    di = d["distributed"]
    assert di["world_size"] == world and di["backend"] == "gloo" and di["ranks_share_devices"] is True
    assert di["gathered_rows"] == world * B and di["floats_per_gathered_row"] == 85 and di["collectives_per_pass"] == 1
    assert di["ranks_seen"] == world and "rccl_version" in di
    ranks = di["per_rank"]
    assert [r["rank"] for r in ranks] == list(range(world)) and len({r["pid"] for r in ranks}) == world
    assert all(r["own_block_in_gather"] and r["kernel_ms"] > 0 and r["rows"] == B for r in ranks)
    # whole-job aggregate over the slowest rank's clock
    slowest = max(r["elapsed_s"] for r in ranks)
    assert abs(d["value"] - world * B * 2 / slowest) < 1e-6 * d["value"]
    assert abs(d["ms_per_step"] - slowest / 2 * 1e3) < 1e-6 * d["ms_per_step"]
    assert "fp32_exact" not in d and "cpu_baseline" not in d        # a scaling run is warm-up + timed passes only


@pytest.mark.gpu
def test_bench_denoise_workload_two_ranks():
    """BASELINE.json configs[4] has an N-rank line too: `--workload denoise` shards whole sequences over the ranks
    (experiments/motion_denoise.py:171-188 loops sequences), runs fused Adam steps of the reference's objective and gathers
    the denoised poses once.  One rank first (the N = 1 line), then two real ranks on the one device."""
    base = [sys.executable, os.path.join(REPO, "bench.py"), "--workload", "denoise", "--steps", "2", "--warmup", "1", "--seqs", "3",
            "--frames", "24", "--adam-steps", "4"]
    env1 = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    d1 = _one_json_line(subprocess.run(base + ["--gpus", "1"], capture_output=True, text=True, timeout=900, cwd=REPO, env=env1))
    assert d1["n_gpus"] == 1 and d1["unit"] == "frame-steps/s" and d1["finite"] and d1["value"] > 0
    assert "configs[4]" in d1["config"]["workload"] and d1["config"]["sequences_total"] == 3 and "distributed" not in d1
    d2 = _one_json_line(subprocess.run(base + ["--gpus", "2"], capture_output=True, text=True, timeout=900, cwd=REPO, env=_two_rank_env(29593)))
    assert d2["n_gpus"] == 2 and d2["config"]["sequences_total"] == 6 and d2["finite"]
    di = d2["distributed"]
    assert di["world_size"] == 2 and di["gathered_rows"] == 6 and all(r["own_block_in_gather"] for r in di["per_rank"])
    slowest = max(r["elapsed_s"] for r in di["per_rank"])
    assert abs(d2["value"] - 6 * 24 * 4 * 2 / slowest) < 1e-6 * d2["value"]


def test_full_size_relations_in_the_committed_bench_line():
    """What the small-size contract test above cannot assert without absorbing launch jitter is asserted where it
    is meaningful: on the newest full-size line committed under profiles/ (B = 65,536 x 100 steps, ~90 ms launches)."""
    import glob
    heads = sorted(glob.glob(os.path.join(REPO, "profiles", "r*", "bench_head.json")))
    assert heads, "no committed full-size bench line"
    with open(heads[-1]) as f:
        d = json.loads([l for l in f.read().splitlines() if l.strip().startswith("{")][-1])
    k = d["roofline"]["kernel_ms"]
    assert d["config"]["global_batch"] == 65536 and 50 < k < 200
    assert d["fp16_checkpoint"]["kernel_ms"] < 0.9 * k                 # two terms instead of three
    assert d["f16_single"]["kernel_ms"] < d["fp16_checkpoint"]["kernel_ms"] < d["softplus"]["kernel_ms"] < d["fp32_exact"]["kernel_ms"]
    assert abs(d["ms_per_step"] - k) < 0.02 * k                         # one launch per harness step, nothing else in the timed region
    assert abs(d["value"] - 65536 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    if "regions" in d:      # (lines committed since round 5) the ring's own numbers: on a healthy box the look-ahead hides the fetches
        ring = d["regions"]["ring"]
        assert ring["wait_cycles_per_slot"] < 40 and ring["barrier_cycles_per_slot"] < 80, ring
        assert d["regions"]["cycles_per_wave_step"] < 1.5 * 404e3
        pw = d["box"]["power_window"]
        assert "error" in pw or 90 < pw["energy_j_per_launch"] < 150


def test_diagnose_box_reads_the_committed_lines(capsys):
    """tools/diagnose_box.py turns a line's `box` / `regions` blocks into a verdict: the profile of record of round 5 ran on
    a chip that needs 125 J per launch (power-limited 97 % of the time, everything else in range); a line without the blocks
    (the driver's r04 line) is reported as undiagnosable rather than guessed at."""
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import diagnose_box
    old = sys.argv
    try:
        sys.argv = ["diagnose_box.py", os.path.join(REPO, "profiles", "r05", "bench_head.json"), os.path.join(REPO, "BENCH_r04.json")]
        if not os.path.exists(sys.argv[2]):
            sys.argv.pop()
        diagnose_box.main()
    finally:
        sys.argv = old
    out = capsys.readouterr().out
    assert "a power-limited chip that spends 125 J" in out and "weight ring: 11.3 cycles per slot in the DMA wait (within" in out
    if os.path.exists(os.path.join(REPO, "BENCH_r04.json")):
        assert "nothing to diagnose from" in out
