#!/usr/bin/env python3
"""Headline benchmark: projected poses/sec (100 gradient steps, 21-joint quaternion poses) on N MI355X.

One "step" of this harness = one pass of the hot path over one batch: PoseNDF.project(q0, steps=100) on
B = 65,536 synthetic poses per GPU (BASELINE.json configs[2]; the batch is sharded by rank with no data-path
collective during the 100 steps; for N > 1 the projected poses are all-gathered over RCCL at the end of each
pass, inside the timed region).  Inputs are resident in HBM before the timed region starts.

Launch:  python bench.py --gpus 1 --steps 5 --warmup 1
         python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

FLOP_PER_POSE_STEP = 5_450_416      # 2 x 1,362,604 MACs forward + the same for d d/d q
PEAK_FP32_MFMA_TFLOPS = 157.3       # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense
PEAK_F16_MFMA_TFLOPS = 2500.0       # MI355X_MICROARCH.md: dense fp16/bf16 MFMA (not the 2:1-sparse figure)
# What dense fp16 MFMA SUSTAINS on this chip under its power limit, with random operand data, one wave per SIMD, registers
# only (tools/ubench/mfma_power.hip, profiles/r02/mfma_power.txt).  It depends on the issue pattern: 1,700 TFLOP/s when the
# MFMAs rotate over eight accumulators, 2,200 on a single accumulator chain, 2,000 in the pattern the split kernel issues
# (chains of three on one accumulator, the weight operand kept for two MFMAs); 2,245-2,484 with constant operands.  The
# split kernel is power-bound, so this -- not the 2.5 PFLOP/s datasheet peak -- is the wall its
# ISSUED rate (3 MFMAs per product block) runs into.
SUSTAINED_F16_MFMA_TFLOPS = 2000.0
KERNELS = {"fp32": ("pndf_fused_relu_kernel", PEAK_FP32_MFMA_TFLOPS, "f32"),
           "f16x3": ("pndf_fused_split_relu_kernel", PEAK_F16_MFMA_TFLOPS,
                     "f16x3 (fp32 operands split into fp16 hi+lo, 3 MFMAs per product block, fp32 accumulate)"),
           "f16": ("pndf_fused_half_relu_kernel", PEAK_F16_MFMA_TFLOPS,
                   "f16 (operands ROUNDED to fp16, fp32 accumulate; NOT within the 1e-4 parity bar)"),
           "bf16": ("pndf_fused_bf16_relu_kernel", PEAK_F16_MFMA_TFLOPS,
                    "bf16 (operands ROUNDED to bfloat16, fp32 accumulate; NOT within the 1e-4 parity bar)")}


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def _getaffinity(tid=0):
    try:
        return os.sched_getaffinity(tid)
    except AttributeError:      # (platform without the call: the baseline then runs unpinned)
        return set(range(os.cpu_count() or 1))


def _physical_cores():
    """One hardware thread per physical core among the CPUs this process may run on (SMT siblings share a core's FPUs:
    two torch threads on one core are one thread's worth of matmul), in CPU order (= socket by socket)."""
    allowed = sorted(_getaffinity(0))
    seen, picked = set(), []
    for c in allowed:
        try:
            with open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list") as f:
                sib = f.read().strip()
        except OSError:
            sib = str(c)
        if sib not in seen:
            seen.add(sib)
            picked.append(c)
    return picked, len(allowed)


def _cgroup_cpu_quota():
    """CPUs' worth of CFS quota of this container (cgroup v2 cpu.max / v1 cpu.cfs_quota_us), None when unlimited or unknown:
    more threads than that are throttled, whatever the affinity mask says."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            q = float(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            per = float(f.read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


def _idlest_first(cpus, window_s=0.25):
    """`cpus` ordered by how idle each was over a short window (/proc/stat), idlest first.  The GPU boxes' hosts are shared
    and every tenant starts pinning at CPU 0: a thread of an OpenMP region that shares its core with another tenant's work
    makes all the others wait at every one of the ~260 barriers of a projection step."""
    def snap():
        out = {}
        try:
            with open("/proc/stat") as f:
                for line in f:
                    if line.startswith("cpu") and line[3].isdigit():
                        v = line.split()
                        t = [float(x) for x in v[1:9]]
                        out[int(v[0][3:])] = (t[3] + t[4], sum(t))      # idle + iowait, total
        except OSError:
            pass
        return out
    a = snap()
    time.sleep(window_s)
    b = snap()

    def busy(c):
        if c not in a or c not in b or b[c][1] <= a[c][1]:
            return 0.0
        return 1.0 - (b[c][0] - a[c][0]) / (b[c][1] - a[c][1])
    return sorted(cpus, key=lambda c: (round(busy(c), 2), c))


def _pin_process(cpus):
    """CPU affinity of EVERY thread of this process (torch's intra-op pool exists already: new masks are not inherited
    by running threads).  Returns what to pass back to `_unpin_process`."""
    before = {"main": _getaffinity(0), "tasks": {}}
    if not hasattr(os, "sched_setaffinity"):
        return before
    for t in os.listdir("/proc/self/task"):
        try:
            before["tasks"][int(t)] = os.sched_getaffinity(int(t))
            os.sched_setaffinity(int(t), cpus)
        except OSError:
            pass
    return before


def _unpin_process(before):
    """Every task that exists NOW gets its old mask back; tasks born while the process was pinned (torch's intra-op and
    OpenMP workers spawned during the baseline inherit the narrowed mask) get the main thread's original one."""
    if not hasattr(os, "sched_setaffinity"):
        return
    for t in os.listdir("/proc/self/task"):
        try:
            os.sched_setaffinity(int(t), before["tasks"].get(int(t), before["main"]))
        except OSError:
            pass


def cpu_baseline(act, sd, proj_steps, budget_s=12.0, runs=3, batch=4096):
    """The reference's CPU PyTorch path (restated in oracle/posendf_torch.py) on this box's host cores, SURVEY.md 8d:
    B = 4,096 poses (fixed: matmuls large enough for the threads to have work), `torch.set_num_threads(all physical
    cores)` with the process confined to ONE hardware thread per physical core, `runs` timed projections (median
    reported).  Every projection step does identical work, so when the full 100 steps at B = 4,096 do not fit the budget
    the timed run does fewer steps and is scaled linearly -- the sample says so.  Beside it, NOT chosen by it (VERDICT r5
    item 6): a 2-step calibration of smaller thread counts (the boxes' hosts are shared: fewer threads are often faster).
    Plus BASELINE.json configs[0] (B = 256, forward only, median of 10)."""
    import statistics
    import torch
    from oracle.posendf_torch import RefNet, project
    from posendf_amd import synth
    phys, visible = _physical_cores()
    phys = _idlest_first(phys)
    quota = _cgroup_cpu_quota()
    net = RefNet(act)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    q = torch.from_numpy(synth.make_poses(batch, seed=1234))
    # all physical cores this container can actually run on: the affinity mask, capped by the cgroup's CPU quota
    threads = max(1, min(len(phys), int(quota)) if quota else len(phys))
    load_before = os.getloadavg()
    calib = {}
    threads_before = torch.get_num_threads()
    restore = _pin_process(set(phys))
    t_begin = time.perf_counter()
    try:
        _pin_process(set(phys[:threads]))
        torch.set_num_threads(threads)
        project(net, q, 1)                                      # warm-up (thread pool, allocator)
        # one projection step of the batch, to size the timed runs: the FASTER of two probes, the second four times as long
        # when the first was short (the first steps after the pool spins up can be 30 x slower than steady state)
        t0 = time.perf_counter()
        project(net, q, 1)
        step_s = time.perf_counter() - t0
        if step_s < 0.25 * budget_s / 3:
            n_probe = 4 if step_s < 0.05 * budget_s / 3 else 1
            t0 = time.perf_counter()
            project(net, q, n_probe)
            step_s = min(step_s, (time.perf_counter() - t0) / n_probe)
        calib[threads] = batch / step_s
        if step_s * 3 > budget_s * 0.8:      # a host so contended that three one-step runs do not fit: one run of one step
            runs = 1
        timed_steps = int(min(proj_steps, max(1, budget_s * 0.8 / runs / step_s)))
        times = []
        for _ in range(runs):
            t0 = time.perf_counter()
            project(net, q, timed_steps)
            times.append(time.perf_counter() - t0)
        dt = statistics.median(times) * proj_steps / timed_steps   # seconds per full projection of the batch
        # configs[0]: batch = 256, PoseNDF.forward() distance only, PyTorch CPU
        q0 = q[:256]
        with torch.no_grad():
            net(q0)
            t0 = time.perf_counter()
            net(q0)
            n0 = 10 if time.perf_counter() - t0 < 0.2 else 3
            f_t = []
            for _ in range(n0):
                t0 = time.perf_counter()
                net(q0)
                f_t.append(time.perf_counter() - t0)
        f_med = statistics.median(f_t)
        for t in sorted({t for t in (8, 16, 32, 64) if t < threads}):      # beside the figure, never its choice
            if time.perf_counter() - t_begin > 2.0 * budget_s:
                break
            _pin_process(set(phys[:t]))
            torch.set_num_threads(t)
            project(net, q, 1)
            t0 = time.perf_counter()
            project(net, q, 1)
            calib[t] = batch / (time.perf_counter() - t0)
    finally:
        torch.set_num_threads(threads_before)
        _unpin_process(restore)
    scaled = "" if timed_steps == proj_steps else f" ({timed_steps} steps timed, scaled linearly to {proj_steps}: every step does identical work)"
    return {"value": batch / dt, "unit": "projected poses/s", "cores": threads, "kind": "port", "cpu": _cpu_model(),
            "batch": batch, "timed_steps": timed_steps, "runs_s": [round(t, 3) for t in times],
            "runs_spread": round((max(times) - min(times)) / statistics.median(times), 3),
            "thread_calibration_pose_steps_per_s": {str(t): round(v, 1) for t, v in sorted(calib.items())},
            "best_calibrated_threads": max(calib, key=calib.get),
            "pinned_to": f"{threads} distinct physical cores (one hardware thread each, the idlest first) of {visible} visible hardware threads",
            "cgroup_cpu_quota": quota, "physical_cores_visible": len(phys),
            "host_loadavg_before_after": [[round(x, 1) for x in load_before], [round(x, 1) for x in os.getloadavg()]],
            "sample": f"B={batch} poses x {proj_steps} steps{scaled}, median of {runs} runs = {dt:.1f} s per projection; "
                      f"PyTorch-CPU restatement of the reference (oracle/posendf_torch.py), {threads} threads = all physical cores"
                      + (f" within the container's CPU quota of {quota:g}" if quota and quota < len(phys) else "")
                      + f", one hardware thread each, of {visible} visible hardware threads of a {_cpu_model()}",
            "config0_forward_only": {"workload": "BASELINE.json configs[0]: batch=256, forward() distance only, PyTorch CPU",
                                     "ms": f_med * 1e3, "poses_per_s": 256 / f_med, "runs": n0}}


class GpuTelemetry:
    """Shader clock and package power of the device while the timed loop runs, sampled by a host thread from the amdgpu
    hwmon files (no subprocess inside the timed region; rocm-smi --showpower --showclocks once as the fallback).  The
    split kernels are power-bound: two boxes can only be compared with these beside the time."""

    def __init__(self, index=0, period_s=0.05):
        import glob
        self.period = period_s
        self.samples = []
        self.src = None
        self.how = "unavailable"
        self._stop = None
        self._thread = None
        cands = []
        for h in sorted(glob.glob("/sys/class/drm/card[0-9]*/device/hwmon/hwmon*")):
            f = os.path.join(h, "freq1_input")
            pw = next((os.path.join(h, n) for n in ("power1_average", "power1_input") if os.path.exists(os.path.join(h, n))), None)
            if os.path.exists(f) and pw:
                pci = os.path.basename(os.path.realpath(os.path.join(h, "..", "..")))     # .../<domain:bus:dev.fn>/hwmon/hwmonN
                cands.append((pci.lower(), f, pw))
        if not cands:
            return
        # a node shows the hwmon files of ALL its GPUs, the process sees one: match the HIP device by its PCI address
        want = None
        try:
            import torch
            pr = torch.cuda.get_device_properties(index)
            want = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        except Exception:
            pass
        hit = [c for c in cands if c[0] == want]
        if hit:
            self.src = [hit[0][1:]]
            self.how = f"amdgpu hwmon of {want} ({os.path.basename(hit[0][2])}, freq1_input), median over the timed loop"
        else:      # no PCI match (container without the address): sample all, report the busiest one
            self.src = [c[1:] for c in cands]
            self.how = (f"amdgpu hwmon ({os.path.basename(cands[0][2])}, freq1_input): busiest of {len(cands)} devices visible in "
                        "sysfs, median over the timed loop")

    @staticmethod
    def _read(path):
        try:
            with open(path) as f:
                return float(f.read().strip())
        except (OSError, ValueError):
            return None

    def start(self):
        if self.src is None:
            return
        import threading
        self._stop = threading.Event()

        def loop():
            while not self._stop.is_set():
                row = []
                for fpath, ppath in self.src:
                    f, p = self._read(fpath), self._read(ppath)
                    row.append((f / 1e6, p / 1e6) if (f is not None and p is not None) else None)      # Hz -> MHz, uW -> W
                self.samples.append(row)
                self._stop.wait(self.period)
        self._thread = threading.Thread(target=loop, daemon=True)
        self._thread.start()

    def stop(self):
        if self._thread is not None:
            self._stop.set()
            self._thread.join()

    def summary(self):
        import statistics
        best = None
        for k in range(len(self.src or [])):
            col = [r[k] for r in self.samples if r[k] is not None]
            if col:
                cand = (statistics.median(c[1] for c in col), statistics.median(c[0] for c in col), len(col))
                if best is None or cand[0] > best[0]:
                    best = cand
        if best:
            return {"sclk_mhz": best[1], "package_w": best[0], "telemetry_samples": best[2], "telemetry_source": self.how}
        return {"sclk_mhz": None, "package_w": None, "telemetry_samples": 0, "telemetry_source": "unavailable"}


def box_block(dev_index, lib):
    """What the box is, so that a slow line can be told from a slow box: partition modes, clock
    tables and firmware of the device matched by PCI address, the runtime's view of it, the host's load, and the memory
    subsystem as the weight stream sees it (pndf_debug_mem_probe: dependent-load latency with the footprint in L2 /
    Infinity Cache / HBM, streaming read bandwidth).  Read once, outside the timed region."""
    import ctypes
    import glob
    import torch
    pr = torch.cuda.get_device_properties(dev_index)
    out = {"device": pr.name, "arch": getattr(pr, "gcnArchName", None), "compute_units": pr.multi_processor_count,
           "hbm_gib": round(pr.total_memory / 2**30, 1), "l2_bytes": getattr(pr, "L2_cache_size", None),
           "hip": torch.version.hip, "torch": torch.__version__,
           "host_loadavg": [round(x, 2) for x in os.getloadavg()], "host_cpus_visible": len(os.sched_getaffinity(0))}
    try:
        want = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
    except AttributeError:
        want = None
    out["pci"] = want

    def rd(path):
        try:
            with open(path) as f:
                return f.read().strip()
        except OSError:
            return None
    devdir = None
    for d in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
        if os.path.basename(os.path.realpath(d)).lower() == want:
            devdir = d
    out["sysfs_devices_visible"] = len(glob.glob("/sys/class/drm/card[0-9]*/device/pp_dpm_sclk"))
    if devdir:
        for key in ("current_compute_partition", "current_memory_partition", "vbios_version", "power_dpm_force_performance_level",
                    "mem_busy_percent", "gpu_busy_percent"):
            out[key] = rd(os.path.join(devdir, key))
        for clk in ("sclk", "mclk", "fclk", "socclk"):
            t = rd(os.path.join(devdir, "pp_dpm_" + clk))
            if t is not None:      # "0: 132Mhz\n1: 2400Mhz *": the table and the level in use
                levels = [ln.strip() for ln in t.splitlines()]
                out["pp_dpm_" + clk] = {"levels": [ln.rstrip(" *") for ln in levels], "current": next((ln.rstrip(" *") for ln in levels if ln.endswith("*")), None)}
        fw = {}
        for f in sorted(glob.glob(os.path.join(devdir, "fw_version", "*_fw_version"))):
            name = os.path.basename(f)[:-len("_fw_version")]
            if name in ("smc", "mec", "sdma", "vcn", "rlc", "pfp", "me", "sos", "asd"):
                fw[name] = rd(f)
        out["fw_version"] = fw
        hw = sorted(glob.glob(os.path.join(devdir, "hwmon", "hwmon*")))
        if hw:
            cap = rd(os.path.join(hw[0], "power1_cap"))
            out["power_cap_w"] = float(cap) / 1e6 if cap else None
            for n, key in (("freq1_input", "sclk_idle_mhz"), ("freq2_input", "mclk_idle_mhz")):
                v = rd(os.path.join(hw[0], n))
                out[key] = float(v) / 1e6 if v else None
    out["amdgpu_driver"] = rd("/sys/module/amdgpu/version")
    probe = (ctypes.c_double * 8)()
    rc = lib.pndf_debug_mem_probe(int(dev_index), probe, 8)
    out["mem_probe"] = ({"l2_hit_latency_ns": round(probe[0], 1), "infinity_cache_latency_ns": round(probe[1], 1),
                         "hbm_latency_ns": round(probe[2], 1), "stream_read_gbps": round(probe[3], 1),
                         # the weight ring alone (no arithmetic), all CUs: what one CU's LDS ring is fed at on this box
                         "ring_only_gbps_per_cu": round(probe[6], 1), "ring_only_ns_per_slot": round(probe[7], 1),
                         "ring_needed_gbps_per_cu": {"f16x3": 51, "fp32": 17},
                         "what": "dependent-load latency of one lane walking 128-byte lines of a 1 MiB / 64 MiB / 1 GiB footprint "
                                 f"({int(probe[5])} hops each, wall-clock counter at {probe[4]:.0f} MHz); read bandwidth over 1 GiB, all CUs "
                                 "(posendf_amd/csrc/pndf_probe.hip)"} if rc == 0 else {"error": rc})
    return out


_AMDSMI_INDEX = {}


def _json_after_banner(out):
    """The JSON document of a tool's stdout that may start with banner / warning lines."""
    for i, ch in enumerate(out):
        if ch in "[{":
            try:
                return json.loads(out[i:])
            except ValueError:
                continue
    raise ValueError("no JSON document in the output")


def device_bdf(dev_index):
    """PCI address (domain:bus:device.function) of HIP device `dev_index`, as sysfs and amd-smi spell it."""
    import torch
    try:
        pr = torch.cuda.get_device_properties(dev_index)
        return f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
    except (AttributeError, RuntimeError):
        return None


def _amdsmi_gpu_index(bdf):
    """amd-smi's own index of the GPU at PCI address `bdf` (`amd-smi list --json`).  amd-smi enumerates every GPU of the
    node whatever HIP_/ROCR_VISIBLE_DEVICES say, so entry 0 of an unfiltered `amd-smi metric` need not be cuda:0.
    None when the tool is missing or no entry matches."""
    import subprocess
    if bdf in _AMDSMI_INDEX:
        return _AMDSMI_INDEX[bdf]
    idx = None
    try:
        out = subprocess.run(["amd-smi", "list", "--json"], capture_output=True, text=True, timeout=40).stdout
        lst = _json_after_banner(out)
        if isinstance(lst, dict):
            lst = lst.get("gpu_data") or lst.get("gpus") or [lst]
        for k, g in enumerate(lst):
            if str(g.get("bdf", "")).lower() == str(bdf).lower():
                idx = int(g.get("gpu", k))
        if idx is None and len(lst) == 1:      # a one-GPU node: nothing to confuse it with, whatever the listing calls its address
            idx = int(lst[0].get("gpu", 0))
    except Exception:
        idx = None
    _AMDSMI_INDEX[bdf] = idx
    return idx


def _amdsmi_metric(bdf=None):
    """One `amd-smi metric --json` reading of the device at PCI address `bdf` (None: the only / first device -- single-GPU
    tools): the firmware's lifetime accumulators (energy, throttler residencies at ~1 kHz) and the instantaneous power /
    clock / temperatures.  None when the tool is missing or fails, or when `bdf` is given and amd-smi lists no such device."""
    import subprocess
    try:
        cmd, idx = ["amd-smi", "metric", "--json"], None
        if bdf is not None:
            idx = _amdsmi_gpu_index(bdf)
            if idx is None:
                return None
            cmd = ["amd-smi", "metric", "-g", str(idx), "--json"]
        g = _json_after_banner(subprocess.run(cmd, capture_output=True, text=True, timeout=40).stdout)
        entries = (g.get("gpu_data") or [g]) if isinstance(g, dict) else g
        g = next((e for e in entries if idx is not None and e.get("gpu") == idx), entries[0])
        thr = g.get("throttle") or {}

        def val(x):
            return x.get("value") if isinstance(x, dict) else x
        num = lambda x: float(x) if isinstance(x, (int, float)) else None      # noqa: E731

        def xcd_sum(key):      # per-XCD accumulators of the first partition: {"xcp_0": [8 values]} -> their sum
            v = thr.get(key)
            v = v.get("xcp_0") if isinstance(v, dict) else None
            return float(sum(x for x in v if isinstance(x, (int, float)))) if isinstance(v, list) and any(isinstance(x, (int, float)) for x in v) else None
        return {"below_limit_power": xcd_sum("gfx_clk_below_host_limit_power_accumulated"),
                "below_limit_thermal": xcd_sum("gfx_clk_below_host_limit_thermal_accumulated"),
                "below_limit_total": xcd_sum("total_gfx_clk_below_host_limit_accumulated"),
                "low_utilization": xcd_sum("low_utilization_accumulated"),
                "energy_j": num(val((g.get("energy") or {}).get("total_energy_consumption"))),
                "acc": num(thr.get("accumulation_counter")), "ppt": num(thr.get("ppt_accumulated")),
                "prochot": num(thr.get("prochot_accumulated")), "socket_thm": num(thr.get("socket_thermal_accumulated")),
                "vr_thm": num(thr.get("vr_thermal_accumulated")), "hbm_thm": num(thr.get("hbm_thermal_accumulated")),
                "socket_power_w": num(val((g.get("power") or {}).get("socket_power"))),
                "gfx_clk_mhz": num(val((((g.get("clock") or {}).get("gfx_0") or {}).get("clk")))),
                "hotspot_c": num(val((g.get("temperature") or {}).get("hotspot"))),
                "mem_c": num(val((g.get("temperature") or {}).get("mem")))}
    except Exception:
        return None


def power_window(hot, sync, kernel_ms, min_s=3.0, max_s=45.0, bdf=None):
    """Which limiter holds the clock while the measured kernel runs, and what a launch costs in energy: two firmware readings
    (`amd-smi metric`: energy accumulator, throttler residency counters) taken by a helper thread WHILE the main thread keeps
    the kernel running back to back.  Between the readings: mean package power = d energy / d time (the firmware's own ~1 kHz
    sample counter is the clock), fraction of the samples in which the package-power (PPT) / thermal / PROCHOT limiters were
    active.  Outside the timed region; one GPU, rank 0.  `bdf`: PCI address of the device the kernel runs on (device_bdf):
    the readings are taken from THAT device of the node; an error block when amd-smi does not list it."""
    import threading
    reads = []
    if bdf is not None and _amdsmi_gpu_index(bdf) is None:
        return {"error": f"amd-smi lists no device at {bdf} (or the tool is missing)"}

    def sampler():
        a = _amdsmi_metric(bdf)
        reads.append(a)
        if a is None:
            return
        time.sleep(min_s)
        reads.append(_amdsmi_metric(bdf))
    th = threading.Thread(target=sampler, daemon=True)
    t0 = time.perf_counter()
    n = 0
    th.start()
    while th.is_alive() and time.perf_counter() - t0 < max_s:
        hot()
        sync()
        n += 1
    th.join(timeout=1.0)
    if len(reads) < 2 or reads[0] is None or reads[1] is None:
        return {"error": "amd-smi metric unavailable"}
    a, b = reads
    if None in (a["acc"], b["acc"], a["energy_j"], b["energy_j"]) or b["acc"] <= a["acc"]:
        return {"error": "accumulators missing", "first": a, "second": b}
    dt = (b["acc"] - a["acc"]) * 1e-3                             # the accumulation counter runs at ~1 kHz
    frac = lambda k: (None if (a[k] is None or b[k] is None) else (b[k] - a[k]) / (b["acc"] - a["acc"]))      # noqa: E731
    watts = (b["energy_j"] - a["energy_j"]) / dt
    return {"what": "two `amd-smi metric` readings while the measured kernel runs back to back (launches kept up by the main "
                    "thread); rates between the readings, the firmware's ~1 kHz accumulation counter as the clock",
            "device_bdf": bdf, "window_s": dt, "launches_during_window_and_tool_startup": n, "mean_package_w": watts,
            "energy_j_per_launch": watts * kernel_ms * 1e-3,
            "ppt_limited_frac": frac("ppt"), "socket_thermal_limited_frac": frac("socket_thm"), "vr_thermal_limited_frac": frac("vr_thm"),
            "hbm_thermal_limited_frac": frac("hbm_thm"), "prochot_frac": frac("prochot"),
            # per-XCD residencies (sum over the 8 XCDs / 8): shader clock below the host limit because of power / temperature / any reason
            "xcd_clk_below_limit_frac": {k: (None if frac(k) is None else frac(k) / 8.0)
                                         for k in ("below_limit_power", "below_limit_thermal", "below_limit_total", "low_utilization")},
            "socket_power_w_inst": [a["socket_power_w"], b["socket_power_w"]], "gfx_clk_mhz_inst": [a["gfx_clk_mhz"], b["gfx_clk_mhz"]],
            "hotspot_c": [a["hotspot_c"], b["hotspot_c"]], "hbm_c": [a["mem_c"], b["mem_c"]]}


def smi_snapshot():
    """One rocm-smi reading (fallback when the hwmon files are absent); called while a launch is in flight."""
    import re
    import subprocess
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=20).stdout
    except (OSError, subprocess.SubprocessError):
        return None
    sclk = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", out)
    power = re.search(r"Package Power \(W\): ([0-9.]+)", out)
    if not (sclk and power):
        return None
    return {"sclk_mhz": float(sclk.group(1)), "package_w": float(power.group(1)), "telemetry_samples": 1,
            "telemetry_source": "rocm-smi --showpower --showclocks, one reading during the timed loop"}


def gpu_torch_baseline(act, sd, B, proj_steps, dev, timed_steps=5):
    """The comparator of north_star's '>= 10x the reference single-GPU PyTorch poses/sec': the same PyTorch restatement
    of the reference run through stock PyTorch-ROCm on this GPU (fp32, same batch), timed over `timed_steps` projection
    steps and extrapolated linearly to `proj_steps` (every step does identical work; the chained autograd graph of the
    reference is detached between steps, which only saves memory)."""
    import torch
    from oracle.posendf_torch import RefNet, project
    from posendf_amd import synth
    net = RefNet(act).to(dev)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    q = torch.from_numpy(synth.make_poses(B, seed=1234)).to(dev)
    project(net, q, 2)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    project(net, q, timed_steps)
    torch.cuda.synchronize(dev)
    per_step = (time.perf_counter() - t0) / timed_steps
    del net, q
    torch.cuda.empty_cache()
    return {"what": "PyTorch-ROCm fp32 restatement of the reference (oracle/posendf_torch.py) on the same MI355X",
            "batch": B, "timed_steps": timed_steps, "ms_per_step": per_step * 1e3,
            "value": B / (per_step * proj_steps), "unit": f"projected poses/s (extrapolated to {proj_steps} steps)",
            "achieved_tflops": B * FLOP_PER_POSE_STEP / per_step / 1e12, "torch": torch.__version__}


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: re-execute under torch.distributed.run, one rank per GPU."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count()
    if have < args.gpus and not (have and os.environ.get("PNDF_BENCH_SHARE_DEVICE") == "1"):      # (tests: ranks share devices)
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {have} GPU(s) visible; refusing to report a smaller job as N={args.gpus}")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=65536, help="poses per GPU")
    ap.add_argument("--proj-steps", type=int, default=100)
    ap.add_argument("--act", default="lrelu")
    ap.add_argument("--precision", default="f16x3", choices=["f16x3", "fp32", "f16", "bf16"],
                    help="trunk arithmetic of the measured path; f16x3 and fp32 meet the 1e-4 parity gates (tests -m "
                         "gpu); f16 / bf16 are the reduced-precision comparison points of BASELINE.json configs[2], not a valid "
                         "headline")
    ap.add_argument("--no-fp32-ref", action="store_true",
                    help="skip the short exact-fp32 and plain-f16 runs reported beside f16x3")
    ap.add_argument("--cpu-budget", type=float, default=12.0, help="seconds of CPU baseline work")
    ap.add_argument("--diagnostics", action="store_true",
                    help="also emit the analysis blocks that are not part of the contract line: box (+ mem_probe, power_window), "
                         "regions, f16_single, fp16_checkpoint, host_twin, host_boundary, motion_denoise_config4 (~40 s more)")
    ap.add_argument("--workload", default="project", choices=["project", "denoise"],
                    help="project (default): BASELINE.json configs[2]/[3], the headline; denoise: configs[4], whole sequences "
                         "sharded over the ranks, fused Adam steps of the reference's objective, final gather of the poses")
    ap.add_argument("--seqs", type=int, default=64, help="--workload denoise: sequences per GPU (configs[4]: 512 over 8 GPUs)")
    ap.add_argument("--frames", type=int, default=300, help="--workload denoise: frames per sequence")
    ap.add_argument("--adam-steps", type=int, default=10, help="--workload denoise: Adam steps per harness step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-box", action="store_true", help="--diagnostics: skip the `box` / `regions` blocks")
    ap.add_argument("--no-power-window", action="store_true", help="--diagnostics: skip the energy / throttler reading of the `box` block (~10 s)")
    ap.add_argument("--no-gpu-torch-baseline", action="store_true")
    ap.add_argument("--lbs-torch-baseline", action="store_true", help="also time a PyTorch restatement of the body-model terms (4 x 300 frames)")
    ap.add_argument("--no-motion-denoise", action="store_true", help="--diagnostics: skip the configs[4] side block")
    ap.add_argument("--no-parity-sample", action="store_true", help="skip the oracle check of a sample of the timed result")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)                                       # does not return

    # stdout carries exactly ONE line, the JSON of rank 0: libraries that print to the C stdout (RCCL announces
    # "Librccl path : ..." there when a process group comes or goes) are sent to stderr by pointing fd 1 at fd 2 for the
    # whole run; the JSON line is written to the saved original descriptor at the very end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    from posendf_amd import PoseNDF, amass_config, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs MI355X GPUs; the engine has no CPU path")
    # One rank per GPU.  PNDF_BENCH_SHARE_DEVICE=1 (tests only) lets several ranks share the visible devices round-robin:
    # a one-GPU box can then run the REAL multi-process path -- N processes, rank / offset / gather / max-over-ranks --
    # with PNDF_BENCH_BACKEND=gloo (RCCL refuses two ranks on one device; gloo does not).  The line then says so.
    shared = os.environ.get("PNDF_BENCH_SHARE_DEVICE") == "1"
    ndev = torch.cuda.device_count()
    if local >= ndev and not shared:
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local} but only {ndev} device(s) visible")
    dev_index = local % ndev
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    backend = os.environ.get("PNDF_BENCH_BACKEND", "nccl")                   # nccl IS RCCL on ROCm
    use_dist = world > 1 or os.environ.get("PNDF_BENCH_FORCE_DIST") == "1"   # the flag exercises RCCL with 1 rank
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    sd = synth.make_weights(0, 2.0, 0.1)                        # BASELINE.md section 3 "live regime"
    precision = "fp32" if (args.act == "softplus" and args.precision in ("f16", "bf16")) else args.precision

    def build(prec, act=None, weights=None):
        cfg = amass_config(act or args.act, f"cuda:{dev_index}")
        cfg["engine"] = {"precision": prec}
        m = PoseNDF(cfg)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in (weights or sd).items()})
        m.eval()
        return m

    net = build(precision)
    B = args.batch
    from posendf_amd.sharding import all_gather_blocks
    if args.workload == "project":
        # shard `rank` of the global batch: reference input distribution (sample_poses.py:96-97), seeded
        q0 = torch.from_numpy(synth.make_poses(B, seed=1234, offset=rank)).to(dev)
        rows = B                                                # rows of the gathered tensor this rank contributes

        def hot():                                              # the dominant kernel: ONE persistent launch
            return net.project(q0, steps=args.proj_steps)
        out_shape = (21, 4)
    else:
        # BASELINE.json configs[4]: whole sequences per rank (experiments/motion_denoise.py:171-188 loops sequences), the
        # reference's objective (pose prior on the engine + SMPL-shaped body model terms), fused Adam steps, no collective
        # until the final gather of the denoised poses (sharding.denoise_sharded)
        from posendf_amd import BodyModel
        from posendf_amd.motion_denoise import MotionDenoise
        S, T = args.seqs, args.frames
        bm = BodyModel(synth.make_body_model(seed=11), device=f"cuda:{dev_index}")
        gen = torch.Generator().manual_seed(1000 + rank)
        theta = (torch.cumsum(0.02 * torch.randn(S, T, 69, generator=gen), dim=1) + 0.3 * torch.randn(S, 1, 69, generator=gen)
                 + 0.1 * torch.randn(S, T, 69, generator=gen)).to(dev)
        md = MotionDenoise(net, body_model=bm, device=f"cuda:{dev_index}")
        rows = S
        half = max(1, args.adam_steps // 2)                     # two outer iterations: the second one has the data term (:92)

        def hot():
            out, _ = md.denoise(theta, iterations=2, steps_per_iter=half, fused=True, record=False)
            return out, None
        out_shape = (T, 69)
    # receive buffer of the final gather, allocated once: equal blocks go straight into it (all_gather_into_tensor).
    # project: ONE collective carries poses AND last distances, 85 floats per pose (sharding.gather_projected)
    from posendf_amd.sharding import ROW_FLOATS, gather_projected
    if args.workload == "project":
        gathered = torch.empty((rows * world, ROW_FLOATS), device=dev, dtype=torch.float32) if use_dist else None
    else:
        gathered = torch.empty((rows * world,) + out_shape, device=dev, dtype=torch.float32) if use_dist else None

    def final_gather(qp, d):                                    # the only collective of a pass: final gather over xGMI
        if args.workload == "project":
            gather_projected(qp, d, rows * world, out=gathered)
        else:
            all_gather_blocks(qp, rows * world, out=gathered)

    def one_pass():
        qp, d = hot()
        if use_dist:
            final_gather(qp, d)
        return qp, d

    for _ in range(args.warmup):
        one_pass()
    ev = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(3)) for _ in range(args.steps)]
    tele = GpuTelemetry(dev_index)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    tele.start()
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record()
        qp, d = hot()                                           # bracketed by HIP events on the launch stream
        ev[k][1].record()
        if use_dist:
            final_gather(qp, d)
        ev[k][2].record()
    smi = smi_snapshot() if (tele.src is None and rank == 0 and args.steps) else None      # (launches still in flight)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    tele.stop()
    telemetry = tele.summary() if (tele.samples or smi is None) else smi
    kern_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in ev])) if args.steps else float("nan")
    gather_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in ev])) if (args.steps and use_dist) else 0.0
    per_rank = None
    if use_dist:
        # every rank's own numbers travel to rank 0, so that the line PROVES its rank count
        # the gathered buffer holds every rank's block in rank order: each rank finds its own result in its own window
        mine_rows = gathered[rank * rows:(rank + 1) * rows]
        if args.workload == "project":      # [rows, 85]: the pose columns and the distance column of this rank's window
            own = bool(torch.equal(mine_rows[:, :ROW_FLOATS - 1], qp.reshape(rows, -1)) and torch.equal(mine_rows[:, ROW_FLOATS - 1:], d.reshape(rows, 1))) if args.steps else None
        else:
            own = bool(torch.equal(mine_rows, qp)) if args.steps else None
        mine = {"rank": rank, "local_rank": local, "device": dev_index, "pid": os.getpid(), "elapsed_s": elapsed,
                "kernel_ms": kern_ms, "gather_ms": gather_ms, "rows": rows, "own_block_in_gather": own}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
        t = torch.tensor([elapsed, kern_ms], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kern_ms = float(t[0].item()), float(t[1].item())
        if rank == 0 and args.steps and not all(r["own_block_in_gather"] for r in per_rank):
            raise SystemExit(f"final gather misplaced a block: {per_rank}")
    dist_info = None
    if use_dist:
        try:
            rccl = ".".join(str(x) for x in torch.cuda.nccl.version())      # torch's nccl IS RCCL on ROCm
        except Exception as exc:
            rccl = f"unavailable ({exc!r})"
        seen = {(r["pid"], r["device"]) for r in per_rank if r}
        dist_info = {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "devices_visible": ndev,
                     "rccl_version": rccl, "ranks_seen": len(seen), "devices_used": len({r["device"] for r in per_rank if r}),
                     "collectives_per_pass": 1,
                     "ranks_share_devices": bool(shared and world > ndev), "gather_ms": gather_ms, "per_rank": per_rank,
                     "gathered_rows": int(gathered.shape[0]), "floats_per_gathered_row": int(gathered[0].numel())}

    if args.workload == "denoise":
        if rank == 0:
            frames = S * T * world
            steps_done = 2 * half
            out = {"metric": "motion-denoise frame-steps/sec (configs[4]: Adam steps of the reference's objective)",
                   "value": frames * steps_done * args.steps / elapsed, "unit": "frame-steps/s", "n_gpus": world,
                   "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / max(args.steps, 1) * 1e3,
                   "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": KERNELS[precision][2], "data": "synthetic",
                   "config": {"workload": f"BASELINE.json configs[4]: {S} sequences/GPU x {T} frames, {steps_done} fused Adam steps per "
                                          f"harness step (2 outer iterations), pose prior ({precision}, act={args.act}) + "
                                          "SMPL-shaped synthetic body model (vertex temporal + joint data terms), whole sequences "
                                          "sharded over the ranks, final all_gather of the denoised poses",
                              "sequences_total": S * world, "frames": T, "adam_steps_per_step": steps_done,
                              "parallelism": f"sequence-sharded x{world}, final all_gather" if world > 1 else "single GPU"},
                   "adam_step_ms": elapsed / max(args.steps, 1) * 1e3 / steps_done,
                   "finite": bool(torch.isfinite(qp).all()), "telemetry": telemetry,
                   "parity": "pose prior pinned; body model unpinned (smplx is third-party and absent; oracle/lbs_np.py)"}
            if dist_info is not None:
                out["distributed"] = dist_info
            json_line = json.dumps(out)
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            os.write(json_fd, (json_line + "\n").encode())
        os.close(json_fd)
        return

    # What was timed is also CHECKED: a sample of the projected poses of the last timed pass against the oracle's fp64
    # trajectory of the same inputs, with the reference arithmetic's own fp32 trajectory beside it (outside the timed region;
    # rank 0 only, like the cpu_baseline leg).
    oracle_traj = {}                                            # the oracle's trajectories of the sample, computed once

    def parity_sample(n=128, result=None):
        from oracle import posendf_np as onp
        try:        # 256 hardware threads on the GPU box: keep numpy's BLAS from spreading small matmuls over all of them
            from threadpoolctl import threadpool_limits
            threadpool_limits(limits=16)
        except ImportError:
            pass
        idx = np.random.default_rng(0).choice(B, min(n, B), replace=False)
        if n not in oracle_traj:
            q_in = q0[idx].cpu().numpy()
            oracle_traj[n] = (onp.project(q_in, sd, steps=args.proj_steps, act=args.act, dtype=np.float64)[0],
                              onp.project(q_in, sd, steps=args.proj_steps, act=args.act)[0])
        q64, q32 = oracle_traj[n]

        def rows(a):
            a = np.asarray(a, np.float64).reshape(len(idx), -1)
            b = q64.reshape(len(idx), -1)
            return np.abs(a - b).max(1) / np.maximum(np.abs(b).max(1), 1e-30)

        mine, ref = rows((qp if result is None else result)[idx].cpu().numpy()), rows(q32)
        return {"what": f"{len(idx)} poses of {'the last timed pass' if result is None else 'this run'} vs the numpy oracle's fp64 {args.proj_steps}-step trajectory "
                        "(per-pose max |dq| / max |q|); `reference_fp32` = the reference arithmetic's own fp32 trajectory",
                "median": float(np.median(mine)), "p95": float(np.percentile(mine, 95)), "max": float(mine.max()),
                "reference_fp32": {"median": float(np.median(ref)), "p95": float(np.percentile(ref, 95)), "max": float(ref.max())},
                "tolerance": 1e-4, "within_tolerance_frac": float((mine <= 1e-4).mean())}

    parity = parity_sample() if (rank == 0 and not args.no_parity_sample) else None

    # the exact-fp32 and the plain-fp16 kernels beside the split-precision one (same inputs, short runs, outside the
    # timed region): the three points of BASELINE.json configs[2] "fp32 vs bf16"
    def side_run(prec, act=None, weights=None, keep=None):
        ref = build(prec, act, weights)
        ref.project(q0, steps=args.proj_steps)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        q_ref, d_ref = ref.project(q0, steps=args.proj_steps)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        tf = B * args.proj_steps * FLOP_PER_POSE_STEP / (ms * 1e-3) / 1e12
        # agreement with the measured kernel on this batch after the full projection (median per-pose relative difference)
        a, b = qp.reshape(B, -1), q_ref.reshape(B, -1)
        diff = ((a - b).abs().amax(1) / b.abs().amax(1).clamp_min(1e-30)).median().item()
        if keep is not None:
            keep.append(q_ref)
        return {"kernel": ref._engine_for(dev).kernel_name(), "kernel_ms": ms, "poses_per_s_per_gpu": B / (ms * 1e-3),
                "achieved_tflops": tf, "median_rel_diff_of_projected_poses_vs_f16x3": diff}

    # BASELINE.json configs[1]: the single forward + d d/d q launch on the same batch (outside the timed region)
    def fwd_grad_ms(model):
        eng = model._engine_for(dev)
        d1 = torch.empty(B, device=dev)
        g1 = torch.empty_like(q0)
        st = torch.cuda.current_stream(dev).cuda_stream
        eng.forward_grad(q0.data_ptr(), None, d1.data_ptr(), g1.data_ptr(), B, st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(5):
            eng.forward_grad(q0.data_ptr(), None, d1.data_ptr(), g1.data_ptr(), B, st)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 5

    # The boundary takes device pointers; a caller that holds HOST tensors (the facade accepts them, posendf.py:64) pays
    # the PCIe copies on top: pinned host -> device, project, device -> pinned host.  Reported beside `value`, never as it.
    def host_boundary_ms():
        qh = q0.cpu().pin_memory()
        oh = torch.empty_like(qh).pin_memory()
        ms = []
        for _ in range(3):
            torch.cuda.synchronize()
            t = time.perf_counter()
            qd = qh.to(dev, non_blocking=True)
            qo, _ = net.project(qd, steps=args.proj_steps)
            oh.copy_(qo, non_blocking=True)
            torch.cuda.synchronize()
            ms.append((time.perf_counter() - t) * 1e3)
        return sorted(ms)[1]

    side = world == 1 and not args.no_fp32_ref      # the side runs belong to the N = 1 line; scaling runs stay short
    fwd_grad = host_ms = None
    # Where the cycles of a step go ON THIS BOX: one 3-step launch of the instrumented build of the measured kernel right
    # after the timed loop (s_memtime stamps per region; the weight ring's counted wait and barrier sampled every 16th slot),
    # and what the box is.  A slow box must be diagnosable from its own line.
    regions = box = None
    diag = bool(args.diagnostics)
    if rank == 0 and world == 1 and args.workload == "project" and diag and not args.no_box:      # (N = 1 line only, like the side runs)
        eng0 = net._engine_for(dev)
        try:
            if not (args.act == "softplus" and B > 64 * torch.cuda.get_device_properties(dev).multi_processor_count):
                regions = eng0.project_timing(q0, steps=3)
                regions["kernel"] = eng0.kernel_name() + "_timing"
            if args.act != "softplus" and precision == "f16x3":
                # the same launch through the exact-fp32 instrumented kernel: a ring that stalls the split kernel (a slot lasts
                # ~300 cycles) but not this one (~2,200) is a latency problem; both alike is something else
                r32 = build("fp32")._engine_for(dev).project_timing(q0, steps=3)
                regions["fp32_kernel"] = {k: r32[k] for k in ("cycles_per_wave_step", "effective_sclk_ghz", "ring", "launch_ms")}
        except Exception as exc:       # an analysis aid must not take the line down
            regions = {"error": repr(exc)}
        try:
            box = box_block(dev_index, eng0.lib)
            if not args.no_power_window:
                box["power_window"] = power_window(hot, torch.cuda.synchronize, kern_ms, bdf=device_bdf(dev_index))
        except Exception as exc:           # (a diagnostics block must never cost the line)
            box = {"error": repr(exc)}
    if side:
        if diag:
            host_ms = host_boundary_ms()
        ms1 = fwd_grad_ms(net)
        fwd_grad = {"workload": f"BASELINE.json configs[1]: one forward + d d/d q launch, batch={B}", "precision": precision,
                    "ms": ms1, "pose_steps_per_s": B / (ms1 * 1e-3),
                    "achieved_tflops": B * FLOP_PER_POSE_STEP / (ms1 * 1e-3) / 1e12}

    # BASELINE.json configs[4] on one GPU's share (64 of 512 sequences x 300 frames) with the REFERENCE's objective: pose prior
    # on the engine + SMPL-shaped body model (synthetic parameters, 6,890 vertices) with the vertex temporal and joint data
    # terms fused (csrc/pndf_lbs.hip).  A side block, outside the timed region; never `value`.
    def motion_denoise_block(S=64, T=300):
        from posendf_amd import BodyModel
        from posendf_amd.motion_denoise import MotionDenoise
        bm = BodyModel(synth.make_body_model(seed=11), device=f"cuda:{dev_index}")
        g = torch.Generator().manual_seed(0)
        theta = (torch.cumsum(0.02 * torch.randn(S, T, 69, generator=g), dim=1) + 0.3 * torch.randn(S, 1, 69, generator=g)
                 + 0.1 * torch.randn(S, T, 69, generator=g)).to(dev)
        j0 = bm.joints_of(theta + 0.02)
        out = torch.empty_like(theta)
        bm.terms_grad(theta, j0, 2, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(5):
            bm.terms_grad(theta, j0, 2, out=out)
        e1.record()
        torch.cuda.synchronize()
        ms_lbs = e0.elapsed_time(e1) / 5
        md = MotionDenoise(net, body_model=bm, device=f"cuda:{dev_index}")
        md.denoise(theta, iterations=1, steps_per_iter=2, fused=True)
        torch.cuda.synchronize()
        t = time.perf_counter()
        res, _ = md.denoise(theta, iterations=2, steps_per_iter=5, fused=True)
        torch.cuda.synchronize()
        ms_step = (time.perf_counter() - t) / 10 * 1e3
        flop = 2 * (2 * 207 * 20670 + 2 * 6890 * 24 * 12)                 # per frame, forward + reverse
        tf = S * T * flop / (ms_lbs * 1e-3) / 1e12
        split = bm.precision == "f16x3"
        peak = PEAK_F16_MFMA_TFLOPS if split else PEAK_FP32_MFMA_TFLOPS
        torch_ref = None
        if args.lbs_torch_baseline:
            # (opt-in: PyTorch 2.10+rocm7.0 dies with a GPU memory-access fault in this pass at 4,800 frames and above)
            # the same body-model terms + autograd as stock PyTorch-ROCm executes them (oracle/lbs_torch.py: smplx's lbs() restated
            # op by op, fp32) on the same GPU and the same poses -- a comparator, like `gpu_torch_baseline` of the headline
            from oracle.lbs_torch import torch_lbs
            bm_np = synth.make_body_model(seed=11)

            St = 4       # 1,200 frames: the largest size tried that stock PyTorch survives (16 x 300 and 64 x 300 fault)

            def torch_pass():
                th = theta[:St].reshape(-1, 69).detach().clone().requires_grad_(True)
                verts, joints = torch_lbs(th, bm_np, torch.float32)
                verts, joints = verts.view(St, T, -1, 3), joints.view(St, T, -1, 3)
                loss = 30.0 * torch.mean(torch.sqrt(torch.sum((verts[:, :-1] - verts[:, 1:]) ** 2, dim=3))) \
                    + 100.0 / 3.0 * torch.mean(torch.sqrt(torch.sum((joints - j0.view(S, T, -1, 3)[:St]) ** 2, dim=3)))
                loss.backward()
                return th.grad

            g_t = torch_pass()
            torch.cuda.synchronize()
            e0.record()
            for _ in range(2):
                g_t = torch_pass()
            e1.record()
            torch.cuda.synchronize()
            ms_t = e0.elapsed_time(e1) / 2
            scale = out[:St].abs().max().item()
            torch_ref = {"what": f"PyTorch-ROCm fp32 restatement of smplx lbs() + the two terms + autograd (oracle/lbs_torch.py), same GPU, "
                                 f"the first {St} of the {S} sequences (larger calls end in a GPU memory-access fault inside PyTorch), "
                                 "weights of iteration 2",
                         "sequences": St, "ms": ms_t, "ms_per_sequence": ms_t / St,
                         "speedup_of_body_model_pass_per_sequence": (ms_t / St) / (ms_lbs / S), "torch": torch.__version__,
                         "max_abs_diff_of_gradient_rel": (g_t.view(St, T, 69) * St - out[:St]).abs().max().item() / max(scale, 1e-30)}
        return {"workload": f"BASELINE.json configs[4], one GPU's share: {S} sequences x {T} frames, reference objective (pose prior + "
                            "SMPL vertex temporal term + joint data term), synthetic SMPL-shaped body model, fused Adam steps",
                "fused_adam_step_ms": ms_step, "frames_per_s": S * T / (ms_step * 1e-3),
                "body_model_pass": {"kernel": ("pndf_lbs_vertex_split_terms_kernel" if split else "pndf_lbs_vertex_terms_kernel") + " (+ pose kernels)",
                                    "precision": bm.precision, "ms": ms_lbs, "bound": "mfma",
                                    "achieved": tf, "peak": peak, "unit": "TFLOP/s", "frac": tf / peak,
                                    "mfma_issued_per_algorithmic_flop": 3 if split else 1, "algorithmic_flop_per_frame": flop},
                "gpu_torch_baseline": torch_ref,
                "finite": bool(torch.isfinite(res).all()), "parity": "unpinned (smplx is third-party and absent; oracle/lbs_np.py)"}

    denoise = motion_denoise_block() if (side and diag and not args.no_motion_denoise and args.act != "softplus") else None

    fp32_ref = f16_ref = sp_ref = h16_ref = bf16_ref = None
    if precision == "f16x3" and side and args.act != "softplus":
        # the activation of the reference's published checkpoints (sample_poses.py:115, motion_denoise.py:162-163)
        sp_ref = side_run("f16x3", "softplus")
        sp_ref["frac_of_fp16_mfma_peak"] = sp_ref["achieved_tflops"] / PEAK_F16_MFMA_TFLOPS
        sp_ref.pop("median_rel_diff_of_projected_poses_vs_f16x3")     # another network: not comparable
    if precision == "f16x3" and side and diag and args.act != "softplus":
        # a half-precision checkpoint (the same network with its weights rounded to fp16): ANOTHER network, on which the
        # lo*hi term of the split arithmetic vanishes identically and the engine selects the two-term kernels by itself
        h16_ref = side_run("f16x3", weights={k: v.astype(np.float16).astype(np.float32) for k, v in sd.items()})
        h16_ref.pop("median_rel_diff_of_projected_poses_vs_f16x3")      # another network: not comparable
        h16_ref["what"] = ("weights rounded to fp16 (a half-precision checkpoint): every lo half of the packed weights is "
                           "zero, pndf_load_weights selects the two-term kernels; bit-identical to the three-term kernels "
                           "on those weights (tests/test_trained_regime.py); not the headline network")
    if precision == "f16x3" and side:
        fp32_ref = side_run("fp32")
        fp32_ref["frac_of_fp32_mfma_peak"] = fp32_ref["achieved_tflops"] / PEAK_FP32_MFMA_TFLOPS
    if precision == "f16x3" and side and args.act != "softplus":
        # the literal second half of BASELINE.json configs[2] "fp32 vs bf16": one bf16 MFMA per product block, same schedule
        kept = []
        bf16_ref = side_run("bf16", keep=kept)
        bf16_ref["frac_of_bf16_mfma_peak"] = bf16_ref["achieved_tflops"] / PEAK_F16_MFMA_TFLOPS
        if not args.no_parity_sample:
            bf16_ref["parity_sample"] = parity_sample(result=kept[0])
        bf16_ref["note"] = ("reduced precision: operands rounded to bfloat16 (8 significant bits), one v_mfma_f32_16x16x32_bf16 per "
                            "product block; two to three orders of magnitude outside the 1e-4 parity bar "
                            "(tests/test_gpu_parity.py::test_one_term_kernels_compute_their_stated_arithmetic), a comparison point only")
        del kept
    if precision == "f16x3" and side and diag and args.act != "softplus":
        f16_ref = side_run("f16")
        f16_ref["frac_of_fp16_mfma_peak"] = f16_ref["achieved_tflops"] / PEAK_F16_MFMA_TFLOPS
        f16_ref["note"] = ("reduced precision: operands rounded to fp16, one MFMA per product block; outside the 1e-4 "
                           "parity bar (tests/test_gpu_parity.py::test_one_term_kernels_compute_their_stated_arithmetic), reported "
                           "as a comparison point only")

    if rank == 0:
        _, peak, dtype = KERNELS[precision]
        kname = net._engine_for(dev).kernel_name()
        # HBM traffic of the dominant kernel: from the committed PMC passes of the same command
        # (tools/gpu_profile.sh -> profiles/traffic.json); bench.py itself cannot run rocprofv3
        traffic, traffic_stale = None, None
        tpath = os.path.join(REPO, "profiles", "traffic.json")
        from posendf_amd.build_id import source_id
        if os.path.exists(tpath) and B == 65536 and args.proj_steps == 100:
            with open(tpath) as f:
                entry = json.load(f).get(kname) or {}
            traffic = entry.get("hbm_bytes_per_launch")
            # the PMC passes are a separate rocprofv3 run (tools/gpu_profile.sh): tie them to the sources that run now
            traffic_stale = entry.get("source_id") != source_id()
        total = B * world * args.steps
        achieved = B * args.proj_steps * FLOP_PER_POSE_STEP / (kern_ms * 1e-3) / 1e12
        out = {
            "metric": "projected poses/sec (100 grad steps, 21-joint quat)",
            "value": total / elapsed,
            "unit": "poses/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / max(args.steps, 1) * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": dtype,
            "data": "synthetic",
            # (the driver's record keeps ~120 characters of a string: what was measured comes first)
            "config": {"workload": (f"cfg[2] B={B}/GPU x{args.proj_steps} steps {args.act}: "
                                    + {"f16x3": "f16x3 (fp32-split) timed; fp32 (exact), bf16 (not parity grade) reported",
                                       "fp32": "fp32 (exact) timed", "f16": "f16 (NOT parity grade) timed",
                                       "bf16": "bf16 (NOT parity grade) timed"}[precision]
                                    + f"; BASELINE.json configs[2], {args.proj_steps}-step project() loop, amass.yaml arch, random-init "
                                      "weights (uniform +-2/sqrt(fan_in), lin6.bias=0.1)"),
                       "precision": precision,
                       "precisions": ("f16x3 = fp32 operands split into fp16 hi+lo, 3 MFMAs per block, fp32 accumulate (timed); "
                                      "fp32 = exact fp32 MFMA (reported beside it: fp32_exact); bf16 = operands rounded to bfloat16, "
                                      "1 MFMA per block (reported beside it: bf16, with its error against the fp64 oracle -- not parity "
                                      "grade; its fp16 sibling with 3 more mantissa bits at the same rate: f16_single under --diagnostics)"),
                       "parity": ("NOT parity grade (operands rounded to 16 bits)" if precision in ("f16", "bf16") else
                                  "same 1e-4 gates as the fp32 kernel (tests/test_gpu_parity.py, both precisions)"),
                       "global_batch": B * world, "proj_steps": args.proj_steps,
                       "parallelism": f"batch-sharded x{world}, final RCCL all_gather" if world > 1 else "single GPU"},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": traffic, "traffic_stale": traffic_stale,
                         "mfma_issued_per_algorithmic_flop": 3 if precision == "f16x3" else 1,
                         **({"issued_tflops": 3 * achieved, "sustained_mfma_wall_tflops": SUSTAINED_F16_MFMA_TFLOPS,
                             "frac_of_sustained_wall": 3 * achieved / SUSTAINED_F16_MFMA_TFLOPS,
                             "wall_note": "dense fp16 MFMA with random operands, issued in the kernel's own pattern, sustains "
                                          "2.0 PFLOP/s under the power limit (1.7 rotating over eight accumulators, 2.2 on one; "
                                          "tools/ubench/mfma_power.hip); the kernel is power-bound"}
                            if precision == "f16x3" else {}),
                         "traffic_unit": "HBM bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE, profiles/traffic.json)",
                         "traffic_source": ("profiled on ANOTHER box with separate rocprofv3 --pmc passes of this command "
                                            "(tools/gpu_profile.sh; same sources: traffic_stale false) -- a property of the kernel, "
                                            "not a measurement of this run") if traffic is not None else None,
                         "algorithmic_bytes_per_launch": B * 676 + 10720 * 1024,
                         "kernel": kname, "kernel_ms": kern_ms,
                         "kernel_ms_median": float(np.median([e[0].elapsed_time(e[1]) for e in ev])) if args.steps else None,
                         **telemetry,
                         "algorithmic_flop_per_launch": B * args.proj_steps * FLOP_PER_POSE_STEP},
        }
        if dist_info is not None:
            out["distributed"] = dist_info
        if box is not None:
            out["box"] = box
        if regions is not None:
            out["regions"] = regions
        if parity is not None:
            out["parity_sample"] = parity
        if host_ms is not None:
            out["host_boundary"] = {"what": "pinned host poses -> device, project(), device -> pinned host (median of 3)",
                                    "ms": host_ms, "poses_per_s": B / (host_ms * 1e-3)}
        if fwd_grad is not None:
            out["forward_grad_single_launch"] = fwd_grad
        rl = out["roofline"]
        if fp32_ref is not None:
            # the SAME-arithmetic figure (exact fp32 products): first-class, and as scalars of `roofline` so that it survives
            # in the driver's record next to the split-precision headline
            out["fp32_exact"] = fp32_ref
            rl["fp32_exact_poses_per_s"] = fp32_ref["poses_per_s_per_gpu"]
            rl["fp32_exact_kernel_ms"] = fp32_ref["kernel_ms"]
            rl["fp32_exact_frac_of_fp32_mfma_peak"] = fp32_ref["frac_of_fp32_mfma_peak"]
        if bf16_ref is not None:
            # (scalars of `roofline` too: the driver's record keeps those)
            out["bf16"] = bf16_ref
            rl["bf16_poses_per_s"] = bf16_ref["poses_per_s_per_gpu"]
            rl["bf16_kernel_ms"] = bf16_ref["kernel_ms"]
            if "parity_sample" in bf16_ref:
                rl["bf16_median_rel_err_vs_fp64"] = bf16_ref["parity_sample"]["median"]
        if f16_ref is not None:
            out["f16_single"] = f16_ref
        if sp_ref is not None:
            if not args.no_gpu_torch_baseline:
                # the reference's scripts load softplus checkpoints (sample_poses.py:115, motion_denoise.py:162-163): the >= 10x
                # denominator for THIS activation too
                gts = gpu_torch_baseline("softplus", sd, B, args.proj_steps, dev)
                gts["speedup_of_softplus_kernel"] = sp_ref["poses_per_s_per_gpu"] / gts["value"]
                sp_ref["gpu_torch_baseline"] = gts
            out["softplus"] = sp_ref
            # the activation of the reference's own checkpoints gets its own roofline block: the same
            # algorithmic work, its own kernel, time and fabric traffic (the fp32 derivative scratch)
            sp_traffic = sp_stale = None
            if os.path.exists(tpath) and B == 65536 and args.proj_steps == 100:
                with open(tpath) as f:
                    e_sp = json.load(f).get(sp_ref["kernel"]) or {}
                sp_traffic, sp_stale = e_sp.get("hbm_bytes_per_launch"), e_sp.get("source_id") != source_id()
            out["roofline_softplus"] = {
                "bound": "mfma", "achieved": sp_ref["achieved_tflops"], "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": sp_ref["achieved_tflops"] / PEAK_F16_MFMA_TFLOPS, "traffic": sp_traffic, "traffic_stale": sp_stale,
                "kernel": sp_ref["kernel"], "kernel_ms": sp_ref["kernel_ms"], "poses_per_s": sp_ref["poses_per_s_per_gpu"],
                "algorithmic_bytes_per_launch": B * 676 + 10720 * 1024,
                "algorithmic_flop_per_launch": B * args.proj_steps * FLOP_PER_POSE_STEP,
                "derivative_scratch_bytes_per_launch": 2 * 2624 * 4 * B * args.proj_steps,
                "vs_lrelu_kernel_ms": sp_ref["kernel_ms"] / kern_ms,
                "what": "one launch of the softplus kernel on the same batch (outside the timed region); traffic = PMC passes of "
                        "`bench.py --act softplus` (profiles/traffic.json): fp32 derivatives written forward, read backward"}
            rl["softplus_poses_per_s"] = sp_ref["poses_per_s_per_gpu"]
            rl["softplus_kernel_ms"] = sp_ref["kernel_ms"]
            rl["softplus_frac"] = sp_ref["achieved_tflops"] / PEAK_F16_MFMA_TFLOPS
            rl["softplus_traffic"] = sp_traffic
        if denoise is not None:
            out["motion_denoise_config4"] = denoise
        if h16_ref is not None:
            out["fp16_checkpoint"] = h16_ref
        if world == 1 and not args.no_gpu_torch_baseline:
            gt = gpu_torch_baseline(args.act, sd, B, args.proj_steps, dev)
            gt["speedup_of_value"] = out["value"] / gt["value"]
            if fp32_ref is not None:
                gt["speedup_of_fp32_exact"] = fp32_ref["poses_per_s_per_gpu"] / gt["value"]
            out["gpu_torch_baseline"] = gt
            rl["gpu_torch_poses_per_s"] = gt["value"]      # (scalars again: the >= 10x denominator stays with the record)
            rl["speedup_vs_gpu_torch"] = gt["speedup_of_value"]
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.act, sd, args.proj_steps, args.cpu_budget)
        if world == 1 and diag:
            # the library's own host twins (pndf_*_cpu, plain C++; what a `train.device: cpu` config runs) on the same sample:
            # product code beside the reference's CPU path, informational -- never `value`, never the baseline
            cfg_h = amass_config(args.act, "cpu")
            host = PoseNDF(cfg_h)
            host.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
            qh = torch.from_numpy(synth.make_poses(4096, seed=1234))
            host.project(qh, steps=2)
            t0 = time.perf_counter()
            host.project(qh, steps=10)
            dt = (time.perf_counter() - t0) * args.proj_steps / 10
            out["host_twin"] = {"what": "pndf_project_cpu (posendf_amd/csrc/pndf_cpu.cpp, fp32, std::threads), B=4096 x 10 steps "
                                        f"scaled to {args.proj_steps}", "value": 4096 / dt, "unit": "projected poses/s",
                                "threads": os.environ.get("PNDF_CPU_THREADS", "all hardware threads")}
        json_line = json.dumps(out)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        os.write(json_fd, (json_line + "\n").encode())
    os.close(json_fd)


if __name__ == "__main__":
    main()
