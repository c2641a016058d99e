"""bench.py's reading of `amd-smi metric --json` (the firmware's energy / throttler accumulators behind `box.power_window`):
the parser against a reading captured on an MI355X box, and the rates computed from two readings.  No GPU, no amd-smi."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402

READING = {"gpu_data": [{"gpu": 0,
                         "power": {"socket_power": {"value": 1340, "unit": "W"}, "throttle_status": "N/A"},
                         "clock": {"gfx_0": {"clk": {"value": 1967, "unit": "MHz"}}},
                         "temperature": {"edge": "N/A", "hotspot": {"value": 53, "unit": "C"}, "mem": {"value": 38, "unit": "C"}},
                         "energy": {"total_energy_consumption": {"value": 135440564.988, "unit": "J"}},
                         "throttle": {"accumulation_counter": 491954292, "prochot_accumulated": 0, "ppt_accumulated": 563360,
                                      "socket_thermal_accumulated": 0, "vr_thermal_accumulated": 0, "hbm_thermal_accumulated": 0,
                                      "gfx_clk_below_host_limit_accumulated": "N/A",
                                      "gfx_clk_below_host_limit_power_accumulated": {"xcp_0": [10, 10, 10, 10, 10, 10, 10, 10]},
                                      "total_gfx_clk_below_host_limit_accumulated": {"xcp_0": [100, 100, 100, 100, 100, 100, 100, "N/A"]}}}]}


def _fake_run(readings):
    it = iter(readings)

    class R:
        def __init__(self, out):
            self.stdout = out

    def run(*a, **k):
        return R("WARNING: a banner line the tool may print\n" + json.dumps(next(it)))
    return run


def test_metric_parser(monkeypatch):
    monkeypatch.setattr(subprocess, "run", _fake_run([READING]))
    m = bench._amdsmi_metric()
    assert m.pop("below_limit_power") == 80.0 and m.pop("below_limit_total") == 700.0
    assert m.pop("below_limit_thermal") is None and m.pop("low_utilization") is None
    assert m == {"energy_j": 135440564.988, "acc": 491954292.0, "ppt": 563360.0, "prochot": 0.0, "socket_thm": 0.0, "vr_thm": 0.0,
                 "hbm_thm": 0.0, "socket_power_w": 1340.0, "gfx_clk_mhz": 1967.0, "hotspot_c": 53.0, "mem_c": 38.0}
    monkeypatch.setattr(subprocess, "run", lambda *a, **k: (_ for _ in ()).throw(OSError("no such tool")))
    assert bench._amdsmi_metric() is None


def test_power_window_rates(monkeypatch):
    second = json.loads(json.dumps(READING))
    g = second["gpu_data"][0]
    g["energy"]["total_energy_consumption"]["value"] += 1350.0 * 3.0         # 3 s at 1,350 W
    g["throttle"]["accumulation_counter"] += 3000                             # the ~1 kHz sample counter
    g["throttle"]["ppt_accumulated"] += 1800                                  # limiter active in 60 % of the samples
    g["throttle"]["gfx_clk_below_host_limit_power_accumulated"]["xcp_0"] = [10 + 1500] * 8      # every XCD: half of the samples
    monkeypatch.setattr(subprocess, "run", _fake_run([READING, second]))
    launches = []
    r = bench.power_window(lambda: launches.append(1), lambda: None, kernel_ms=88.0, min_s=0.05, max_s=5.0)
    assert abs(r["window_s"] - 3.0) < 1e-9 and abs(r["mean_package_w"] - 1350.0) < 1e-6
    assert abs(r["energy_j_per_launch"] - 1350.0 * 0.088) < 1e-6 and abs(r["ppt_limited_frac"] - 0.6) < 1e-9
    assert abs(r["xcd_clk_below_limit_frac"]["below_limit_power"] - 0.5) < 1e-9 and r["xcd_clk_below_limit_frac"]["below_limit_thermal"] is None
    assert r["socket_thermal_limited_frac"] == 0.0 and r["prochot_frac"] == 0.0 and launches
    monkeypatch.setattr(subprocess, "run", lambda *a, **k: (_ for _ in ()).throw(OSError("no such tool")))
    assert "error" in bench.power_window(lambda: None, lambda: None, kernel_ms=88.0, min_s=0.01, max_s=1.0)


def test_readings_follow_the_pci_address(monkeypatch):
    """on a multi-GPU node amd-smi lists every GPU whatever HIP_VISIBLE_DEVICES says; the readings must come
    from the device the kernel runs on (matched by PCI address), and an unknown address is an error block, never entry 0."""
    listing = [{"gpu": 0, "bdf": "0000:05:00.0"}, {"gpu": 3, "bdf": "0000:72:00.0"}]
    other = json.loads(json.dumps(READING))
    other["gpu_data"][0]["gpu"] = 3
    other["gpu_data"][0]["power"]["socket_power"]["value"] = 777
    calls = []

    class R:
        def __init__(self, out):
            self.stdout = out

    def run(cmd, *a, **k):
        calls.append(cmd)
        if cmd[1] == "list":
            return R(json.dumps(listing))
        assert cmd[1:4] == ["metric", "-g", "3"], cmd
        return R(json.dumps(other))
    monkeypatch.setattr(subprocess, "run", run)
    bench._AMDSMI_INDEX.clear()
    assert bench._amdsmi_gpu_index("0000:72:00.0") == 3
    assert bench._amdsmi_metric("0000:72:00.0")["socket_power_w"] == 777.0
    assert bench._amdsmi_metric("0000:99:00.0") is None
    r = bench.power_window(lambda: None, lambda: None, kernel_ms=88.0, min_s=0.01, max_s=1.0, bdf="0000:99:00.0")
    assert "error" in r and "0000:99:00.0" in r["error"]
    bench._AMDSMI_INDEX.clear()
