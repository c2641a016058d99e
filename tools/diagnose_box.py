#!/usr/bin/env python3
"""Read a bench.py line (or the driver's BENCH_rNN.json wrapped around one) and say, from the line's own `box` / `regions`
blocks, why the number is what it is: a chip that needs more joules per launch, a weight ring that waits for its fetches, a
memory system with other latencies, a throttler other than package power -- against the ranges of the round-5 boxes
(profiles/r05/).  usage: python tools/diagnose_box.py <file.json> [...]"""
import json
import sys

# what the boxes of round 5 showed (profiles/r05/bench*.json, ring_margin.txt, power_window*.txt)
REF = {"kernel_ms": (84.4, 91.2), "energy_j": (113.0, 120.0), "package_w": (1290.0, 1385.0),
       "ring_wait": (10.0, 13.0), "ring_barrier": (26.0, 32.0), "fp32_wait": (15.0, 19.0), "cycles": (470e3, 485e3),
       "l2_ns": (205.0, 230.0), "mall_ns": (220.0, 232.0), "hbm_ns": (338.0, 352.0), "read_gbps": (6100.0, 6700.0),
       "ring_only_gbps": (118.0, 125.0)}


def load(path):
    with open(path) as f:
        text = f.read().strip()
    try:
        d = json.loads(text)
    except ValueError:                                          # a log with the line at its end
        d = json.loads([ln for ln in text.splitlines() if ln.startswith("{")][-1])
    if "metric" in d:
        return d
    tail = ((d.get("run") or {}).get("stdout_tail")) or ""      # the driver's wrapper: the line is a JSON object inside the tail
    for line in tail.splitlines():
        if line.startswith('{"metric"'):
            return json.loads(line)
    raise SystemExit(f"{path}: no bench line found")


def rng(v, key):
    lo, hi = REF[key]
    return "within" if lo <= v <= hi else ("BELOW" if v < lo else "ABOVE")


def main():
    for path in sys.argv[1:]:
        d = load(path)
        roof, box, reg = d.get("roofline", {}), d.get("box"), d.get("regions")
        k = roof.get("kernel_ms")
        print(f"== {path}: {d.get('value', 0):,.0f} {d.get('unit')}  ({roof.get('kernel')}: {k:.2f} ms per launch, frac {roof.get('frac', 0):.3f}; "
              f"hwmon {roof.get('sclk_mhz')} MHz, {roof.get('package_w')} W)")
        print(f"   kernel time {rng(k, 'kernel_ms')} the round-5 range {REF['kernel_ms']}")
        if not box or not reg or "error" in reg:
            print("   no `box` / `regions` blocks in this line (written before round 5, N > 1, or --no-box): nothing to diagnose from")
            continue
        findings = []
        pw = box.get("power_window") or {}
        if "error" not in pw and pw:
            e, w, ppt = pw["energy_j_per_launch"], pw["mean_package_w"], pw.get("ppt_limited_frac")
            print(f"   energy {e:.1f} J per launch ({rng(e, 'energy_j')} {REF['energy_j']}), package {w:.0f} W ({rng(w, 'package_w')}), power limiter active "
                  f"{100 * (ppt or 0):.0f} % of the time; thermal {pw.get('socket_thermal_limited_frac')}, HBM thermal {pw.get('hbm_thermal_limited_frac')}, "
                  f"PROCHOT {pw.get('prochot_frac')}; hotspot {pw.get('hotspot_c')} C")
            if e > REF["energy_j"][1]:
                findings.append(f"a power-limited chip that spends {e:.0f} J on the launch most boxes do in 114-119 J (seen in round 5: up to 125): "
                                f"time = energy / {w:.0f} W")
            if any((pw.get(x) or 0) > 0.02 for x in ("socket_thermal_limited_frac", "vr_thermal_limited_frac", "hbm_thermal_limited_frac", "prochot_frac")):
                findings.append("a THERMAL / PROCHOT limiter was active: the box is hot, not the kernel slow")
            if w < REF["package_w"][0] and (ppt or 0) < 0.05 and k > REF["kernel_ms"][1]:
                findings.append("slow AND below the power limit with the limiter idle: the matrix pipe is waiting (see the ring and the memory probe)")
        else:
            print(f"   power window: {pw}")
        ring, f32 = reg["ring"], (reg.get("fp32_kernel") or {}).get("ring", {})
        print(f"   instrumented step {reg['cycles_per_wave_step']:,.0f} cycles ({rng(reg['cycles_per_wave_step'], 'cycles')} {REF['cycles']}); weight ring: "
              f"{ring['wait_cycles_per_slot']:.1f} cycles per slot in the DMA wait ({rng(ring['wait_cycles_per_slot'], 'ring_wait')} {REF['ring_wait']}), "
              f"{ring['barrier_cycles_per_slot']:.1f} in the barrier ({rng(ring['barrier_cycles_per_slot'], 'ring_barrier')}); fp32 kernel wait "
              f"{f32.get('wait_cycles_per_slot', float('nan')):.1f}")
        if ring["wait_cycles_per_slot"] > 2 * REF["ring_wait"][1]:
            both = f32 and f32.get("wait_cycles_per_slot", 0) > 2 * REF["fp32_wait"][1]
            findings.append("the split kernel's ring WAITS for its fetches" + (" and so does the fp32 kernel's (2,200-cycle slots): not a latency margin problem, "
                            "the stream is not being delivered" if both else ": fetch latency above the look-ahead (the fp32 kernel, with 7 x longer slots, is fine)"))
        mp = box.get("mem_probe") or {}
        if mp and "error" not in mp:
            print(f"   memory: L2 hit {mp['l2_hit_latency_ns']} ns ({rng(mp['l2_hit_latency_ns'], 'l2_ns')}), Infinity Cache {mp['infinity_cache_latency_ns']} ns "
                  f"({rng(mp['infinity_cache_latency_ns'], 'mall_ns')}), HBM {mp['hbm_latency_ns']} ns ({rng(mp['hbm_latency_ns'], 'hbm_ns')}), read {mp['stream_read_gbps']} GB/s "
                  f"({rng(mp['stream_read_gbps'], 'read_gbps')})"
                  + (f", ring alone {mp['ring_only_gbps_per_cu']} GB/s per CU ({rng(mp['ring_only_gbps_per_cu'], 'ring_only_gbps')}; the f16x3 kernel consumes 51)"
                     if mp.get("ring_only_gbps_per_cu") is not None else ""))
            for key, ref, name in (("l2_hit_latency_ns", "l2_ns", "L2-hit"), ("infinity_cache_latency_ns", "mall_ns", "Infinity-Cache"), ("hbm_latency_ns", "hbm_ns", "HBM")):
                if mp[key] > 1.15 * REF[ref][1]:
                    findings.append(f"{name} latency {mp[key]} ns is above every box of round 5 ({REF[ref]})")
            if mp.get("ring_only_gbps_per_cu", 1e9) < 0.8 * REF["ring_only_gbps"][0]:
                findings.append(f"the weight ring alone is fed at {mp['ring_only_gbps_per_cu']} GB/s per CU (round 5: 121-122): the box cannot deliver the stream")
        print(f"   partitions {box.get('current_compute_partition')} / {box.get('current_memory_partition')}, {box.get('compute_units')} CUs, cap {box.get('power_cap_w')} W, "
              f"mclk {(box.get('pp_dpm_mclk') or {}).get('current')}, fclk {(box.get('pp_dpm_fclk') or {}).get('current')}, host load {box.get('host_loadavg')}")
        if box.get("current_compute_partition") not in (None, "SPX") or box.get("current_memory_partition") not in (None, "NPS1"):
            findings.append("the device is not in SPX / NPS1 mode")
        if box.get("compute_units") and box["compute_units"] != 256:
            findings.append(f"{box['compute_units']} compute units: 1,024 workgroups no longer make four even rounds")
        print("   => " + ("; ".join(findings) if findings else "nothing out of the round-5 ranges: an ordinary box"))


if __name__ == "__main__":
    main()
