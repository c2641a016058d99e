#!/bin/bash
# Every measurement of a round from ONE box (run through gpurun from the repo root): bench line, rocprofv3 kernel stats and
# PMC passes of the three fused kernels, region timing, package power, LBS kernel stats.  usage: profile_round.sh [out_dir]
set -u
OUT=${1:-gpurun_out/round}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$ROOT/$OUT"
cd "$ROOT"
for cfg in "f16x3 lrelu" "f16x3 softplus" "fp32 lrelu"; do
    set -- $cfg
    bash tools/gpu_profile.sh "$OUT/prof_$1_$2" "$1" "$2" > "$OUT/prof_$1_$2.log" 2>&1
    echo "profile $1 $2 rc=$?"
done
cd "$ROOT"
# the HBM bytes of THIS build into profiles/traffic.json before the bench line reads it (else the line says traffic_stale)
W="(B=65536 x 100 steps)"
bash tools/collect_profiles.sh "$OUT/prof_f16x3_lrelu" "$OUT/collected/f16x3" pndf_fused_split_relu_kernel "bench.py --precision f16x3 --act lrelu $W" > /dev/null
bash tools/collect_profiles.sh "$OUT/prof_f16x3_softplus" "$OUT/collected/f16x3_softplus" pndf_fused_split_softplus_kernel "bench.py --precision f16x3 --act softplus $W" > /dev/null
bash tools/collect_profiles.sh "$OUT/prof_fp32_lrelu" "$OUT/collected/fp32" pndf_fused_relu_kernel "bench.py --precision fp32 --act lrelu $W" > /dev/null
cp profiles/traffic.json "$OUT/traffic.json"
# the driver's own command (the default, slim line) with its wall time, then the same with every analysis block
( time python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver_cmd.json" 2> "$OUT/bench_driver_cmd.err" ) 2> "$OUT/bench_driver_cmd.time"
echo "bench (driver command) rc=$?"
python bench.py --steps 20 --warmup 5 --diagnostics > "$OUT/bench_head.json" 2> "$OUT/bench_head.err"
echo "bench --diagnostics rc=$?"
python tools/bench_generic.py > "$OUT/generic_arch.jsonl" 2> "$OUT/generic_arch.err"
python tools/gpu_region_timing.py 3 f16x3 lrelu > "$OUT/regions_f16x3_lrelu.txt" 2>&1
python tools/gpu_region_timing.py 3 f16x3 softplus > "$OUT/regions_f16x3_softplus.txt" 2>&1
python tools/gpu_region_timing.py 3 fp32 lrelu > "$OUT/regions_fp32_lrelu.txt" 2>&1
# energy per launch and throttler residency of the four fused kernels (firmware accumulators, tools/power_window.py)
python tools/power_window.py > "$OUT/power_window.txt" 2> "$OUT/power_window.err"
# package power and shader clock while the projection loop runs (rocm-smi sampled every 0.6 s)
smi() { rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk|Package Power" | sed 's/GPU\[0\]\s*: //' | tr '\n' ' '; echo; }
{
    echo "# rocm-smi --showpower --showclocks sampled every 0.6 s while python bench.py loops over project()"
    echo "== idle"; smi
    for cfg in "f16x3 lrelu" "f16x3 softplus" "fp32 lrelu"; do
        set -- $cfg
        python bench.py --steps 150 --warmup 1 --no-cpu-baseline --no-fp32-ref --no-gpu-torch-baseline --no-parity-sample --no-motion-denoise --precision $1 --act $2 > /dev/null 2>&1 &
        pid=$!
        sleep 9
        echo "== $1 $2 project loop running"
        for i in 1 2 3 4 5 6 7 8; do smi; sleep 0.6; done
        wait $pid
    done
} > "$OUT/power_smi.txt" 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/prof_lbs" -o lbs -- python "$ROOT/tools/bench_lbs.py" --seqs 512 --reps 3 > "$ROOT/$OUT/lbs_bench_512x300.json" 2> "$ROOT/$OUT/lbs_prof.err"
cd "$ROOT"
python tools/bench_lbs.py --seqs 64 --reps 5 > "$OUT/lbs_bench_64x300.json" 2> "$OUT/lbs64.err"
python tools/bench_denoise.py --seqs 512 > "$OUT/denoise_512.json" 2> "$OUT/denoise512.err"
ls "$OUT"
