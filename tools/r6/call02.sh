#!/bin/bash
# round 6, GPU call 2: XCD stagger arm (lrelu), softplus scratch-footprint arms, the slimmed bench line, contract tests
set -u
OUT=gpurun_out/r6_02
mkdir -p $OUT
P=posendf_amd/lib/libposendf_amd.so
{ echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>&1)"; echo "cfs_quota: $(cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>&1) / $(cat /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>&1)"; echo "nproc: $(nproc)"; cat /proc/loadavg; } > $OUT/host.txt 2>&1
python tools/ab_bench.py --rounds 3 product=$P stagger3=gpurun_ab/lib_stagger3.so stagger6=gpurun_ab/lib_stagger6.so > $OUT/stagger_ab.txt 2>&1
python tools/power_window.py --libs product=$P stagger3=gpurun_ab/lib_stagger3.so stagger6=gpurun_ab/lib_stagger6.so f16x3:lrelu > $OUT/stagger_power.jsonl 2> $OUT/stagger_power.err
python tools/ab_bench.py --act softplus --rounds 2 product=$P spwrap154=gpurun_ab/lib_spwrap154.so spwrap103=gpurun_ab/lib_spwrap103.so spwrap52=gpurun_ab/lib_spwrap52.so > $OUT/spwrap_ab.txt 2>&1
python tools/power_window.py --libs product=$P spwrap154=gpurun_ab/lib_spwrap154.so spwrap103=gpurun_ab/lib_spwrap103.so spwrap52=gpurun_ab/lib_spwrap52.so f16x3:softplus > $OUT/spwrap_power.jsonl 2> $OUT/spwrap_power.err
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2> $OUT/bench_default.time
echo "bench rc=$?"; tail -3 $OUT/bench_default.time
python -m pytest tests/test_bench_contract.py tests/test_cabi.py tests/test_sharding_gloo.py tests/test_power_window.py -q > $OUT/pytest_contract.txt 2>&1
echo "pytest rc=$?"; tail -5 $OUT/pytest_contract.txt
cat $OUT/host.txt $OUT/stagger_ab.txt $OUT/spwrap_ab.txt
