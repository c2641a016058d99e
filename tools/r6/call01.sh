#!/bin/bash
# round 6, GPU call 1: baseline suite + the timing / energy arms of VERDICT r5 items 1 and 2 (one box, same-box A/B)
set -u
OUT=gpurun_out/r6_01
mkdir -p $OUT
P=posendf_amd/lib/libposendf_amd.so
amd-smi list --json > $OUT/amdsmi_list.json 2>&1
amd-smi metric -g 0 --json 2>&1 | head -c 3000 > $OUT/amdsmi_metric_head.json
python -m pytest tests -m gpu -q --ignore=tests/test_bench_contract.py > $OUT/pytest_gpu.txt 2>&1
echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.txt
# item 1: what the (lin3^T,lin2^T) / both big phases' slot fetches cost (WRONG results: timing / energy only)
python tools/power_window.py --libs product=$P nofetch_bwd=gpurun_ab/lib_nofetch_bwd.so nofetch_both=gpurun_ab/lib_nofetch_both.so product2=$P f16x3:lrelu > $OUT/shared_slot_power.jsonl 2> $OUT/shared_slot_power.err
python tools/ab_bench.py --rounds 3 product=$P nofetch_bwd=gpurun_ab/lib_nofetch_bwd.so nofetch_both=gpurun_ab/lib_nofetch_both.so > $OUT/shared_slot_ab.txt 2>&1
# item 2 (i): the softplus ring's counted wait relaxed by two operations
python tools/power_window.py --libs product=$P spwait=gpurun_ab/lib_spwait.so product2=$P f16x3:softplus > $OUT/sp_wait_power.jsonl 2> $OUT/sp_wait_power.err
python tools/ab_bench.py --act softplus --rounds 3 product=$P spwait=gpurun_ab/lib_spwait.so > $OUT/sp_wait_ab.txt 2>&1
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2> $OUT/bench_default.time
echo "bench rc=$?"; tail -3 $OUT/bench_default.time
cat $OUT/shared_slot_ab.txt $OUT/sp_wait_ab.txt
