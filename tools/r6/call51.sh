#!/bin/bash
# round 6, GPU call 51: the profile of record on the round's FINAL build (precision bf16 added: tools/profile_round.sh refreshes profiles/traffic.json),
# a fresh held-out sweep (seeds never run before), the whole GPU suite with its gate numbers
set -u
OUT=gpurun_out/r6_final
mkdir -p $OUT
bash tools/profile_round.sh $OUT > $OUT/profile_round.log 2>&1
echo "profile_round rc=$?"; tail -5 $OUT/profile_round.log
PNDF_SWEEP_HELDOUT=5 timeout 900 python -m pytest tests/test_gpu_sweep.py -m gpu -q -s > $OUT/sweep_heldout5.txt 2>&1
echo "heldout sweep rc=$?"; tail -3 $OUT/sweep_heldout5.txt
timeout 1500 python -m pytest tests -m gpu -q -s > $OUT/gates.txt 2>&1
echo "suite rc=$?"; tail -4 $OUT/gates.txt
timeout 600 python -m pytest tests -m gpu -q --durations=15 -x -k "not sweep" > $OUT/pytest_timing.txt 2>&1
tail -25 $OUT/pytest_timing.txt | head -22
cat $OUT/bench_driver_cmd.time
