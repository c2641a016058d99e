#!/bin/bash
# round 6, GPU call 3: the runtime-planned kernels (tests + bench), the contract tests, then the whole GPU suite
set -u
OUT=gpurun_out/r6_03
mkdir -p $OUT
timeout 900 python -m pytest tests/test_depth.py -m gpu -q -x > $OUT/pytest_depth.txt 2>&1
echo "depth rc=$?"; tail -15 $OUT/pytest_depth.txt
timeout 900 python tools/bench_generic.py > $OUT/generic_arch.jsonl 2> $OUT/generic_arch.err
echo "bench_generic rc=$?"; cat $OUT/generic_arch.jsonl | cut -c1-400
timeout 900 python -m pytest tests/test_bench_contract.py tests/test_cabi.py -q > $OUT/pytest_contract.txt 2>&1
echo "contract rc=$?"; tail -5 $OUT/pytest_contract.txt
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2> $OUT/bench_default.time
echo "bench rc=$?"; tail -3 $OUT/bench_default.time
timeout 1200 python -m pytest tests -m gpu -q --deselect tests/test_depth.py --deselect tests/test_bench_contract.py > $OUT/pytest_gpu.txt 2>&1
echo "suite rc=$?"; tail -5 $OUT/pytest_gpu.txt
