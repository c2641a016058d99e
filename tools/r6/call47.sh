#!/bin/bash
# round 6, GPU call 47: Softplus networks with a layer of a few units take the runtime-planned split kernels
set -u
OUT=gpurun_out/r6_47
mkdir -p $OUT
timeout 900 python -m pytest tests/test_depth.py tests/test_gpu_parity.py tests/test_noenc.py -m gpu -q > $OUT/pytest.txt 2>&1
echo "rc=$?"; grep -E "^FAILED|passed|failed" $OUT/pytest.txt | tail -8
timeout 900 python tools/r6/fused_narrow.py 2>/dev/null | grep softplus | grep f16x3
