#!/bin/bash
# round 6, GPU call 50: precision bf16 (pndf_fused_bf16_relu_kernel): its tests, the bench contract, and the driver's command
set -u
OUT=gpurun_out/r6_50
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_bench_contract.py -m gpu -x -q -k "one_term or bf16 or f16_single or bench_json_line or diagnostics" -s > $OUT/bf16_tests.txt 2>&1
echo "tests rc=$?"; grep -E "^(f16|bf16):|passed|failed|Error|assert" $OUT/bf16_tests.txt | cut -c1-300 | head -20
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err
echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r6_50/bench_driver_cmd.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], d["config"]["workload"][:120])
print("bf16", json.dumps(d.get("bf16"))[:900])
print("fp32", d["roofline"].get("fp32_exact_poses_per_s"), "traffic_stale", d["roofline"].get("traffic_stale"))
PY
