#!/bin/bash
# round 6, GPU call 4: runtime-planned kernels v2 (resident accumulators per pass), softplus 12-byte arm
set -u
OUT=gpurun_out/r6_04
mkdir -p $OUT
P=posendf_amd/lib/libposendf_amd.so
timeout 900 python -m pytest tests/test_depth.py -m gpu -q -x > $OUT/pytest_depth.txt 2>&1
echo "depth rc=$?"; tail -6 $OUT/pytest_depth.txt
timeout 900 python tools/bench_generic.py > $OUT/generic_arch.jsonl 2> $OUT/generic_arch.err
echo "bench_generic rc=$?"; python - <<'PY'
import json
for l in open('gpurun_out/r6_04/generic_arch.jsonl'):
    d=json.loads(l); print(d.get('arm'), d.get('kernel'), round(d.get('ms',0),2), 'ms', round(d.get('frac_of_fp32_mfma_peak',0),3), d.get('error','')[:300])
PY
python tools/ab_bench.py --act softplus --rounds 3 product=$P sp12=gpurun_ab/lib_sp12.so > $OUT/sp12_ab.txt 2>&1
python tools/power_window.py --libs product=$P sp12=gpurun_ab/lib_sp12.so product2=$P f16x3:softplus > $OUT/sp12_power.jsonl 2> $OUT/sp12_power.err
cat $OUT/sp12_ab.txt; cut -c1-330 $OUT/sp12_power.jsonl
