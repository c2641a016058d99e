#!/bin/bash
# round 6, GPU call 23: why the tile reads spread over rounds 0 - 1 (v10) cost 75 cycles each: LDS counters of the bare loop (arm h7)
set -u
OUT=gpurun_out/r6_23
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/$OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/tools/bench_generic.py 1"
for v in h7; do
  export PNDF_LIBRARY=$ROOT/gpurun_ab/lib_$v.so
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS --output-format csv -d $ROOT/$OUT/${v}_a -o pmc -- $CMD > $ROOT/$OUT/${v}_a.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_ACTIVE_INST_VALU --output-format csv -d $ROOT/$OUT/${v}_b -o pmc -- $CMD > $ROOT/$OUT/${v}_b.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_BRANCH --output-format csv -d $ROOT/$OUT/${v}_c -o pmc -- $CMD > $ROOT/$OUT/${v}_c.log 2>&1
done
cd $ROOT
for f in $(find $OUT -name "*counter_collection.csv" | sort); do echo "== $f"; python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    if 'generic' in r.get('Kernel_Name', ''):
        agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in agg.items():
    print(f"  {k:36s} launches {len(v):3d}  mean per launch {sum(v)/len(v):.6g}")
PY
done
for f in $OUT/*.log; do echo $f; tail -n 2 $f; done
