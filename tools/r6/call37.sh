#!/bin/bash
# round 6, GPU call 37: (non-temporal scratch loads adopted) counters of the split-precision runtime-planned kernel (amass.yaml dims, f16x3)
set -u
OUT=gpurun_out/r6_37
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/$OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/tools/bench_generic.py 11"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/trace -o trace -- $CMD > $ROOT/$OUT/trace.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $ROOT/$OUT/sq -o pmc -- $CMD > $ROOT/$OUT/sq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $ROOT/$OUT/sq2 -o pmc -- $CMD > $ROOT/$OUT/sq2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_INSTS_SALU --output-format csv -d $ROOT/$OUT/sq3 -o pmc -- $CMD > $ROOT/$OUT/sq3.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $ROOT/$OUT/l2 -o pmc -- $CMD > $ROOT/$OUT/l2.log 2>&1
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
grep generic $OUT/trace/trace_kernel_stats.csv | head -3
