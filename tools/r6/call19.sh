#!/bin/bash
# round 6, GPU call 19: counters of the bare MFMA + tile-read loop of the runtime-planned kernel (arm h7: no epilogue memory, no ring
# events) beside the product form -- clock, MFMA busy, waits, instruction fetch
set -u
OUT=gpurun_out/r6_19
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/$OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > $ROOT/$OUT/sq_counters.txt
CMD="python $ROOT/tools/bench_generic.py 1"
for v in product h7; do
  if [ $v = product ]; then unset PNDF_LIBRARY; else export PNDF_LIBRARY=$ROOT/gpurun_ab/lib_$v.so; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/${v}_trace -o trace -- $CMD > $ROOT/$OUT/${v}_trace.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $ROOT/$OUT/${v}_sq -o pmc -- $CMD > $ROOT/$OUT/${v}_sq.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $ROOT/$OUT/${v}_sq2 -o pmc -- $CMD > $ROOT/$OUT/${v}_sq2.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INST_LEVEL_VMEM --output-format csv -d $ROOT/$OUT/${v}_sq3 -o pmc -- $CMD > $ROOT/$OUT/${v}_sq3.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_BRANCH SQ_IFETCH SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC --output-format csv -d $ROOT/$OUT/${v}_sq4 -o pmc -- $CMD > $ROOT/$OUT/${v}_sq4.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $ROOT/$OUT/${v}_sq5 -o pmc -- $CMD > $ROOT/$OUT/${v}_sq5.log 2>&1
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
grep generic $OUT/*_trace/trace_kernel_stats.csv | head -4
tail -3 $OUT/*sq4.log $OUT/*sq5.log | head -40
