#!/bin/bash
# round 6, GPU call 7: where does the runtime-planned kernel (v3) wait?  rocprofv3 counters on the amass.yaml arm
set -u
OUT=gpurun_out/r6_14
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/$OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/tools/bench_generic.py 1"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/trace -o trace -- $CMD > $ROOT/$OUT/trace.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $ROOT/$OUT/pmc_sq -o pmc -- $CMD > $ROOT/$OUT/pmc_sq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $ROOT/$OUT/pmc_sq2 -o pmc -- $CMD > $ROOT/$OUT/pmc_sq2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INST_LEVEL_VMEM --output-format csv -d $ROOT/$OUT/pmc_sq3 -o pmc -- $CMD > $ROOT/$OUT/pmc_sq3.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $ROOT/$OUT/pmc_l2 -o pmc -- $CMD > $ROOT/$OUT/pmc_l2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum --output-format csv -d $ROOT/$OUT/pmc_l1 -o pmc -- $CMD > $ROOT/$OUT/pmc_l1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $ROOT/$OUT/pmc_fetch -o pmc -- $CMD > $ROOT/$OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $ROOT/$OUT/pmc_write -o pmc -- $CMD > $ROOT/$OUT/pmc_write.log 2>&1
cd $ROOT
for f in $(find $OUT -name "*counter_collection.csv"); do echo "== $f"; python - "$f" <<'PY'
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
