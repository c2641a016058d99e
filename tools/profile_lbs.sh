#!/bin/bash
# PMC passes for the body-model kernels (run through gpurun from the repo root): MFMA busy, waits, LDS bank conflicts, fabric
# traffic of `tools/bench_lbs.py --seqs 512`.  Counters with --kernel-trace only, separate passes (pool rule).
set -u
OUT=${1:-gpurun_out/prof_lbs}
PREC=${2:-f16x3}          # arithmetic of the body-model passes: f16x3 | fp32
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/tools/bench_lbs.py --seqs 512 --reps 2 --lbs-precision $PREC"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d "$ROOT/$OUT/pmc_sq" -o pmc -- $CMD > "$ROOT/$OUT/pmc_sq.log" 2>&1
echo "pmc_sq rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d "$ROOT/$OUT/pmc_sq2" -o pmc -- $CMD > "$ROOT/$OUT/pmc_sq2.log" 2>&1
echo "pmc_sq2 rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$ROOT/$OUT/pmc_fetch" -o pmc -- $CMD > "$ROOT/$OUT/pmc_fetch.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$ROOT/$OUT/pmc_write" -o pmc -- $CMD > "$ROOT/$OUT/pmc_write.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d "$ROOT/$OUT/pmc_l2" -o pmc -- $CMD > "$ROOT/$OUT/pmc_l2.log" 2>&1
for p in sq sq2 fetch write l2; do
    grep -E "Counter_Name|pndf_lbs_vertex_(split_)?terms_kernel\"" "$ROOT/$OUT/pmc_$p/pmc_counter_collection.csv" > "$ROOT/$OUT/lbs_pmc_${p}_counters.csv"
done
ls "$ROOT/$OUT"
