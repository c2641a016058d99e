#!/bin/bash
# rocprofv3 passes for the dominant kernel (run on the GPU box from the repo root):
#   1. kernel trace + stats (durations)      2. SQ counters (MFMA busy, wave cycles, clocks)
#   3. FETCH_SIZE                            4. WRITE_SIZE        (separate passes: TCC has 4 slots)
# Counters are collected with --kernel-trace only, as the pool requires.
set -u
OUT=${1:-gpurun_out/prof}
PREC=${2:-f16x3}
ACT=${3:-lrelu}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fp32-ref --no-gpu-torch-baseline --no-parity-sample --no-motion-denoise --precision $PREC --act $ACT"
rocprofv3 -L > "$ROOT/$OUT/counters_available.txt" 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/trace" -o trace -- $BENCH > "$ROOT/$OUT/trace.log" 2>&1
echo "trace rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d "$ROOT/$OUT/pmc_sq" -o pmc -- $BENCH > "$ROOT/$OUT/pmc_sq.log" 2>&1
echo "pmc_sq rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d "$ROOT/$OUT/pmc_sq2" -o pmc -- $BENCH > "$ROOT/$OUT/pmc_sq2.log" 2>&1
echo "pmc_sq2 rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$ROOT/$OUT/pmc_fetch" -o pmc -- $BENCH > "$ROOT/$OUT/pmc_fetch.log" 2>&1
echo "pmc_fetch rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$ROOT/$OUT/pmc_write" -o pmc -- $BENCH > "$ROOT/$OUT/pmc_write.log" 2>&1
echo "pmc_write rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d "$ROOT/$OUT/pmc_l2" -o pmc -- $BENCH > "$ROOT/$OUT/pmc_l2.log" 2>&1
echo "pmc_l2 rc=$?"
find "$ROOT/$OUT" -name "*.csv" | head -40
for f in $(find "$ROOT/$OUT" -name "*counter_collection.csv"); do echo "== $f"; grep "pndf_fused" "$f" | head -12; done
