#!/bin/bash
# two SQ passes over tools/bench_lbs.py --seqs 512 --terms-only (through gpurun): clock / MFMA busy, then issue / wait / LDS
# counters of the fused body-model pass.  Usage: bash tools/profile_lbs_quick.sh [outdir] [f16x3|fp32]
set -u
OUT=${1:-gpurun_out/prof_lbs_q}
PREC=${2:-f16x3}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/tools/bench_lbs.py --seqs 512 --reps 2 --lbs-precision $PREC --terms-only"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_MFMA --output-format csv -d "$ROOT/$OUT/pmc1" -o pmc -- $CMD > "$ROOT/$OUT/pmc1.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_INST_LDS --output-format csv -d "$ROOT/$OUT/pmc2" -o pmc -- $CMD > "$ROOT/$OUT/pmc2.log" 2>&1
for p in 1 2; do
    grep -E "Counter_Name|pndf_lbs_vertex_(split_)?terms_kernel\"" "$ROOT/$OUT/pmc$p/pmc_counter_collection.csv" > "$ROOT/$OUT/lbs_pmc_quick$p.csv"
done
python3 - "$ROOT/$OUT/lbs_pmc_quick1.csv" "$ROOT/$OUT/lbs_pmc_quick2.csv" <<'PY'
import csv, sys, collections
for f in sys.argv[1:]:
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(list)
    for r in rows:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
    ms = sum(dur) / max(len(dur), 1) / 1e6
    print("launches", len(dur) // max(len(agg), 1), "avg ms %.3f" % ms)
    for k, v in agg.items():
        print(f"  {k:26s} {sum(v) / len(v):.4g}")
    if "GRBM_GUI_ACTIVE" in agg:
        g = sum(agg["GRBM_GUI_ACTIVE"]) / len(agg["GRBM_GUI_ACTIVE"]) / 8
        b = sum(agg["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(agg["SQ_VALU_MFMA_BUSY_CYCLES"]) / 1024
        print("  shader clock %.2f GHz, MFMA busy %.1f %%" % (g / ms / 1e6, 100 * b / g))
PY
