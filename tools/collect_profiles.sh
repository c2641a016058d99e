#!/bin/bash
# Copy the summaries of one tools/gpu_profile.sh run (under gpurun_out/) into profiles/<round>/<tag>/ and refresh
# the kernel's entry of profiles/traffic.json.   usage: collect_profiles.sh <gpurun_out/prof_dir> <profiles/rNN/tag> <kernel> <workload text>
set -eu
SRC=$1; DST=$2; KERNEL=$3; WORK=$4
mkdir -p "$DST"
cp "$SRC/trace/trace_kernel_stats.csv" "$DST/kernel_stats.csv"
for p in sq sq2 fetch write l2; do
    grep -E "Counter_Name|$KERNEL\"" "$SRC/pmc_$p/pmc_counter_collection.csv" > "$DST/pmc_${p}_counters.csv"
done
python "$(dirname "$0")/make_traffic.py" "$KERNEL" "$DST/pmc_fetch_counters.csv" "$DST/pmc_write_counters.csv" "$WORK"
head -3 "$DST/kernel_stats.csv"
