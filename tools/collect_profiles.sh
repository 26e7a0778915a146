#!/bin/bash
# Copies the summaries tools/profile_round.sh <round> left under gpurun_out/ into profiles/ (tracked): tools/collect_profiles.sh r02
R=${1:-r04}
cd "$(dirname "$0")/.."
for t in readme hetero512 hetero4096 hetero4096_8M core c4 c4_10-5-3; do
  d=gpurun_out/prof_${R}_$t
  [ -f $d/summary_kernel_stats.csv ] || { echo "missing $d"; continue; }
  cp $d/summary_kernel_stats.csv profiles/${R}_${t}_kernel_stats.csv
  cp $d/summary_pmc.json profiles/${R}_${t}_pmc.json
  raw=$(ls -t $d/kt/*/*_kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$raw" ] && cp "$raw" profiles/${R}_${t}_rocprofv3_kernel_stats_raw.csv
  for f in $d/traffic_*.json; do [ -f "$f" ] && cp "$f" profiles/; done
done
ls -la profiles/traffic_*.json
