#!/bin/bash
# After `gpurun -- bash tools/final_profiles.sh <sha>`: copy the judged summaries from gpurun_out/final/ into profiles/.
set -e
cd "$(dirname "$0")/.."
F=gpurun_out/final
cp $F/r03_pmc.json profiles/r03_pmc.json
cp $F/valu_rates.txt profiles/r03_valu_rates.txt
cp $F/bench.json profiles/r03_bench_c2.json
S=$(ls $F/stats/*/*kernel_stats.csv $F/stats/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$S" ] && cp "$S" profiles/r03_kernel_stats_c2.csv
ls -la profiles/r03_*
