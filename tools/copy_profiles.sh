#!/bin/bash
# After `gpurun -- bash tools/final_profiles.sh <sha>`: copy the judged summaries from gpurun_out/final/ into profiles/.
set -e
cd "$(dirname "$0")/.."
F=gpurun_out/final
cp $F/r04_pmc.json profiles/r04_pmc.json
cp $F/valu_rates.txt profiles/r04_valu_rates.txt
cp $F/bench.json profiles/r04_bench_c2.json
S=$(ls $F/stats/*/*kernel_stats.csv $F/stats/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$S" ] && cp "$S" profiles/r04_kernel_stats_c2.csv
ls -la profiles/r04_*
