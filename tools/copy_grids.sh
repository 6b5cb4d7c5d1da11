#!/bin/bash
# After `gpurun -- bash tools/refresh_grids.sh <sha>`: copy the secondary profile files from gpurun_out/ into profiles/.
set -e
cd "$(dirname "$0")/.."
G=gpurun_out/grids
cp $G/greedy_band_grid.txt profiles/r04_greedy_band_grid.txt
cp $G/dispatch_grid_c2.txt profiles/r04_dispatch_grid_c2.txt
for w in C3 C4 C5; do tail -1 $G/bench_${w}_1e7.json > profiles/r04_bench_$(echo $w | tr A-Z a-z)_1e7.json; done
for w in c3 c4 c5; do
  cp gpurun_out/final/r04_pmc_$w.json profiles/r04_pmc_$w.json
  PYTHONPATH=$PWD python3 tools/pmc_collect.py --remix --workload $(echo $w | tr a-z A-Z) | tail -4
done
