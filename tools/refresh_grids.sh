#!/bin/bash
# Regenerates the secondary profile files (on the GPU box, repo root): band grid, dispatch grid, C3/C4/C5 counters and bench lines.
#   bash tools/refresh_grids.sh <git sha>; outputs in gpurun_out/grids/
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/grids
mkdir -p $O
export PYTHONPATH=$R
cd $R
{ echo "# C2"; python3 tools/bench_k.py C2 1e6; echo "# C3"; python3 tools/bench_k.py C3 1e6; } > $O/greedy_band_grid.txt 2>&1
echo "band grid done"
python3 tools/bench_cliffs.py C2 1e6 > $O/dispatch_grid_c2.txt 2>&1
echo "dispatch grid done"
for w in C3 C4 C5; do
  python3 tools/pmc_collect.py --head ${1:-unknown} --workload $w --steps 3 > $O/pmc_$w.log 2>&1
  echo "$w counters done"
done
for w in C3 C4 C5; do
  python3 bench.py --workload $w --pairs 10000000 --steps 10 --warmup 2 --cpu-sample 100000 > $O/bench_${w}_1e7.json 2> $O/bench_${w}.err
  echo "bench $w done"
done
