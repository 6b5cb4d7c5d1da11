#!/bin/bash
# Regenerates the judged measurement files of round 4 on the GPU box (repo root), one part per gpurun call (20-minute limit):
#   bash tools/r04_profiles.sh c2 <sha>      counters (4 --pmc passes), rocprofv3 kernel stats and the bench line at C2
#   bash tools/r04_profiles.sh c3|c4|c5 <sha> the same at 10^7 pairs of BASELINE config 3 / 4 / 5
#   bash tools/r04_profiles.sh grids <sha>    band grid, dispatch grid, filter stage, host path
# Outputs land in gpurun_out/r04/; `bash tools/r04_profiles.sh copy` (in the build container) copies them into profiles/.
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
PART=${1:-c2}
HEAD=${2:-unknown}
O=$R/gpurun_out/r04
mkdir -p $O
export PYTHONPATH=$R
cd $R
stats() { # rocprofv3 --kernel-trace --stats of one bench command; summary csv -> $O/kernel_stats_$1.csv
  local tag=$1; shift
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$tag -o $tag -- python3 $R/bench.py "$@" \
      > $O/stats_bench_$tag.json 2> $O/stats_$tag.err)
  local f=$(ls $O/stats_$tag/*/*kernel_stats.csv $O/stats_$tag/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && cp "$f" $O/kernel_stats_$tag.csv
  echo "kernel stats $tag done"
}
case $PART in
c2)
  python3 tools/pmc_collect.py --head $HEAD > $O/pmc_collect_c2.log 2>&1; echo "pmc c2 done"
  cp gpurun_out/final/valu_rates.txt $O/valu_rates.txt
  stats c2 --steps 200 --warmup 2 --no-cpu-baseline --no-standalone --no-sequential --no-in-order
  python3 bench.py > $O/bench_c2.json 2> $O/bench_c2.err; echo "bench c2 done"; tail -c 600 $O/bench_c2.json ;;
c3|c4|c5)
  W=$(echo $PART | tr a-z A-Z)
  python3 tools/pmc_collect.py --head $HEAD --workload $W --pairs 10000000 --steps 2 > $O/pmc_collect_$PART.log 2>&1; echo "pmc $PART done"
  stats $PART --workload $W --pairs 10000000 --steps 4 --warmup 1 --no-cpu-baseline --no-standalone --no-sequential --no-in-order
  python3 bench.py --workload $W --pairs 10000000 --steps 10 --warmup 2 --cpu-sample 100000 > $O/bench_${PART}_1e7.json 2> $O/bench_$PART.err
  echo "bench $PART done"; tail -c 400 $O/bench_${PART}_1e7.json ;;
grids)
  { echo "# C2"; python3 tools/bench_k.py C2 1e6; echo "# C3"; python3 tools/bench_k.py C3 1e6; } > $O/greedy_band_grid.txt 2>&1; echo "band grid done"
  python3 tools/bench_cliffs.py C2 1e6 > $O/dispatch_grid_c2.txt 2>&1; echo "dispatch grid done"
  python3 tools/bench_filter.py > $O/filter_bench_c2.txt 2>&1; echo "filter bench done"
  python3 tools/bench_host_path.py > $O/host_path_c2.txt 2>&1; echo "host path done" ;;
copy)
  for w in c2 c3 c4 c5; do
    [ -f $O/kernel_stats_$w.csv ] && cp $O/kernel_stats_$w.csv profiles/r04_kernel_stats_$w.csv
  done
  [ -f $O/bench_c2.json ] && tail -1 $O/bench_c2.json > profiles/r04_bench_c2.json
  for w in c3 c4 c5; do [ -f $O/bench_${w}_1e7.json ] && tail -1 $O/bench_${w}_1e7.json > profiles/r04_bench_${w}_1e7.json; done
  [ -f gpurun_out/final/r04_pmc.json ] && cp gpurun_out/final/r04_pmc.json profiles/r04_pmc.json
  for w in c3 c4 c5; do [ -f gpurun_out/final/r04_pmc_$w.json ] && cp gpurun_out/final/r04_pmc_$w.json profiles/r04_pmc_$w.json; done
  [ -f $O/valu_rates.txt ] && cp $O/valu_rates.txt profiles/r04_valu_rates.txt
  for f in greedy_band_grid dispatch_grid_c2 filter_bench_c2 host_path_c2; do [ -f $O/$f.txt ] && cp $O/$f.txt profiles/r04_$f.txt; done
  ls -la profiles/r04_* ;;
esac
