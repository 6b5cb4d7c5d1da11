#!/bin/bash
# Regenerates every number the BENCH line quotes, in one go (on the GPU box, from the repo root):
#   bash tools/final_profiles.sh <git sha of the tree>
# 1. tools/pmc_collect.py  -> profiles/r04_pmc.json (HBM bytes, SQ counters, lane utilisation, ubench issue costs + clock)
# 2. rocprofv3 --kernel-trace --stats of the bench command -> kernel stats csv
# 3. python3 bench.py (reads the json written in step 1)   -> the bench line
# Outputs land in gpurun_out/final/; copy what is to be judged into profiles/ (tools/copy_profiles.sh).
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
HEAD=${1:-unknown}
O=$R/gpurun_out/final
mkdir -p $O
export PYTHONPATH=$R
cd $R
python3 tools/pmc_collect.py --head $HEAD > $O/pmc_collect.log 2>&1
echo "pmc_collect done"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o c2 -- python3 $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-standalone --no-sequential --no-in-order > $O/stats_bench.json 2> $O/stats.err
echo "kernel stats done"
cd $R
python3 bench.py > $O/bench.json 2> $O/bench.err
echo "bench done"
cat $O/bench.json
