#!/bin/bash
# Regenerates the evidence under profiles/ in one go (run on the GPU box from the repo root: bash tools/final_profiles.sh).
# Outputs land in gpurun_out/final/; copy what is to be judged into profiles/.
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
mkdir -p $O
export PYTHONPATH=$R
bash $R/tools/pmc_sq.sh > $O/pmc_sq.txt 2>&1
echo "pmc_sq done"
bash $R/tools/pmc_traffic.sh 5 > $O/pmc_traffic.txt 2>&1
echo "pmc_traffic done"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o c2 -- python3 $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-standalone > $O/stats_bench.json 2> $O/stats.err
echo "kernel stats done"
cd $R
python3 bench.py > $O/bench.json 2> $O/bench.err
echo "bench done"
cat $O/bench.json
