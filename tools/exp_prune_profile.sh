# development tool: kernel split of the pruned wide-band Greedy under rocprofv3 (GPU box, from the repo root)
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3
mkdir -p $O
PYTHONPATH=$GRAFT_REPO_ROOT timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_prune -o c3 -- python3 $GRAFT_REPO_ROOT/tools/bench_quick.py C3 2e6 > $O/prof_prune.txt 2>&1 || { tail -5 $O/prof_prune.txt; exit 1; }
find $O/prof_prune -name "*kernel_stats.csv" | while read f; do cut -c1-180 "$f" | sed -n 1,8p; done
