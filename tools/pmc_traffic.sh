#!/bin/bash
# HBM traffic per kernel launch from PMC counters (MI355X_MICROARCH.md §HBM): separate --pmc passes, csv output.
# usage (on the GPU box, from the repo root): bash tools/pmc_traffic.sh [steps]
set -e
STEPS=${1:-5}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/$C -- python3 $GRAFT_REPO_ROOT/bench.py --steps $STEPS --warmup 1 --no-cpu-baseline > $OUT/$C.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for c in ("FETCH_SIZE","WRITE_SIZE"):
    files = glob.glob("$OUT/%s/**/*counter_collection.csv" % c, recursive=True)
    agg = collections.defaultdict(list)
    for f in files:
        for row in csv.DictReader(open(f)):
            agg[row["Kernel_Name"].split("(")[0][:60]].append(float(row["Counter_Value"]))
    for k, v in sorted(agg.items()):
        if any(s in k for s in ("greedy","leap","nw_","pack","accuracy")):
            print("%-12s %-62s launches=%4d mean=%.1f (counter units: KB)" % (c, k, len(v), sum(v)/len(v)))
PY
