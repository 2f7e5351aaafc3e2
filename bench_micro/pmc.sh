#!/bin/bash
# usage: pmc.sh <tag> <n_reads> <k> <kernel substring> <counter list...>   (one rocprofv3 --pmc pass, kernel-trace only)
tag=$1; shift; n=$1; shift; k=$1; shift; kern=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag -o p -- python $GRAFT_REPO_ROOT/bench_micro/quick_timing.py $n $k 1 > $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.log 2>&1
python - <<PY
import csv, collections, glob
f = glob.glob("$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag/**/*counter_collection.csv", recursive=True)
if not f: print("no counter csv"); raise SystemExit
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].split("(")[0][:60]
    if "$kern" in k: agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in agg.items():
    print("$tag", k, {c: f"{v:.4g}" for c, v in d.items()})
PY
