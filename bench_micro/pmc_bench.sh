#!/bin/bash
# HBM traffic of every kernel of one bench step: two separate rocprofv3 --pmc passes
# (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950), kernel-trace only.
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$c -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/pmc_$c.log 2>&1
done
python - <<PY
import csv, collections, glob, os
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out"
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(collections.Counter)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{root}/pmc_{c}/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[k][r["Counter_Name"]] += 1
print("kernel,launches,FETCH_SIZE_KB_per_launch,WRITE_SIZE_KB_per_launch,traffic_GB_per_launch(2*FETCH+WRITE)")
for k in sorted(agg, key=lambda x: -agg[x].get("FETCH_SIZE", 0)):
    n = max(calls[k].values()); f = agg[k].get("FETCH_SIZE", 0) / n; w = agg[k].get("WRITE_SIZE", 0) / n
    print(f"{k},{n},{f:.0f},{w:.0f},{(2*f+w)*1024/1e9:.3f}")
PY
