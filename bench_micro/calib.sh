#!/bin/bash
# dev tool: counter bytes per access for each pattern of micro_calib.hip (two separate rocprofv3 --pmc passes, kernel-trace only)
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
$GRAFT_REPO_ROOT/bench_micro/micro_calib > $O/calib_timing.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/calib_$c -o p -- $GRAFT_REPO_ROOT/bench_micro/micro_calib > $O/calib_$c.log 2>&1
done
python3 - <<PY
import csv, glob, os, collections
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out"
rows = collections.OrderedDict()
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{root}/calib_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if not k.startswith("calib_"): continue
            rows.setdefault((k, r["Dispatch_Id"]), {})[c] = float(r["Counter_Value"])
N = 1 << 30
print("kernel,dispatch,FETCH_SIZE_KB,WRITE_SIZE_KB,fetch_bytes_per_access,write_bytes_per_access")
for (k, d), v in rows.items():
    f, w = v.get("FETCH_SIZE", 0.0), v.get("WRITE_SIZE", 0.0)
    print(f"{k},{d},{f:.0f},{w:.0f},{f*1024/N:.2f},{w*1024/N:.2f}")
PY
cat $O/calib_timing.log
