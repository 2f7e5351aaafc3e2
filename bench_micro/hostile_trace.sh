#!/bin/bash
# dev tool: per-launch kernel trace of the hostile config-3 line (gpurun_out/<tag>_hostile_trace.csv) + bucket-size histogram
tag=${1:-rXX}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
CDBG_DEBUG_SEGHIST=1 python $R/bench_micro/hostile_timing.py 100000000 31 2 > $O/${tag}_hostile_timing.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/${tag}_hprof -o p -- python $R/bench_micro/hostile_timing.py 100000000 31 2 > $O/${tag}_hprof.log 2>&1
python - <<PY
import csv, glob
f = glob.glob("$O/${tag}_hprof/**/p_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
out = open("$O/${tag}_hostile_trace.csv", "w")
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    if d >= 0.05: out.write("%10.2f %8.2f ms  grid %-9s %s\n" % ((int(r["Start_Timestamp"]) - t0) / 1e6, d, r.get("Grid_Size_X", r.get("Grid_Size", "")), r["Kernel_Name"][:110]))
PY
tail -4 $O/${tag}_hostile_timing.log
