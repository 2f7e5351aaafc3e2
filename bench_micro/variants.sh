#!/bin/bash
# dev tool: per-kernel times (rocprofv3 --kernel-trace --stats) of the shipped library and of every experimental build
# bench_micro/variants/*.so on the config-3 workload: variants.sh <kernel grep pattern>
pat=${1:-k_}
cd /tmp && export TMPDIR=/tmp
for lib in $GRAFT_REPO_ROOT/bcalm_amd/_build/libcdbg.so $GRAFT_REPO_ROOT/bench_micro/variants/*.so; do
  tag=$(basename $lib .so)
  CDBG_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/var_$tag -o p -- python $GRAFT_REPO_ROOT/bench_micro/quick_timing.py ${N_READS:-100000000} ${K:-31} 2 > $GRAFT_REPO_ROOT/gpurun_out/var_$tag.log 2>&1
  echo "== $tag"; grep '^{' $GRAFT_REPO_ROOT/gpurun_out/var_$tag.log | tail -1 | cut -c1-400
  python3 - "$pat" $GRAFT_REPO_ROOT/gpurun_out/var_$tag/p_kernel_stats.csv <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[2])):
    if re.search(sys.argv[1], r["Name"]) and float(r["TotalDurationNs"]) > 1e5:
        print("%-72s calls %4s avg_ms %8.3f" % (r["Name"].split("(")[0][:72], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
done
