#!/bin/bash
# dev tool: per-kernel times (rocprofv3 --kernel-trace --stats) of the shipped library and of every experimental build
# bench_micro/variants/*.so on the config-3 workload: variants.sh <kernel grep pattern>
pat=${1:-k_}
cd /tmp && export TMPDIR=/tmp
for lib in $GRAFT_REPO_ROOT/bcalm_amd/_build/libcdbg.so $GRAFT_REPO_ROOT/bench_micro/variants/*.so; do
  tag=$(basename $lib .so)
  CDBG_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/var_$tag -o p -- python $GRAFT_REPO_ROOT/bench_micro/quick_timing.py ${N_READS:-100000000} ${K:-31} 2 > $GRAFT_REPO_ROOT/gpurun_out/var_$tag.log 2>&1
  echo "== $tag"; grep '^{' $GRAFT_REPO_ROOT/gpurun_out/var_$tag.log | tail -1 | cut -c1-400
  grep -E "$pat" $GRAFT_REPO_ROOT/gpurun_out/var_$tag/*kernel_stats.csv | awk -F, '{printf "%-70s calls %s avg_ms %.3f\n", substr($1,1,70), $2, $4/1e6}'
done
