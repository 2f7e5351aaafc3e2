#!/bin/bash
# dev tool: per-kernel times of the hostile generator at k = 127 and k = 55 (one GPU's share of configs 5 / 4)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; tag=${1:-r05v}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_p127 -o p -- python $R/bench_micro/hostile_timing.py 6250000 127 2 0x105 > $O/${tag}_p127.log 2>&1
cp $O/${tag}_p127/p_kernel_stats.csv $O/${tag}_kernel_stats_cfg5_hostile.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_p55 -o p -- python $R/bench_micro/hostile_timing.py 125000000 55 2 0x104 > $O/${tag}_p55.log 2>&1
cp $O/${tag}_p55/p_kernel_stats.csv $O/${tag}_kernel_stats_cfg4_hostile.csv
head -12 $O/${tag}_kernel_stats_cfg5_hostile.csv | cut -c1-160; head -12 $O/${tag}_kernel_stats_cfg4_hostile.csv | cut -c1-160
