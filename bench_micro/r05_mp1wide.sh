#!/bin/bash
# dev tool: hostile line, the multi-pass count kernel of one-word k-mers at 8192 slots / 1024 threads (shipped since round 5): workgroups of its launch
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; tag=${1:-r05p}; cd $R
L=$O/${tag}_mp1wide.log; : > $L
run() { echo "# $*" >> $L; env "$@" python bench_micro/hostile_timing.py 100000000 31 3 $GEN 2>/dev/null | tail -2 >> $L; }
GEN=0x103 run X=1
GEN=0x103 run CDBG_LIB=$R/bench_micro/variants/libcdbg_MPG2048.so
GEN=0x103 run CDBG_LIB=$R/bench_micro/variants/libcdbg_MPG4096.so
cat $L
timeout 900 python -m pytest tests -m gpu -x -q -k "config2 or multipass or hostile or ceiling or saturat or tier or count" > $O/${tag}_gputest_subset.log 2>&1; grep -E "passed|failed" $O/${tag}_gputest_subset.log
