#!/bin/bash
# dev tool: config-5 share (6.25 M x 1 kbp, k = 127): partition count and the admission thresholds of the count tiers, after the round-5 sifting tier
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; tag=${1:-r05i}; cd $R
L=$O/${tag}_cfg5_sweep.log; : > $L
run() { echo "# $*" >> $L; env "$@" python bench_micro/hostile_timing.py 6250000 127 3 0x5 2>/dev/null | tail -1 >> $L; }
run X=1
run CDBG_LOG_NP=21
run CDBG_LOG_NP=23
run CDBG_FAST_SKIP_Q8=160
run CDBG_FAST_SKIP_Q8=230
run CDBG_FAST_SKIP2_Q8=160
run CDBG_FAST_SKIP2_Q8=224
run CDBG_COUNT_MAX_SUB=2
run CDBG_COUNT_MAX_SUB=4
cat $L
