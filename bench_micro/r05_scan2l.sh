#!/bin/bash
# dev tool: config-4 share (k = 55, window of 39 keys): the compile-time doubling window against the two-level window minimum
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; tag=${1:-r05n}; cd $R
L=$O/${tag}_scan2l.log; : > $L
run() { echo "# $*" >> $L; env "$@" python bench_micro/hostile_timing.py 125000000 55 3 0x4 2>/dev/null | tail -2 >> $L; }
run X=1
run CDBG_SCAN_TWO_LEVEL=1
cat $L
