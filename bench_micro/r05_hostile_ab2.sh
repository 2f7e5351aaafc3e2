#!/bin/bash
# dev tool: what makes the hostile scan slower than the uniform one (65 vs 85 ms for the same 1.6 G records), and the multi-pass count
# with record-level passes (sub-partition bits in the record meta)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; tag=${1:-r05d}; cd $R
L=$O/${tag}_hostile_ab2.log; : > $L
run() { echo "# $1" >> $L; shift; env "$@" python bench_micro/hostile_timing.py 100000000 31 2 $GEN 2>/dev/null | tail -1 >> $L; }
GEN=0x103 run "hostile, default" X=1
GEN=0x103 run "hostile, CDBG_SCAN_MODE=capped (uniform regions + spill list)" CDBG_SCAN_MODE=capped
GEN=0x103 run "hostile, CDBG_SCAN_MODE=exact (two passes)" CDBG_SCAN_MODE=exact
GEN=0x3 run "uniform, default (capped)" X=1
GEN=0x3 run "uniform, CDBG_SCAN_MODE=var" CDBG_SCAN_MODE=var
GEN=0x3 run "uniform, CDBG_SCAN_MODE=exact" CDBG_SCAN_MODE=exact
cat $L
