#!/bin/bash
# dev tool: is the scan's placement atomic cheaper as a 32-bit operation?  (uniform config 3; the FILL64 variant's results are void by design)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; tag=${1:-r05e}; cd $R
L=$O/${tag}_atomic_width_ab.log; : > $L
run() { echo "# $1" >> $L; shift; env "$@" python bench_micro/hostile_timing.py 100000000 31 3 ${GEN:-0x3} 2>/dev/null | tail -2 >> $L; }
run "shipped library, capped scan (32-bit fill counters)" X=1
run "FILL64 variant: capped scan with a 64-bit atomic on an 8-byte counter per partition (count stage void)" CDBG_LIB=$R/bench_micro/variants/libcdbg_FILL64.so
run "shipped library, exact layout (64-bit cursors)" CDBG_SCAN_MODE=exact
run "EXACT32 variant: exact layout, 32-bit atomic on the low half of the same 8-byte cursors" CDBG_SCAN_MODE=exact CDBG_LIB=$R/bench_micro/variants/libcdbg_EXACT32.so
GEN=0x103 run "hostile, shipped library (count: multi-pass by k-mer hash again)" X=1
cat $L
