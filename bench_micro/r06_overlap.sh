#!/bin/bash
# dev tool (VERDICT r5 "next" #1, step 0): a pure placement kernel beside k_count_fast<1> on a config-3 context -- alone / at once.
# Needs bench_micro/variants/libcdbg_OVERLAP.so (hipcc ... -DCDBG_AB_OVERLAP).  r06_overlap.sh [tag]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; tag=${1:-r06a}; cd $R
L=$O/${tag}_ab_overlap_place_vs_count.log; : > $L
run() { echo "# $*" >> $L; env "$@" CDBG_LIB=$R/bench_micro/variants/libcdbg_OVERLAP.so python bench_micro/quick_timing.py 100000000 31 ${REPS:-6} 2>&1 | grep -E "ab-overlap|Error" | cut -c1-400 >> $L; }
# where do the waves land (quota 0 = only logged)?  1024 one-wave workgroups = the 256 x 256-thread grid of the run before
run CDBG_AB_OVERLAP=800000000 CDBG_AB_PLACE_GRID=1024 CDBG_AB_PLACE_QUOTA=0
run CDBG_AB_OVERLAP=800000000 CDBG_AB_PLACE_GRID=2048 CDBG_AB_PLACE_QUOTA=0
# a quota of waves per CU, eight times as many workgroups as stay
for q in 2 4 8; do run CDBG_AB_OVERLAP=800000000 CDBG_AB_PLACE_GRID=$((256 * q * 8)) CDBG_AB_PLACE_QUOTA=$q; done
run CDBG_AB_OVERLAP=800000000 CDBG_AB_PLACE_GRID=8192 CDBG_AB_PLACE_QUOTA=4 CDBG_AB_PLACE_PRIO=1
run CDBG_AB_OVERLAP=800000000 CDBG_AB_PLACE_GRID=8192 CDBG_AB_PLACE_QUOTA=4 CDBG_AB_PLACE_UNROLL=1
run CDBG_AB_OVERLAP=1200000000 CDBG_AB_PLACE_GRID=8192 CDBG_AB_PLACE_QUOTA=4
cat $L
