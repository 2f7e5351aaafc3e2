#!/bin/bash
# dev tool: the scan's record store -- 8-byte pairs / one 16-byte store, issued at once / behind the next record's placement atomic (round 5, found last)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; tag=${1:-r05s17}; cd $R
L=$O/${tag}_store16.log; : > $L
run() { echo "# k=$K gen=$GEN $*" >> $L; env "$@" python bench_micro/hostile_timing.py $N $K 3 $GEN 2>&1 | grep -E "^\{|Error" | tail -2 | cut -c1-420 >> $L; }
N=100000000 K=31 GEN=0x3 run X=1
N=100000000 K=31 GEN=0x3 run CDBG_LIB=$R/bench_micro/variants/libcdbg_PIPE.so
N=100000000 K=31 GEN=0x3 run CDBG_LIB=$R/bench_micro/variants/libcdbg_PIPE16.so
N=125000000 K=55 GEN=0x4 run X=1
N=125000000 K=55 GEN=0x4 run CDBG_LIB=$R/bench_micro/variants/libcdbg_PIPE.so
cat $L
