#!/bin/bash
# dev tool: config-5 share (k = 127) and k = 96: a third one-wave compaction tier (1024 slots) for three- and four-word k-mers
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; tag=${1:-r05y}; cd $R
L=$O/${tag}_t3w4.log; : > $L
run() { echo "# k=$K $*" >> $L; env "$@" python bench_micro/hostile_timing.py $N $K 3 $GEN 2>&1 | grep -E "^\{|Error" | tail -2 | cut -c1-420 >> $L; }
N=6250000 K=127 GEN=0x5 run X=1
N=6250000 K=127 GEN=0x5 run CDBG_LIB=$R/bench_micro/variants/libcdbg_T3W4.so
N=6250000 K=127 GEN=0x5 run CDBG_LIB=$R/bench_micro/variants/libcdbg_T3W4.so CDBG_CW_TIER3=off
cat $L
