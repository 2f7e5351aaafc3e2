#!/bin/bash
# dev tool: the hostile generator at longer k (one GPU's share of configs 4 / 5) -- a cliff the k = 31 line does not show
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; tag=${1:-r05u}; cd $R
L=$O/${tag}_hostile_long_k.log; : > $L
run() { echo "# $*" >> $L; env "$@" python bench_micro/hostile_timing.py $N $K 3 $GEN 2>&1 | grep -E "^\{|Error" | tail -2 | cut -c1-420 >> $L; }
N=125000000 K=55 GEN=0x104 run X=1
N=6250000 K=127 GEN=0x105 run X=1
N=100000000 K=31 GEN=0x103 run X=1
cat $L
