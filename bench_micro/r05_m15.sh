#!/bin/bash
# dev tool: minimizer length 15 behind the 16-key window specialisation (k_scan_fast<1,.,16>) against m = 16 at config 3 (round 5, last)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; tag=${1:-r05m15}; cd $R
L=$O/${tag}_m15.log; : > $L
run() { echo "# k=$K gen=$GEN $*" >> $L; env "$@" python bench_micro/hostile_timing.py $N $K 4 $GEN 2>&1 | grep -E "^\{|Error" | tail -3 | cut -c1-460 >> $L; }
N=100000000 K=31 GEN=0x3 run CDBG_M=16
N=100000000 K=31 GEN=0x3 run CDBG_M=15
N=100000000 K=31 GEN=0x3 run CDBG_M=16
N=100000000 K=31 GEN=0x3 run CDBG_M=15
N=100000000 K=31 GEN=0x103 run CDBG_M=16
N=100000000 K=31 GEN=0x103 run CDBG_M=15
cat $L
