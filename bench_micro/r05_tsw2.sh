#!/bin/bash
# dev tool: config-4 share (k = 55): a third one-wave compaction tier (1024 slots) before the workgroup tiers; CDBG_CW_TIER3 set: without it
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; tag=${1:-r05s}; cd $R
L=$O/${tag}_cwtier3.log; : > $L
run() { echo "# $*" >> $L; env "$@" python bench_micro/hostile_timing.py $N $K 3 $GEN 2>/dev/null | tail -2 >> $L; }
N=125000000 K=55 GEN=0x4 run X=1
N=125000000 K=55 GEN=0x4 run CDBG_CW_TIER3=off
N=125000000 K=55 GEN=0x104 run X=1
N=125000000 K=55 GEN=0x104 run CDBG_CW_TIER3=off
cat $L
