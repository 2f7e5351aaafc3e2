#!/bin/bash
# dev tool: expanded passes of the multi-pass count kernel (round 5): hostile generator at k = 127 / 55 / 31, A/B against CDBG_NO_EXPAND and with more LDS passes allowed
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; tag=${1:-r05e2}; cd $R
L=$O/${tag}_expand.log; : > $L
run() { echo "# k=$K $*" >> $L; env "$@" python bench_micro/hostile_timing.py $N $K 3 $GEN 2>&1 | grep -E "^\{|Error|expanded passes" | tail -3 | cut -c1-420 >> $L; }
N=6250000 K=127 GEN=0x105 run CDBG_HOST_MARKS=1
N=6250000 K=127 GEN=0x105 run CDBG_NO_EXPAND=1
N=6250000 K=127 GEN=0x105 run CDBG_MAX_PASSES=64
N=125000000 K=55 GEN=0x104 run CDBG_HOST_MARKS=1
N=125000000 K=55 GEN=0x104 run CDBG_NO_EXPAND=1
N=125000000 K=55 GEN=0x104 run CDBG_MAX_PASSES=64
N=100000000 K=31 GEN=0x103 run CDBG_HOST_MARKS=1
N=100000000 K=31 GEN=0x103 run CDBG_NO_EXPAND=1
N=6250000 K=127 GEN=0x5 run CDBG_HOST_MARKS=1
N=6250000 K=127 GEN=0x5 run CDBG_NO_EXPAND=1
cat $L
timeout 900 python -m pytest tests -m gpu -x -q -k "config2 or multipass or hostile or ceiling or saturat or tier or count or big or hbm or overflow" > $O/${tag}_gputest_subset.log 2>&1; grep -E "passed|failed" $O/${tag}_gputest_subset.log
