#!/bin/bash
# dev tool: second one-wave compaction tier of one-word k-mers (buckets of 257 .. 512 entries; 10-bit end ids), hostile and uniform config 3
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; tag=${1:-r05q}; cd $R
L=$O/${tag}_cwtier2.log; : > $L
run() { echo "# $*" >> $L; env "$@" python bench_micro/hostile_timing.py 100000000 31 3 $GEN 2>/dev/null | tail -2 >> $L; }
GEN=0x103 run X=1
GEN=0x103 run CDBG_CW_TIER2=0
GEN=0x3 run X=1
GEN=0x3 run CDBG_CW_TIER2=1
cat $L
timeout 900 python -m pytest tests -m gpu -x -q -k "second_wave or hostile or compact or split or glue_record" > $O/${tag}_gputest_subset.log 2>&1; grep -E "passed|failed" $O/${tag}_gputest_subset.log
