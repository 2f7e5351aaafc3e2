#!/bin/bash
# dev tool: record placement at two memory requests per record in every layout (round 5): exact layout with 32-bit cursors,
# skewed inputs through capped regions + overflow regions
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; tag=${1:-r05g}; cd $R
L=$O/${tag}_scan32.log; : > $L
run() { echo "# $1" >> $L; shift; env "$@" python bench_micro/hostile_timing.py 100000000 31 3 $GEN 2>/dev/null | tail -2 >> $L; }
GEN=0x103 run "hostile, default (capped regions + overflow regions)" X=1
GEN=0x103 run "hostile, exact layout (32-bit cursors)" CDBG_SCAN_MODE=exact
GEN=0x3 run "uniform, default (capped)" X=1
GEN=0x3 run "uniform, capped regions + overflow regions forced" CDBG_SCAN_MODE=var
GEN=0x3 run "uniform, exact layout (32-bit cursors)" CDBG_SCAN_MODE=exact
GEN=0x3 run "uniform, exact layout, 32-bit index + offset load (CDBG_EXACT_NO_CUR32)" CDBG_SCAN_MODE=exact CDBG_EXACT_NO_CUR32=1
cat $L
timeout 1500 python -m pytest tests -m gpu -x -q -k "capped or scan or hostile or skew or multi_rank or estimated or spill or overflow" > $O/${tag}_gputest_subset.log 2>&1; grep -E "passed|failed" $O/${tag}_gputest_subset.log
