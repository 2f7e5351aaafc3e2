#!/bin/bash
# dev tool: round 6's fuzz session on the GPU (oracle-checked): deferred placement forced (slices, spills in k_place, stream segments that fill up, overflow-region layout),
# the grid-wide HBM-table pass forced (one LDS pass at most), the default paths, multi-rank, links / abundances
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; tag=${1:-r06f}; sb=${2:-90}; cd $R   # r06_fuzz.sh <tag> [seed base]
L=$O/${tag}_fuzz.log; : > $L
f() { echo "# fuzz_gpu $1 s, $2" >> $L; shift; local t=$1; shift; env "$@" timeout $((t * 2 + 100)) python bench_micro/fuzz_gpu.py $t $SEED 2>/dev/null | tail -2 >> $L; }
n=0; nx() { n=$((n + 1)); SEED=$((sb + n)); }
nx; f "default" 150 X=1
nx; f "capped layout, two halves deferred, 2^10 - 2^12 partitions" 150 CDBG_SCAN_MODE=capped CDBG_DEFER_SLICES=2 FUZZ_LOG_NP=10,11,12
nx; f "capped layout, slices 4,4,4,2,1,1, regions of 24 records (k_place spills, repair behind the last stream)" 150 CDBG_SCAN_MODE=capped CDBG_DEFER_SLICES=4,4,4,2,1,1 CDBG_PART_CAP=24 FUZZ_LOG_NP=10,11,12
nx; f "capped layout, 16 slices, stream segments of 5 records (most records placed by the scan after all)" 100 CDBG_SCAN_MODE=capped CDBG_DEFER_SLICES=16 CDBG_DEFER_CAP=5 FUZZ_LOG_NP=10,12
nx; f "overflow-region layout, four quarters deferred, regions of 16 records" 150 CDBG_SCAN_MODE=var CDBG_DEFER_SLICES=4 CDBG_PART_CAP=16 FUZZ_LOG_NP=10,11,12
nx; f "overflow-region layout too small (spill + repair), two halves deferred" 100 CDBG_SCAN_MODE=var CDBG_DEFER_SLICES=2 CDBG_PART_CAP=8 CDBG_VAR_SCALE=0.4 FUZZ_LOG_NP=10,12
nx; f "one LDS pass at most: every larger partition through the grid-wide HBM tables" 150 CDBG_MAX_PASSES=1 FUZZ_LOG_NP=0,1,3
nx; f "no second count tier, one LDS pass: tables for everything the one-pass tier refuses" 100 CDBG_MAX_PASSES=1 CDBG_NO_COUNT_TIER2=1 FUZZ_LOG_NP=0,2,5
nx; echo "# fuzz_dist_gpu 150 s" >> $L; timeout 400 python bench_micro/fuzz_dist_gpu.py 150 $SEED 2>/dev/null | tail -2 >> $L
nx; echo "# fuzz_aux_gpu 100 s (links, abundance vectors)" >> $L; timeout 300 python bench_micro/fuzz_aux_gpu.py 100 $SEED 2>/dev/null | tail -2 >> $L
cat $L | cut -c1-300
