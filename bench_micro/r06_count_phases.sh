#!/bin/bash
# dev tool (round 6): where k_count_fast<1> loses its 9 ms beside the placement kernel -- phase cycles (first / last wave of every workgroup, summed) with and without deferral
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; tag=${1:-r06w}; cd $R
L=$O/${tag}_count_phases.log; : > $L
for s in 0 8,8; do echo "# CDBG_DEFER_SLICES=$s" >> $L; CDBG_DEFER_SLICES=$s CDBG_LIB=$R/bench_micro/variants/libcdbg_PHASES.so python bench_micro/quick_timing.py 100000000 31 3 2>&1 | grep -E "k_count_fast phase|^\{" | tail -3 | cut -c1-360 >> $L; done
cat $L
