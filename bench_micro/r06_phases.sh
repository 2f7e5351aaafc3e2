#!/bin/bash
# dev tool (round 6): where the scan's time goes per phase (wall-clock ticks of thread 0, summed over the workgroups), with and without deferred placement
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; tag=${1:-r06i}; cd $R
L=$O/${tag}_scan_phases.log; : > $L
for s in 0 2 4; do echo "# CDBG_DEFER_SLICES=$s" >> $L; CDBG_DEFER_SLICES=$s CDBG_PLACE_GRID=512 CDBG_LIB=$R/bench_micro/variants/libcdbg_PHASES.so python bench_micro/quick_timing.py 100000000 31 2 2>&1 | grep -E "k_scan phase|^\{" | tail -2 | cut -c1-330 >> $L; done
cat $L
