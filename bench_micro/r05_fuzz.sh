#!/bin/bash
# dev tool: the round's fuzz session on the GPU (oracle-checked): default paths, the overflow-region layout forced, exact layout, multi-rank, links / abundances, odd shapes
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; tag=${1:-r05m}; sb=${2:-50}; cd $R   # r05_fuzz.sh <tag> [seed base]
L=$O/${tag}_fuzz.log; : > $L
echo "# fuzz_gpu 200 s, default" >> $L; timeout 400 python bench_micro/fuzz_gpu.py 200 $((sb+1)) 2>/dev/null | tail -3 >> $L
echo "# fuzz_gpu 150 s, CDBG_SCAN_MODE=var CDBG_PART_CAP=16 (every busy partition through an overflow region)" >> $L; CDBG_SCAN_MODE=var CDBG_PART_CAP=16 timeout 400 python bench_micro/fuzz_gpu.py 150 $((sb+2)) 2>/dev/null | tail -3 >> $L
echo "# fuzz_gpu 100 s, CDBG_SCAN_MODE=var CDBG_PART_CAP=8 CDBG_VAR_SCALE=0.4 (overflow regions too small: spill + repair)" >> $L; CDBG_SCAN_MODE=var CDBG_PART_CAP=8 CDBG_VAR_SCALE=0.4 timeout 400 python bench_micro/fuzz_gpu.py 100 $((sb+3)) 2>/dev/null | tail -3 >> $L
echo "# fuzz_gpu 80 s, CDBG_SCAN_MODE=capped CDBG_PART_CAP=32" >> $L; CDBG_SCAN_MODE=capped CDBG_PART_CAP=32 timeout 400 python bench_micro/fuzz_gpu.py 80 $((sb+4)) 2>/dev/null | tail -3 >> $L
echo "# fuzz_dist_gpu 150 s" >> $L; timeout 400 python bench_micro/fuzz_dist_gpu.py 150 $((sb+5)) 2>/dev/null | tail -3 >> $L
echo "# fuzz_dist_gpu 80 s, CDBG_SCAN_MODE=capped CDBG_PART_CAP=4 (spills in the multi-rank pack)" >> $L; CDBG_SCAN_MODE=capped CDBG_PART_CAP=4 timeout 400 python bench_micro/fuzz_dist_gpu.py 80 $((sb+6)) 2>/dev/null | tail -3 >> $L
echo "# fuzz_aux_gpu 100 s (links, abundance vectors)" >> $L; timeout 300 python bench_micro/fuzz_aux_gpu.py 100 $((sb+7)) 2>/dev/null | tail -3 >> $L
echo "# stress_shapes" >> $L; timeout 900 python bench_micro/stress_shapes.py 2>/dev/null | tail -14 >> $L
cat $L | cut -c1-260
