#!/bin/bash
# dev tool (round 6): the scan with the next tile's loads in flight across a tile (k_scan_fast.h) -- slice patterns at config 3, config-4 / -5 shares
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; tag=${1:-r06p}; cd $R
L=$O/${tag}_scan_pipelined.log; : > $L
run() { echo "# n=$N k=$K $*" >> $L; env "$@" python bench_micro/quick_timing.py $N $K ${REPS:-3} 2>&1 | grep -E "^\{|Error|error" | tail -2 | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()[:300]); continue
    print('  wall %.1f  scan %.1f  count %.1f  place %.1f  compact %.1f  glue %.1f  slices %d' % (d['run_wall_ms'], d['ms_scan_emit'], d['ms_count'], d['ms_place'], d['ms_compact'], d['ms_glue'], d['count_slices']))
" >> $L; }
N=100000000 K=31
for pat in ${PATS:-4 4,5,4,3 4,5,5,2 5,4,4,3 5,5,4,2 6,5,3,2 4,4,4,3,1 6,6,4 3,5,5,3 8}; do run CDBG_DEFER_SLICES=$pat; done
N=125000000 K=55; run X=1
cat $L
