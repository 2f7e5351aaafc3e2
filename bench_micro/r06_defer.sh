#!/bin/bash
# dev tool (round 6): deferred record placement -- slices of the partition space (sixteenths per slice), placement grid; config 3
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; tag=${1:-r06e}; cd $R
L=$O/${tag}_defer.log; : > $L
run() { echo "# n=$N k=$K $*" >> $L; env "$@" python bench_micro/quick_timing.py $N $K ${REPS:-3} 2>&1 | grep -E "^\{|Error|error" | tail -${TAIL:-2} | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()[:300]); continue
    print('  wall %.1f  scan %.1f  count %.1f  place %.1f  compact %.1f  glue %.1f  slices %d' % (d['run_wall_ms'], d['ms_scan_emit'], d['ms_count'], d['ms_place'], d['ms_compact'], d['ms_glue'], d['count_slices']))
" >> $L; }
N=100000000 K=31
for pat in ${PATS:-0 8,8 4 6,10 10,6 12,4}; do run CDBG_DEFER_SLICES=$pat; done
cat $L
