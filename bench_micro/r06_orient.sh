#!/bin/bash
# dev tool (round 6): strand-normalised records of multi-word k-mers -- config-4 / -5 shares with and without (CDBG_NO_ORIENT=1), launch bounds of the 39-key scan
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; tag=${1:-r06s}; cd $R
L=$O/${tag}_orient.log; : > $L
run() { echo "# n=$N k=$K $*" >> $L; env "$@" python bench_micro/hostile_timing.py $N $K 3 $GEN 2>&1 | grep -E "^\{|Error|error" | tail -2 | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()[:300]); continue
    print('  wall %.1f  scan %.1f  count %.1f  compact %.1f  glue %.1f   records %d  digest %s' % (d['run_wall_ms'], d['ms_scan_emit'], d['ms_count'], d['ms_compact'], d['ms_glue'], d['n_records'], d['set_digest']))
" >> $L; }
N=125000000 K=55 GEN=0x4
run X=1
run CDBG_NO_ORIENT=1
run CDBG_LIB=$R/bench_micro/variants/libcdbg_W39_4.so
run CDBG_LIB=$R/bench_micro/variants/libcdbg_HEAD.so
N=6250000 K=127 GEN=0x5
run X=1
run CDBG_NO_ORIENT=1
run CDBG_LIB=$R/bench_micro/variants/libcdbg_HEAD.so
cat $L
