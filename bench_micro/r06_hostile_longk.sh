#!/bin/bash
# dev tool (round 6): the hostile generator at k = 127 / 55 / 31 -- HBM-table pass one workgroup per partition (round 5) against the grid-wide pass, LDS pass limits
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; tag=${1:-r06k}; cd $R
L=$O/${tag}_hostile_long_k.log; : > $L
run() { echo "# k=$K $*" >> $L; env "$@" timeout 600 python bench_micro/hostile_timing.py $N $K 2 2>&1 | grep -E "^\{|Error|error" | tail -1 | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()[:300]); continue
    print('  scan %6.2f | count %8.2f | compact %7.2f | glue %6.2f | total %8.1f ms   digest %s  multipass %d  HBM-table partitions %d' % (d['ms_scan_emit'], d['ms_count'], d['ms_compact'], d['ms_glue'], d['ms_total'], d['set_digest'], d['n_multipass_partitions'], d['n_big_partitions']))
" >> $L; }
N=6250000 K=127
run CDBG_BIG_ONE_WG=1
run X=1
for mp in 1 2 4 8; do run CDBG_MAX_PASSES=$mp; done
N=125000000 K=55
run CDBG_BIG_ONE_WG=1
run X=1
for mp in 1 2 4 8; do run CDBG_MAX_PASSES=$mp; done
N=100000000 K=31
run CDBG_BIG_ONE_WG=1
run X=1
for mp in 2 4 8; do run CDBG_MAX_PASSES=$mp; done
cat $L
