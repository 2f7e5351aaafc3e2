#!/bin/bash
# dev tool (round 6): records in flight per lane of k_place<1> (2 / 4 / 8) x its grid, bench.py --steps 15 per setting (one process each)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; tag=${1:-r06v}; cd $R
L=$O/${tag}_place_u.log; : > $L
run() { echo "# $*" >> $L; env "$@" python bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-e2e 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ms_per_step %.2f  ' % d['ms_per_step'], {k: round(v, 1) for k, v in d['stage_ms'].items() if k in ('ms_scan_emit', 'ms_count', 'ms_place')}, d['checks_passed'])" >> $L; }
run X=1
run CDBG_LIB=$R/bench_micro/variants/libcdbg_PU2.so
run CDBG_LIB=$R/bench_micro/variants/libcdbg_PU2.so CDBG_PLACE_GRID=1024
run CDBG_LIB=$R/bench_micro/variants/libcdbg_PU8.so
run CDBG_LIB=$R/bench_micro/variants/libcdbg_PU8.so CDBG_PLACE_GRID=256
run X=2
cat $L
