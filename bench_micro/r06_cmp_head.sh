#!/bin/bash
# dev tool (round 6): which half of the strand-normalisation costs what -- HEAD library / HEAD scan + new count code / the working tree; config-4 share
R=$GRAFT_REPO_ROOT; cd $R
for lib in $R/bench_micro/variants/libcdbg_HEAD.so $R/bench_micro/variants/libcdbg_V1.so ""; do echo "== lib=$lib"; CDBG_LIB=$lib python bench_micro/quick_timing.py 125000000 55 2 2>&1 | grep "^{" | tail -1 | cut -c150-420; done
