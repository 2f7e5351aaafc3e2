#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
for i in 1 2 3 4 5 6; do timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "count_sift_tier" 2>&1 | tail -2 | head -1; done
