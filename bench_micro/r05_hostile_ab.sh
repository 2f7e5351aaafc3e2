#!/bin/bash
# dev tool: A/B of the hostile config-3 line (round 5): packed region cursors, sizing from the first sample, partition count
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; tag=${1:-r05c}; cd $R
L=$O/${tag}_hostile_ab.log; : > $L
echo "# default (packed region cursors, regions sized from the first 1/64 sample)" >> $L
python bench_micro/hostile_timing.py 100000000 31 3 2>/dev/null >> $L
echo "# CDBG_VAR_RESAMPLE=4 (round 4's second sample of a quarter of the tiles; packed cursors)" >> $L
CDBG_VAR_RESAMPLE=4 python bench_micro/hostile_timing.py 100000000 31 3 2>/dev/null >> $L
echo "# 2^23 partitions" >> $L
CDBG_LOG_NP=23 python bench_micro/hostile_timing.py 100000000 31 3 2>/dev/null >> $L
echo "# uniform config 3, default" >> $L
python bench_micro/hostile_timing.py 100000000 31 3 3 2>/dev/null >> $L
cat $L
HSA_ENABLE_IPC_MODE_LEGACY=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 1 --force-dist --steps 3 --warmup 1 --no-cpu-baseline --no-e2e > $O/${tag}_bench_forcedist_1rank.json 2> $O/${tag}_forcedist.err
python - <<PY
import json
d=json.loads(open("$O/${tag}_bench_forcedist_1rank.json").read().strip().splitlines()[-1])
print("forcedist", d["ms_per_step"], d["stage_ms"], d["checks_passed"], d["digest"]["set_digest"])
PY
