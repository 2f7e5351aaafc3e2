R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; tag=${1:-r06z}; cd $R
timeout 2000 python -m pytest tests -m gpu -x -q > $O/${tag}_gputest.log 2>&1; grep -E "passed|failed" $O/${tag}_gputest.log
python bench.py --steps 20 --warmup 5 > $O/${tag}_bench_n1.json 2> $O/${tag}_bench_n1.err
python bench.py --cfg 4 --steps 3 --warmup 1 --no-cpu-baseline --no-e2e > $O/${tag}_bench_cfg4_n1.json 2> $O/${tag}_bench_cfg4.err
python bench.py --cfg 5 --steps 3 --warmup 1 --no-cpu-baseline --no-e2e > $O/${tag}_bench_cfg5_n1.json 2> $O/${tag}_bench_cfg5.err
python bench.py --cfg 3 --skewed --steps 3 --warmup 1 --no-cpu-baseline --no-e2e > $O/${tag}_bench_cfg3_skewed.json 2> $O/${tag}_bench_cfg3_skewed.err
HSA_ENABLE_IPC_MODE_LEGACY=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 1 --force-dist --steps 3 --warmup 1 --no-cpu-baseline --no-e2e > $O/${tag}_bench_forcedist_1rank.json 2> $O/${tag}_forcedist.err
for f in n1 cfg4_n1 cfg5_n1 cfg3_skewed forcedist_1rank; do python - <<PY
import json
try:
    d=json.loads(open("$O/${tag}_bench_$f.json").read().strip().splitlines()[-1])
    print("$f", round(d["ms_per_step"],1), {k:round(v,1) for k,v in d["stage_ms"].items()}, d["checks_passed"])
except Exception as e: print("$f", "ERR", e)
PY
done
