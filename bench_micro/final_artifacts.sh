#!/bin/bash
# dev tool: everything profiles/ holds for a round, in one GPU call: final_artifacts.sh <tag> <round>   (outputs under gpurun_out/<tag>_*)
# Order (round 5): the PMC traffic tables FIRST and into the box's profiles/ under the round's name, so that every bench line taken after them quotes
# roofline.traffic (bench.py takes the table whose source hash matches the loaded library).
tag=${1:-rXX}; rnd=${2:-r05}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
for c in 3 4 5; do
  bash $R/bench_micro/pmc_bench.sh $c --no-e2e > $O/${tag}_pmc_hbm_traffic_per_kernel_cfg$c.csv 2> $O/${tag}_pmc_cfg$c.err
  cp $O/${tag}_pmc_hbm_traffic_per_kernel_cfg$c.csv $R/profiles/${rnd}_pmc_hbm_traffic_per_kernel_cfg$c.csv
done
cd $R
python bench.py --steps 20 --warmup 5 > $O/${tag}_bench_n1.json 2> $O/${tag}_bench_n1.err
python bench.py --cfg 4 --steps 3 --warmup 1 > $O/${tag}_bench_cfg4_n1.json 2> $O/${tag}_bench_cfg4.err
python bench.py --cfg 5 --steps 3 --warmup 1 --no-cpu-baseline > $O/${tag}_bench_cfg5_n1.json 2> $O/${tag}_bench_cfg5.err
python bench.py --cfg 3 --skewed --steps 3 --warmup 1 > $O/${tag}_bench_cfg3_skewed.json 2> $O/${tag}_bench_cfg3_skewed.err
HSA_ENABLE_IPC_MODE_LEGACY=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 1 --force-dist --steps 3 --warmup 1 --no-cpu-baseline --no-e2e > $O/${tag}_bench_forcedist_1rank.json 2> $O/${tag}_forcedist.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_prof -o p -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-e2e > $O/${tag}_prof.log 2>&1
cp $O/${tag}_prof/p_kernel_stats.csv $O/${tag}_kernel_stats.csv
for c in 4 5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_prof_cfg$c -o p -- python $R/bench.py --cfg $c --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > $O/${tag}_prof_cfg$c.log 2>&1
  cp $O/${tag}_prof_cfg$c/p_kernel_stats.csv $O/${tag}_kernel_stats_cfg$c.csv
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_prof_skewed -o p -- python $R/bench.py --cfg 3 --skewed --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > $O/${tag}_prof_skewed.log 2>&1
cp $O/${tag}_prof_skewed/p_kernel_stats.csv $O/${tag}_kernel_stats_cfg3_skewed.csv
# the multi-GPU code path through a 1-rank RCCL communicator: sharded reads (X1) and replicated reads (X0)
for mode in sharded replicated; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_prof_fd_$mode -o p -- python $R/bench_micro/forcedist_timing.py 100000000 31 3 $mode > $O/${tag}_forcedist_${mode}_timing.log 2>&1
  cp $O/${tag}_prof_fd_$mode/p_kernel_stats.csv $O/${tag}_kernel_stats_forcedist_$mode.csv
done
bash $R/bench_micro/calib.sh > $O/${tag}_counter_calibration.csv 2> $O/${tag}_calib.err
cd $R
timeout 1800 python -m pytest tests -m gpu -q > $O/${tag}_gputest.log 2>&1; grep -E "passed|failed" $O/${tag}_gputest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/${tag}_smoke.log 2>&1; tail -2 $O/${tag}_smoke.log
for f in n1 cfg4_n1 cfg5_n1 cfg3_skewed forcedist_1rank; do python - <<PY
import json
try:
    d=json.loads(open("$O/${tag}_bench_$f.json").read().strip().splitlines()[-1]); r=d["roofline"]
    print("$f", round(d["ms_per_step"],2), "%.2f G/s" % (d["value"]/1e9), {k:round(v,1) for k,v in d["stage_ms"].items()}, d["checks_passed"], r["kernel"], round(r["frac"],4), "traffic", r["traffic"] and round(r["traffic"]/1e9,1))
except Exception as e: print("$f", "ERR", e)
PY
done
