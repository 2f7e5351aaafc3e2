#!/bin/bash
# dev tool: everything profiles/ holds for a round, in one GPU call: final_artifacts.sh <tag>   (outputs under gpurun_out/<tag>_*)
tag=${1:-rXX}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
python bench.py --steps 20 --warmup 5 > $O/${tag}_bench_n1.json 2> $O/${tag}_bench_n1.err
python bench.py --cfg 4 --steps 3 --warmup 1 --no-cpu-baseline > $O/${tag}_bench_cfg4_n1.json 2> $O/${tag}_bench_cfg4.err
python bench.py --cfg 5 --steps 3 --warmup 1 --no-cpu-baseline > $O/${tag}_bench_cfg5_n1.json 2> $O/${tag}_bench_cfg5.err
python bench.py --cfg 3 --skewed --steps 3 --warmup 1 > $O/${tag}_bench_cfg3_skewed.json 2> $O/${tag}_bench_cfg3_skewed.err
HSA_ENABLE_IPC_MODE_LEGACY=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 1 --force-dist --steps 3 --warmup 1 --no-cpu-baseline > $O/${tag}_bench_forcedist_1rank.json 2> $O/${tag}_forcedist.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_prof -o p -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/${tag}_prof.log 2>&1
cp $O/${tag}_prof/p_kernel_stats.csv $O/${tag}_kernel_stats.csv
for c in 4 5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_prof_cfg$c -o p -- python $R/bench.py --cfg $c --steps 2 --warmup 1 --no-cpu-baseline > $O/${tag}_prof_cfg$c.log 2>&1
  cp $O/${tag}_prof_cfg$c/p_kernel_stats.csv $O/${tag}_kernel_stats_cfg$c.csv
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_prof_skewed -o p -- python $R/bench.py --cfg 3 --skewed --steps 2 --warmup 1 --no-cpu-baseline > $O/${tag}_prof_skewed.log 2>&1
cp $O/${tag}_prof_skewed/p_kernel_stats.csv $O/${tag}_kernel_stats_cfg3_skewed.csv
# the multi-GPU code path through a 1-rank RCCL communicator: sharded reads (X1) and replicated reads (X0)
for mode in sharded replicated; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_prof_fd_$mode -o p -- python $R/bench_micro/forcedist_timing.py 100000000 31 3 $mode > $O/${tag}_forcedist_${mode}_timing.log 2>&1
  cp $O/${tag}_prof_fd_$mode/p_kernel_stats.csv $O/${tag}_kernel_stats_forcedist_$mode.csv
done
for c in 3 4 5; do bash $R/bench_micro/pmc_bench.sh $c > $O/${tag}_pmc_hbm_traffic_per_kernel_cfg$c.csv 2> $O/${tag}_pmc_cfg$c.err; done
bash $R/bench_micro/calib.sh > $O/${tag}_counter_calibration.csv 2> $O/${tag}_calib.err
tail -c 300 $O/${tag}_bench_n1.json; head -6 $O/${tag}_pmc_hbm_traffic_per_kernel_cfg3.csv
