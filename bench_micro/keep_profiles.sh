#!/bin/bash
# dev tool (build container): copy the artifacts of final_artifacts.sh <tag> from gpurun_out/ into profiles/ under the round's names
#   keep_profiles.sh r03c r03
tag=$1; rnd=$2; O=gpurun_out; P=profiles
for f in bench_n1.json bench_cfg4_n1.json bench_cfg5_n1.json bench_cfg3_skewed.json bench_forcedist_1rank.json \
         kernel_stats.csv kernel_stats_cfg4.csv kernel_stats_cfg5.csv kernel_stats_cfg3_skewed.csv kernel_stats_forcedist_sharded.csv kernel_stats_forcedist_replicated.csv \
         pmc_hbm_traffic_per_kernel_cfg3.csv pmc_hbm_traffic_per_kernel_cfg4.csv pmc_hbm_traffic_per_kernel_cfg5.csv counter_calibration.csv \
         forcedist_sharded_timing.log forcedist_replicated_timing.log; do
  [ -f $O/${tag}_$f ] && cp $O/${tag}_$f $P/${rnd}_$f
done
sha=$(git rev-parse --short HEAD)
sed -i "s/source tree: snapshot (gpurun snapshot of the working tree)/source tree: commit $sha (+ working tree at the time of the run)/" $P/${rnd}_pmc_hbm_traffic_per_kernel_cfg*.csv
for f in $P/${rnd}_forcedist_*_timing.log; do grep "^{" $f > $f.tmp && mv $f.tmp $f; done
ls -la $P | grep ${rnd}_
