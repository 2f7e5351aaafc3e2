#!/bin/bash
# HBM traffic of every kernel of one bench step: two separate rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit
# one pass on gfx950), kernel-trace only.   pmc_bench.sh [cfg] [extra bench.py flags...]   -> CSV on stdout
# Counter -> bytes per ACCESS PATTERN, calibrated on this hardware with known access counts (bench_micro/micro_calib.hip,
# profiles/r03_counter_calibration.csv):
#   wide coalesced reads      FETCH_SIZE reports  8.0 B per 16-byte load  -> x 2   (MI355X_MICROARCH.md, HBM)
#   random 8-byte gathers     FETCH_SIZE reports 64.0 B per gather        -> x 1   (one 64-byte request each, tallied as is)
#   coalesced 16-byte stores  WRITE_SIZE reports 16.0 B per store         -> x 1
#   random 8/16-byte stores   WRITE_SIZE reports 32.0 B per store         -> x 1   (a 32-byte sector each)
#   device atomics            WRITE_SIZE reports 32.0 B per atomic, FETCH_SIZE nothing
# so traffic = f * FETCH_SIZE + WRITE_SIZE with f = 1 for the kernels whose reads are random gathers (list below), 2 otherwise.
cfg=${1:-3}; shift
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc${cfg}_$c -o p -- python $GRAFT_REPO_ROOT/bench.py --cfg $cfg --steps 1 --warmup 0 --no-cpu-baseline "$@" > $GRAFT_REPO_ROOT/gpurun_out/pmc${cfg}_$c.log 2>&1
done
CFG=$cfg python - <<PY
import csv, collections, glob, os, subprocess
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out"; cfg = os.environ["CFG"]
GATHER = ("k_walk_measure", "k_walk_copy", "k_rank8_jump", "k_rank_jump", "k_emit", "k_unitig_heads", "k_dr_jump", "k_dr_reply", "k_dr_apply", "k_pair_apply", "k_link_", "k_heads_measure")
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(collections.Counter)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{root}/pmc{cfg}_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[k][r["Counter_Name"]] += 1
try:
    sha = subprocess.run(["git", "-C", os.environ["GRAFT_REPO_ROOT"], "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or "snapshot"
except Exception:
    sha = "snapshot"
try:
    srchash = open(os.environ["GRAFT_REPO_ROOT"] + "/bcalm_amd/_build/libcdbg.so.srchash").read().strip()
except Exception:
    srchash = "unknown"
# (srchash: the content hash of the kernel sources libcdbg.so was built from -- bench.py quotes this file as roofline.traffic only while it matches)
print(f"# bench.py --cfg {cfg} --steps 1, two --pmc passes; source tree: {sha} (gpurun snapshot of the working tree); srchash={srchash}; factors: bench_micro/pmc_bench.sh header")
print("kernel,launches,FETCH_SIZE_KB_per_launch,WRITE_SIZE_KB_per_launch,fetch_factor,traffic_GB_per_launch")
for k in sorted(agg, key=lambda x: -(agg[x].get("FETCH_SIZE", 0) + agg[x].get("WRITE_SIZE", 0))):
    n = max(calls[k].values()); f = agg[k].get("FETCH_SIZE", 0) / n; w = agg[k].get("WRITE_SIZE", 0) / n
    fac = 1 if any(g in k for g in GATHER) else 2
    print(f"{k},{n},{f:.0f},{w:.0f},{fac},{(fac*f+w)*1024/1e9:.3f}")
PY
