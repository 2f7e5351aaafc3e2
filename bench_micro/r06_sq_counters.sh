#!/bin/bash
# Round 6: shader-side counters of the config-3 kernels (what "VALU- / LDS-bound" rests on): rocprofv3 --kernel-trace --pmc <group>, one group per process,
# the LAST launch of each stage kernel of a 2-step run.
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/sq_pmc; rm -rf $OUT; mkdir -p $OUT
GROUPS_=( "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAIT_INST_ANY" )
gi=0
for g in "${GROUPS_[@]}"; do
  rocprofv3 --kernel-trace --pmc $g --output-format csv -d $OUT/g$gi -o p -- python $GRAFT_REPO_ROOT/bench_micro/hostile_timing.py 100000000 31 2 0x3 > $OUT/g$gi.log 2>&1
  gi=$((gi+1))
done
python - <<'PY'
import csv, glob, os, collections
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/sq_pmc"
want = ("k_scan_fast<1, 2, 15>", "k_place<1>", "k_count_fast<1, 4096", "k_compact_wave<1, 512>", "k_walk_copy", "k_join_bucket<1>", "k_walk_measure")
tab = collections.defaultdict(dict)
for d in sorted(glob.glob(root + "/g*")):
    if not os.path.isdir(d): continue
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            for w in want:
                if w in r["Kernel_Name"]: per[(w, int(r["Dispatch_Id"]))][r["Counter_Name"]] += float(r["Counter_Value"])
    for w in want:
        ds = sorted(x[1] for x in per if x[0] == w)
        if ds: tab[w].update(per[(w, ds[-1])])
for w in want:
    print("%-26s " % w + "  ".join("%s=%.4g" % (k, v) for k, v in sorted(tab[w].items())))
PY
rm -rf $OUT/*/
