#!/bin/bash
# VERDICT r4 next #3b: what differs between a process whose read scan runs in the fast mode (66 - 68 ms at config 3) and one in the slow
# mode (73 - 78 ms)?  Every process lands in one mode for its lifetime; here REPS processes per counter group run the config-3 job under
# rocprofv3 --kernel-trace --pmc <group> (counters in their own runs); per process: duration of k_scan_fast<1, 2, 15> and its counters.
REPS=${1:-4}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/mode_pmc; rm -rf $OUT; mkdir -p $OUT
GROUPS_=(
 "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum"
 "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_TAG_STALL_sum"
 "TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"
 "TCC_BUSY_sum TCC_CYCLE_sum TCC_EA0_WRREQ_LEVEL_sum GRBM_UTCL2_BUSY"
 "TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_THRASHING_STALL_sum"
)
gi=0
for g in "${GROUPS_[@]}"; do
  for r in $(seq 1 $REPS); do
    rocprofv3 --kernel-trace --pmc $g --output-format csv -d $OUT/g${gi}_r$r -o p -- python $GRAFT_REPO_ROOT/bench_micro/quick_timing.py 100000000 31 2 > $OUT/g${gi}_r$r.log 2>&1
  done
  gi=$((gi+1))
done
# plain processes in between: which mode does each land in without the profiler?
for r in 1 2 3; do python $GRAFT_REPO_ROOT/bench_micro/quick_timing.py 100000000 31 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('plain process %s: scan %.2f ms' % ('$r', d['ms_scan_emit']))"; done
python - <<'PY'
import csv, glob, os, collections
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/mode_pmc"
rows = []
for d in sorted(glob.glob(root + "/g*_r*")):
    if not os.path.isdir(d): continue
    per = collections.defaultdict(lambda: collections.defaultdict(float)); dur = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            if not r["Kernel_Name"].startswith("void cdbg::k_scan_fast<1, 2, 15>"): continue
            per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Dispatch_Id"] not in seen and "Start_Timestamp" in r:
                seen.add(r["Dispatch_Id"]); dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    if not per:
        print(os.path.basename(d), "no scan dispatch found"); continue
    disp = sorted(per, key=lambda x: int(x))[-1]                     # the last full scan of the process
    print("%-8s scan %7.2f ms  " % (os.path.basename(d), dur.get(disp, float("nan"))) + "  ".join("%s=%.4g" % (k, v) for k, v in sorted(per[disp].items())))
PY
rm -rf $OUT/*/  # the raw traces are large; the table above is what is kept
