#!/bin/bash
# Round 5: the L2's request counters of the scan's emit pass under its three placements (config 3, 1.6 G records): capped layout (a returning 32-bit atomic +
# a 16-byte store per record), exact layout with pre-loaded 32-bit cursors (the same two), exact layout with a 32-bit index + a load of the region's offset (three).
# rocprofv3 --kernel-trace --pmc <group>, counters in their own runs; per process the last full emit launch of k_scan_fast<1, MODE, 15>.
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/scan_req_pmc; rm -rf $OUT; mkdir -p $OUT
GROUPS_=( "TCC_REQ_sum TCC_ATOMIC_sum TCC_WRITE_sum TCC_READ_sum" "TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum TCC_EA0_ATOMIC_sum TCC_HIT_sum" )
run() { # name envs...
  local name=$1; shift; local gi=0
  for g in "${GROUPS_[@]}"; do
    env "$@" rocprofv3 --kernel-trace --pmc $g --output-format csv -d $OUT/${name}_g$gi -o p -- python $GRAFT_REPO_ROOT/bench_micro/hostile_timing.py 100000000 31 2 0x3 > $OUT/${name}_g$gi.log 2>&1
    gi=$((gi+1))
  done
}
run capped X=1
run exact_cur32 CDBG_SCAN_MODE=exact
run exact_index_plus_load CDBG_SCAN_MODE=exact CDBG_EXACT_NO_CUR32=1
python - <<'PY'
import csv, glob, os, collections
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/scan_req_pmc"
for name in ("capped", "exact_cur32", "exact_index_plus_load"):
    tot = {}; durs = []
    for d in sorted(glob.glob(root + "/" + name + "_g*")):
        if not os.path.isdir(d): continue
        per = collections.defaultdict(lambda: collections.defaultdict(float)); dur = {}; grid = {}
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                kn = r["Kernel_Name"]
                if not (kn.startswith("void cdbg::k_scan_fast<1, 2, 15>") or kn.startswith("void cdbg::k_scan_fast<1, 1, 15>")): continue
                per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
                if "Start_Timestamp" in r: dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        if not per: continue
        disp = sorted(per, key=lambda x: int(x))[-1]
        tot.update(per[disp]); durs.append(dur.get(disp, float("nan")))
    print("%-24s emit launch %s ms   " % (name, " / ".join("%.1f" % x for x in durs)) + "  ".join("%s=%.4g" % (k, v) for k, v in sorted(tot.items())))
PY
rm -rf $OUT/*/
