#!/bin/bash
# dev tool: where does the CLI's ingest time go?  kernel + memory-copy totals (rocprofv3 --stats) of `bcalm` on a synthetic FASTA, with and without the streaming scan
N=${1:-30000000}; T=${2:-16}
D=${CDBG_E2E_DIR:-/tmp}/cli_prof; rm -rf $D && mkdir -p $D && cd $D && export TMPDIR=/tmp
python - <<PY
import sys
sys.path.insert(0, "$GRAFT_REPO_ROOT")
import bcalm_amd
g = bcalm_amd.Graph(31, 2)
g.generate_reads($N, 150, 3)
with open("reads.fa", "wb") as f:
    step = 151 * 1000000
    for off in range(0, $N * 151, step):
        chunk = g.read_text(off, min(step, $N * 151 - off))
        f.write(b">r\n" + chunk[:-1].replace(b"\n", b"\n>r\n") + b"\n")
g.close()
PY
B=$GRAFT_REPO_ROOT/bcalm_amd/_build/bcalm
for mode in "" "-no-stream-scan"; do
  echo "== plain run $mode"; CDBG_HOST_MARKS=1 $B -in reads.fa -kmer-size 31 -abundance-min 2 -nb-cores $T $mode -out p 2>&1 | grep "host:\|GPU:\|\[host\]"
  echo "== rocprofv3 $mode"; rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $D/prof$mode -o p -- $B -in reads.fa -kmer-size 31 -abundance-min 2 -nb-cores $T $mode -out p 2>&1 | grep "host:"
  for f in $(find $D/prof$mode -name "*_stats.csv" | sort); do echo "-- $(basename $f)"; head -8 $f | cut -c1-160; done
done
rm -rf $D
