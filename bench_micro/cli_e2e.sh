#!/bin/bash
# dev tool: end-to-end run of the bcalm CLI on a synthetic FASTA (reads given as $1, default 20M x 150 bp), with timing
set -e
N=${1:-20000000}
cd /tmp && rm -rf cli_e2e && mkdir cli_e2e && cd cli_e2e
python - <<PY
import sys
sys.path.insert(0, "$GRAFT_REPO_ROOT")
import bcalm_amd
g = bcalm_amd.Graph(31, 2)
g.generate_reads($N, 150, 3)
with open("reads.txt", "wb") as f:
    step = 151 * 1000000
    for off in range(0, $N * 151, step):
        f.write(g.read_text(off, min(step, $N * 151 - off)))
g.close()
PY
awk '{print ">" NR "\n" $0}' reads.txt > reads.fa
ls -la reads.fa | awk '{print "fasta bytes", $5}'
for rep in 1 2; do ( time $GRAFT_REPO_ROOT/bcalm_amd/_build/bcalm -in reads.fa -kmer-size 31 -abundance-min 2 | grep "input:\|GPU:\|unitigs written" ) 2>&1 | grep -v "^$\|user\|sys"; done
grep -c ">" reads.unitigs.fa
