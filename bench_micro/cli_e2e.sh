#!/bin/bash
# dev tool: end-to-end run of the bcalm CLI on a synthetic FASTA (reads given as $1, default 5M x 150 bp), with timing
set -e
N=${1:-5000000}
cd /tmp && rm -rf cli_e2e && mkdir cli_e2e && cd cli_e2e
python - <<PY
import sys
sys.path.insert(0, "$GRAFT_REPO_ROOT/tests")
import oracle_lib
orc = oracle_lib.load()
open("reads.txt", "wb").write(orc.synth_reads($N, 150, 3))
PY
awk '{print ">" NR "\n" $0}' reads.txt > reads.fa
ls -la reads.fa | awk '{print "fasta bytes", $5}'
( time $GRAFT_REPO_ROOT/bcalm_amd/_build/bcalm -in reads.fa -kmer-size 31 -abundance-min 2 -gfa | tail -6 ) 2>&1 | grep -v "^$\|user\|sys"
head -c 300 reads.unitigs.fa; echo; grep -c ">" reads.unitigs.fa; grep -c "^L" reads.unitigs.gfa
$GRAFT_REPO_ROOT/bcalm_amd/_build/bcalm_tools abundance_stats reads.unitigs.fa | head -5
gzip -1 -c reads.fa > reads.fa.gz
( time $GRAFT_REPO_ROOT/bcalm_amd/_build/bcalm -in reads.fa.gz -kmer-size 31 -abundance-min 2 -out gz | grep "input:\|unitigs written" ) 2>&1 | grep -v "^$\|user\|sys"
python - <<PY
comp = str.maketrans("ACGT", "TGCA")
def canon(path):
    out = []
    for line in open(path):
        if line[0] != ">":
            x = line.strip(); r = x.translate(comp)[::-1]; out.append(min(x, r))
    return sorted(out)
a, b = canon("reads.unitigs.fa"), canon("gz.unitigs.fa")
print("gz run: same canonical unitig sequences" if a == b else "gz run: DIFFERENT unitig sets (%d vs %d)" % (len(a), len(b)))
PY
