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
/usr/bin/time -v $GRAFT_REPO_ROOT/bcalm_amd/_build/bcalm -in reads.fa -kmer-size 31 -abundance-min 2 -gfa 2> time.log | tail -6
grep "Elapsed\|Maximum resident" time.log
head -c 300 reads.unitigs.fa; echo; grep -c ">" reads.unitigs.fa; grep -c "^L" reads.unitigs.gfa
$GRAFT_REPO_ROOT/bcalm_amd/_build/bcalm_tools abundance_stats reads.unitigs.fa | head -5
gzip -1 -c reads.fa > reads.fa.gz
/usr/bin/time -v $GRAFT_REPO_ROOT/bcalm_amd/_build/bcalm -in reads.fa.gz -kmer-size 31 -abundance-min 2 -out gz 2> time2.log | grep "input:\|unitigs written"
grep "Elapsed" time2.log
cmp <(grep -v ">" reads.unitigs.fa | sort | md5sum) <(grep -v ">" gz.unitigs.fa | sort | md5sum) && echo "gz run: same unitig sequences"
