#!/bin/bash
# dev tool: the bcalm CLI end to end on a synthetic FASTA of $1 reads (default 30 M x 150 bp = 4.6 GB), parser / writer threads swept ($2, default "1 4 8 16 32"),
# then FASTQ and gzip forms of a tenth of it.  Prints the CLI's own input: / host: / GPU: lines and the wall clock.
N=${1:-30000000}; SWEEP=${2:-"1 4 8 16 32"}
D=${CDBG_E2E_DIR:-/tmp}/cli_e2e; rm -rf $D && mkdir -p $D && cd $D
python - <<PY
import sys, time
sys.path.insert(0, "$GRAFT_REPO_ROOT")
import bcalm_amd
g = bcalm_amd.Graph(31, 2)
g.generate_reads($N, 150, 3)
t0 = time.time()
with open("reads.fa", "wb") as f, open("small.fq", "wb") as q:
    step = 151 * 1000000
    for off in range(0, $N * 151, step):
        chunk = g.read_text(off, min(step, $N * 151 - off))
        f.write(b">r\n" + chunk[:-1].replace(b"\n", b"\n>r\n") + b"\n")
        if off < $N * 151 // 10:
            q.write(b"@r\n" + chunk[:-1].replace(b"\n", b"\n+\n" + b"I" * 150 + b"\n@r\n") + b"\n+\n" + b"I" * 150 + b"\n")
g.close()
print("wrote reads.fa + small.fq in %.1f s" % (time.time() - t0))
PY
run() { local t0=$(date +%s%N); "$@"; local rc=$?; echo "wall $(( ($(date +%s%N) - t0) / 1000000 )) ms (exit $rc)"; }
ls -la reads.fa small.fq | awk '{print $9, $5, "bytes"}'; nproc
for c in $SWEEP; do
  echo "== -nb-cores $c"; run $GRAFT_REPO_ROOT/bcalm_amd/_build/bcalm -in reads.fa -kmer-size 31 -abundance-min 2 -nb-cores $c -out o$c 2>&1 | grep "input:\|host:\|GPU:\|wall\|EXCEPTION"
done
md5sum o*.unitigs.fa | awk '{print $1}' | sort | uniq -c
echo "== FASTQ, -nb-cores 16"; run $GRAFT_REPO_ROOT/bcalm_amd/_build/bcalm -in small.fq -kmer-size 31 -abundance-min 2 -nb-cores 16 -out q 2>&1 | grep "input:\|host:\|wall\|EXCEPTION"
gzip -1 -k small.fq; echo "== FASTQ.gz (one inflate thread)"; run $GRAFT_REPO_ROOT/bcalm_amd/_build/bcalm -in small.fq.gz -kmer-size 31 -abundance-min 2 -out qz 2>&1 | grep "input:\|host:\|wall\|EXCEPTION"
cmp q.unitigs.fa qz.unitigs.fa > /dev/null && echo "FASTQ and FASTQ.gz outputs identical" || echo "FASTQ / FASTQ.gz outputs differ in order (sets compared by the tests)"
rm -rf $D
