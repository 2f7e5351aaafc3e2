#!/bin/bash
# dev tool (VERDICT r5 missing #8): one FASTQ.gz through the CLI, inflated by one zlib thread (BCALM_GZ_SERIAL=1) and by all threads (pgz.h).
# r06_gz_ingest.sh [reads, default 5 M x 150 bp] [tag].  The FASTQ has numbered headers and noisy qualities (gzip -6: about 2.2 x), not the constant lines of e2e_sweep.sh.
N=${1:-5000000}; tag=${2:-r06g}; R=$GRAFT_REPO_ROOT; L=$R/gpurun_out/${tag}_gz_ingest.log
D=/tmp/gz_ingest; rm -rf $D && mkdir -p $D && cd $D
python - <<PY
import sys, time
import numpy as np
sys.path.insert(0, "$R")
import bcalm_amd
g = bcalm_amd.Graph(31, 2)
g.generate_reads($N, 150, 3)
rng = np.random.default_rng(1)
with open("reads.fq", "wb") as f:
    step = 1000000
    for r0 in range(0, $N, step):
        n = min(step, $N - r0)
        seq = np.frombuffer(g.read_text(r0 * 151, n * 151), dtype=np.uint8).reshape(n, 151)
        hdr = np.frombuffer(("".join("@SRR0000001.%09d\n" % (r0 + i) for i in range(n))).encode(), dtype=np.uint8).reshape(n, 22)
        q = np.clip(rng.normal(68, 5, size=(n, 150)), 35, 74).astype(np.uint8)
        plus = np.tile(np.frombuffer(b"+\n", dtype=np.uint8), (n, 1)); nl = np.full((n, 1), 10, dtype=np.uint8)
        f.write(np.concatenate([hdr, seq, plus, q, nl], axis=1).tobytes())
g.close()
PY
B=$R/bcalm_amd/_build/bcalm
run() { local t0=$(date +%s%N); "$@"; local rc=$?; echo "wall $(( ($(date +%s%N) - t0) / 1000000 )) ms (exit $rc)"; }
{
echo "# $(nproc) host threads; $(ls -l reads.fq | awk '{print $5}') bytes of FASTQ ($N reads x 150 bp)"
t0=$(date +%s%N); gzip -6 -k reads.fq; echo "# gzip -6: $(ls -l reads.fq.gz | awk '{print $5}') bytes in $(( ($(date +%s%N) - t0) / 1000000 )) ms"
echo "== gzip -dc > /dev/null"; run sh -c 'gzip -dc reads.fq.gz > /dev/null'
for t in 2 4 8 16 32 64; do [ $t -le $(( $(nproc) * 1 )) ] && { echo "== pgz_cat, $t threads (no output)"; PGZ_NO_OUTPUT=1 $R/bcalm_amd/_build/pgz_cat reads.fq.gz $t 2>&1; }; done
echo "== pgz_cat | cmp"; $R/bcalm_amd/_build/pgz_cat reads.fq.gz 16 2>/dev/null | cmp - reads.fq && echo "identical to the plain file"
echo "== bcalm, plain FASTQ, -nb-cores 16"; run $B -in reads.fq -kmer-size 31 -abundance-min 2 -nb-cores 16 -out p 2>&1 | grep "input:\|host:\|wall\|EXCEPTION"
echo "== bcalm, FASTQ.gz, ONE zlib thread (BCALM_GZ_SERIAL=1)"; BCALM_GZ_SERIAL=1 run $B -in reads.fq.gz -kmer-size 31 -abundance-min 2 -nb-cores 16 -out s 2>&1 | grep "input:\|host:\|wall\|EXCEPTION"
for t in 8 16 32; do
  echo "== bcalm, FASTQ.gz, -nb-cores $t (all threads inflate)"; BCALM_GZ_VERBOSE=1 run $B -in reads.fq.gz -kmer-size 31 -abundance-min 2 -nb-cores $t -out z$t 2>&1 | grep "input:\|host:\|wall\|EXCEPTION\|inflated"
done
echo "== bcalm, FASTQ.gz, default threads (min(32, CPUs the container grants))"; BCALM_GZ_VERBOSE=1 run $B -in reads.fq.gz -kmer-size 31 -abundance-min 2 -out zd 2>&1 | grep "input:\|host:\|wall\|EXCEPTION\|inflated"
echo "== bcalm, plain FASTQ, default threads"; run $B -in reads.fq -kmer-size 31 -abundance-min 2 -out pd 2>&1 | grep "input:\|host:\|wall\|EXCEPTION"
python - <<PY
import hashlib
comp = bytes.maketrans(b"ACGT", b"TGCA")
for f in ("p", "s", "z8", "z16", "z32"):            # (a unitig is written in the orientation its chain was walked in: compare canonical forms)
    seqs = [l.rstrip(b"\n") for l in open(f + ".unitigs.fa", "rb") if not l.startswith(b">")]
    canon = sorted(min(x, x.translate(comp)[::-1]) for x in seqs)
    print("unitigs of %-3s: %d, md5 of the sorted canonical sequences %s" % (f, len(canon), hashlib.md5(b"\n".join(canon)).hexdigest()))
PY
} 2>&1 | tee $L | cut -c1-400
rm -rf $D
