#!/bin/bash
# dev tool: a larger FASTQ.gz through the CLI (default 20 M x 150 bp = 6.5 GB of text), compressed as 64 gzip members in parallel (gzip -6 of one stream would take 8 minutes):
# many waves, member ends inside chunks.  r06_gz_ingest_big.sh [reads] [tag]
N=${1:-20000000}; tag=${2:-r06j}; R=$GRAFT_REPO_ROOT; L=$R/gpurun_out/${tag}_gz_ingest_big.log
D=/tmp/gz_ingest; rm -rf $D && mkdir -p $D && cd $D
python - <<PY
import sys
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
t0=$(date +%s%N); split -n l/64 reads.fq piece_; ls piece_* | xargs -P 64 -n 1 gzip -6; cat piece_*.gz > reads.fq.gz; rm -f piece_*
echo "# 64 members, gzip -6 in parallel: $(ls -l reads.fq.gz | awk '{print $5}') bytes in $(( ($(date +%s%N) - t0) / 1000000 )) ms"
echo "== bcalm, plain FASTQ, -nb-cores 32"; run $B -in reads.fq -kmer-size 31 -abundance-min 2 -nb-cores 32 -out p 2>&1 | grep "input:\|host:\|wall\|EXCEPTION"
echo "== bcalm, FASTQ.gz, ONE zlib thread (BCALM_GZ_SERIAL=1)"; BCALM_GZ_SERIAL=1 run $B -in reads.fq.gz -kmer-size 31 -abundance-min 2 -nb-cores 32 -out s 2>&1 | grep "input:\|host:\|wall\|EXCEPTION"
for t in 16 32 60; do
  echo "== bcalm, FASTQ.gz, -nb-cores $t (all threads inflate)"; BCALM_GZ_VERBOSE=1 python -c "import resource, subprocess, sys; rc = subprocess.call(sys.argv[1:]); print('max RSS %.1f GB (exit %d)' % (resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss / 1048576.0, rc))" $B -in reads.fq.gz -kmer-size 31 -abundance-min 2 -nb-cores $t -out z$t 2>&1 | grep "input:\|host:\|EXCEPTION\|inflated\|max RSS"
done
python - <<PY
import hashlib
comp = bytes.maketrans(b"ACGT", b"TGCA")
for f in ("p", "s", "z16", "z32", "z60"):
    seqs = [l.rstrip(b"\n") for l in open(f + ".unitigs.fa", "rb") if not l.startswith(b">")]
    canon = sorted(min(x, x.translate(comp)[::-1]) for x in seqs)
    print("unitigs of %-3s: %d, md5 of the sorted canonical sequences %s" % (f, len(canon), hashlib.md5(b"\n".join(canon)).hexdigest()))
PY
} 2>&1 | tee $L | cut -c1-400
rm -rf $D
