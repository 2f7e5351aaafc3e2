#!/usr/bin/env python3
"""bench.py -- reads -> unitigs throughput on MI355X (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--cfg 3|4|5]
  N > 1:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one full pass of the hot path (count -> compact -> glue) over one resident
batch of synthetic reads.  Workload at N=1: BASELINE config 3 (100 M x 150 bp synthetic
reads, k=31, abundance-min 2), generated directly in HBM by the counter-based generator
of BASELINE.md section 2, so inputs are resident when the timed region starts.

value   = distinct canonical k-mers processed per second, whole job (all ranks).
roofline = the dominant kernel's ALGORITHMIC bytes (SURVEY.md section 8d stage-interface
          model, evaluated with the measured n_occ / S / P / U) / its launch duration,
          measured with HIP events on the library's own stream (cdbg_stats ms_*), against
          the 8 TB/s HBM3E peak.  `pipeline` inside it is the same for the whole step.
          With deferred record placement (one-word k-mers: half of the records are scattered by
          k_place on a second HIP stream WHILE k_count_fast counts the other half) the scan kernel
          and the count stage are rated as ONE entry -- the bytes of both over the wall of both,
          the placement stream's busy span listed beside it, not added -- see `roofline.kernels`.
cpu_baseline = on the GPU box's host cores, a bounded sample of the same workload (10 M reads
          at k <= 31): a real BCALM 2 binary when one is reachable ($BCALM_BIN / `bcalm` on PATH;
          kind "reference", its unitig set is then diffed against the GPU's), otherwise the
          multithreaded CPU restatement of the spec oracle/cpu_mt.cpp (kind "port", all cores,
          one shared lock-free table; its set digest must equal the GPU's on the same sample).
          A reported baseline, not the target.  The reference itself cannot be built here
          (its gatb-core submodule is absent).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
def pmc_csv(cfg):
    """the newest committed PMC table of this config (profiles/rNN_pmc_hbm_traffic_per_kernel_cfgC.csv)"""
    import glob
    c = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]*_pmc_hbm_traffic_per_kernel_cfg%d.csv" % cfg)))
    return c[-1] if c else os.path.join(ROOT, "profiles", "none.csv")


def lib_srchash():
    """content hash of the kernel sources the loaded libcdbg.so was built from (__graft_entry__.build writes it)"""
    try:
        return open(os.path.join(ROOT, "bcalm_amd", "_build", "libcdbg.so.srchash")).read().strip()
    except Exception:
        return None


GLUE_LABEL = "glue(k_join_bucket+k_walk_init+k_walk_measure+k_walk_place+k_walk_copy)"


def pmc_traffic(kernel_key, cfg):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of this same bench
    (bench_micro/pmc_bench.sh: separate --pmc FETCH_SIZE / WRITE_SIZE runs; bytes = (f * FETCH_SIZE + WRITE_SIZE) * 1024 with
    the per-access-pattern factor f calibrated in profiles/r03_counter_calibration.csv: 2 for wide coalesced reads -- the
    gfx950 FETCH_SIZE halving of MI355X_MICROARCH.md section HBM -- 1 for kernels whose reads are random gathers).
    -> (bytes or None, first line of the CSV: the commit it was taken at)"""
    keys = kernel_key if isinstance(kernel_key, (list, tuple)) else [kernel_key]
    tot, found, stamp = 0.0, False, None
    try:
        for line in open(pmc_csv(cfg)):
            if line.startswith("#"):
                stamp = line[1:].strip(); continue
            row = line.rstrip("\n").rsplit(",", 5)          # kernel names contain commas
            if len(row) == 6 and row[1].isdigit() and any(x in row[0] for x in keys):
                tot += float(row[5]) * float(row[1]) / 2 * 1e9 if len(keys) > 1 else float(row[5]) * 1e9   # (a stage: all launches of its kernels in one of the 2 profiled steps)
                found = True
                if len(keys) == 1:
                    break
    except Exception:
        return None, None
    return (tot if found else None), stamp


def W_OF(k):
    """64-bit words per k-mer: the reference's span rule k < 32 W (README.md:91-99), W = 1 .. 8 (k <= 255)"""
    return k // 32 + 1


def alg_bytes(k, st, n_reads, read_len):
    """SURVEY.md section 8(d) stage-interface bytes, with measured counts."""
    W = W_OF(k)
    Kb = 8 * W
    sbar = (k - 10 + 2) / 2.0
    n_occ, S, P, U = st["n_occurrences"], st["n_solid"], st["n_pieces"], st["n_unitigs"]
    A1 = n_reads * read_len
    A2 = 2 * n_occ * (1 + (k - 1) / sbar) / 4
    A3 = 2 * S * (Kb + 4)
    A4 = 2 * ((S + P * (k - 1)) / 4 + 16 * P)
    A5 = (S + U * (k - 1)) / 4 + 16 * U
    per_kernel = {
        "k_scan<hist>": A1,                    # reads every ASCII base once
        "k_scan<emit>": A1 + A2 / 2,           # reads bases again, writes the super-k-mer records
        "k_count_fast": A2 / 2 + A3 / 2,       # reads records, writes solid (k-mer, count)   (+ k_count for multi-pass partitions)
        "k_compact_wave": A3 / 2 + A4 / 2,     # reads solid k-mers, writes pieces + glue records   (+ k_compact for big buckets)
        GLUE_LABEL: A4 / 2 + A5,
    }
    return per_kernel, A1 + A2 + A3 + A4 + A5


def _cpu_worker(args):
    """one scalar run of the CPU port on its own read sample (separate process: the port is single-threaded)"""
    k, amin, read_len, cfg, sample_reads = args
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    orc = oracle_lib.load()
    text = orc.synth_reads(sample_reads, read_len, cfg)
    t0 = time.time()
    r = orc.run(text, k, amin)
    return r["stats"]["distinct"], time.time() - t0


def _reference_binary():
    """a real BCALM 2 binary, if the box happens to have one"""
    import shutil
    cand = os.environ.get("BCALM_BIN") or shutil.which("bcalm")
    if cand and os.path.isfile(cand) and os.path.realpath(cand) != os.path.realpath(os.path.join(ROOT, "bcalm_amd", "_build", "bcalm")):
        return cand
    return None



def _cpus_granted():
    """CPUs this process may use: affinity mask and the container's CPU quota (cgroup v2 cpu.max / v1 cfs quota) -- a 256-thread host may grant 16"""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, -(-int(q) // int(per))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                n = min(n, max(1, -(-q // per)))
        except (OSError, ValueError):
            pass
    return n

def cpu_baseline(k, amin, read_len, cfg, sample_reads):
    """CPU baseline on the host cores of this box (reported, not the target).
    Preferred: the reference's own binary ($BCALM_BIN / `which bcalm`) on a FASTA dump of the sample, all cores.
    Otherwise (expected: gatb-core is absent, nothing to build): the scalar CPU port of the spec (oracle/), one
    process per core on up to 32 cores, each on its own sample of the same generator -- the aggregate rate of an
    embarrassingly parallel use of the port, which flatters the CPU (no shared table, no merge)."""
    import subprocess
    import tempfile
    ref = _reference_binary()
    if ref:
        try:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib
            orc = oracle_lib.load()
            text = orc.synth_reads(sample_reads, read_len, cfg).decode()
            with tempfile.TemporaryDirectory() as t:
                fa = os.path.join(t, "sample.fa")
                with open(fa, "w") as f:
                    for i, line in enumerate(text.split("\n")):
                        if line:
                            f.write(">r%d\n%s\n" % (i, line))
                cores = os.cpu_count() or 1
                t0 = time.time()
                subprocess.run([ref, "-in", fa, "-kmer-size", str(k), "-abundance-min", str(amin), "-nb-cores", str(cores)],
                               cwd=t, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
                dt = time.time() - t0
                # "also diff its canonicalised unitig set against ours" -- the reference's output on
                # this sample against the HIP path's (tests/parity.py), printed with the line
                import re as _re
                ref_ut, kc = [], None
                for line in open(os.path.join(t, "sample.unitigs.fa")):
                    if line.startswith(">"):
                        kc = int(_re.search(r"KC:i:(\d+)", line).group(1))
                    elif line.strip():
                        ref_ut.append((line.strip(), kc))
            distinct = orc.run(text, k, amin)["stats"]["distinct"]
            import bcalm_amd
            g = bcalm_amd.Graph(k, amin, lib=bcalm_amd.load())
            g.push_text(text.encode()); g.run(); ours = g.unitigs(); g.close()
            a = oracle_lib.canonical_set(orc, ours, k); b = oracle_lib.canonical_set(orc, ref_ut, k)
            return {"value": distinct / dt, "unit": "kmers/s", "cores": cores, "kind": "reference",
                    "sample": f"{sample_reads} x {read_len} bp synthetic reads through {ref} in {dt:.1f} s",
                    "unitig_sets_equal": a == b, "unitigs_ours": len(a), "unitigs_reference": len(b),
                    "only_ours": sorted(set(a) - set(b))[:3], "only_reference": sorted(set(b) - set(a))[:3]}
        except Exception:
            pass                                             # fall through to the port
    if k <= 63:
        # the multithreaded shared-table restatement (oracle/cpu_mt.cpp: std::thread x all cores, ONE input, one lock-free
        # table; SURVEY.md section 8 d ii), pinned against the oracle in tests/test_oracle.py
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib
        orc = oracle_lib.load()
        granted = _cpus_granted()
        # threads: all hardware threads -- unless the container grants fewer CPUs than the host shows: 256 threads on a 16-CPU grant run the restatement 1.5 x
        # slower than 32 (bench_micro/cpu_baseline_threads.py on the GPU box: 8.1 M vs 12.4 M distinct k-mers/s; 16 threads: 10.9 M), so: twice the grant
        cores = min(os.cpu_count() or 1, 2 * granted)
        text = orc.synth_reads(sample_reads, read_len, cfg)
        r = oracle_lib.cpu_mt_run(text, k, amin, cores)
        return {"value": r["distinct"] / r["s_total"], "unit": "kmers/s", "cores": cores, "cpus_granted": granted, "kind": "port",
                "seconds": {"count": r["s_count"], "solid_table": r["s_solid"], "unitigs": r["s_unitigs"], "total": r["s_total"]},
                "set_digest": "%016x" % r["set_digest"], "distinct": r["distinct"], "solid": r["solid"], "unitigs": r["unitigs"],
                "sample": f"{sample_reads} x {read_len} bp synthetic reads (same generator, its own 30x genome), {r['distinct']} distinct k-mers, "
                          f"{cores} threads (the container grants {granted} CPUs) on one shared lock-free table, reads -> unitigs in {r['s_total']:.2f} s; multithreaded CPU restatement "
                          f"of the spec (oracle/cpu_mt.cpp), NOT BCALM 2 (its gatb-core sources are absent)"}
    cores = max(1, min(os.cpu_count() or 1, 32, 2 * _cpus_granted()))
    t0 = time.time()
    res = []
    if cores > 1:                                            # one child process per core (never a Pool: a bench must not hang)
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", "--k", str(k), "--abundance-min", str(amin),
                                   "--read-len", str(read_len), "--cfg", str(cfg & 0xF), "--gen-cfg", str(cfg + 16 * i), "--cpu-sample-reads", str(sample_reads)],
                                  stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for i in range(cores)]
        for p in procs:
            try:
                out, _ = p.communicate(timeout=300)
                d, dt = out.strip().split()[-2:]
                res.append((int(d), float(dt)))
            except Exception:
                p.kill()
    if not res:
        cores = 1
        res = [_cpu_worker((k, amin, read_len, cfg, sample_reads))]
    cores = len(res)
    wall = time.time() - t0
    total = sum(d for d, _ in res)
    one = res[0][0] / res[0][1]
    slowest = max(dt for _, dt in res)
    return {"value": total / slowest, "unit": "kmers/s", "cores": cores, "kind": "port",
            "single_core_value": one,
            "sample": f"{cores} processes x {sample_reads} x {read_len} bp synthetic reads (same generator, 30x coverage, one sample "
                      f"per core), {total} distinct k-mers, slowest process {slowest:.1f} s ({wall:.1f} s with start-up); "
                      f"scalar CPU restatement of the spec run once per core, NOT BCALM2 (its gatb-core sources are absent)"}


def cli_end_to_end(lib, k, amin, read_len, gen_cfg, n_reads, device_id):
    """T_e2e: `bcalm -in reads.fa -kmer-size k -abundance-min a` (the repo's CLI, bcalm_amd/host/bcalm_main.cpp) on a FASTA dump of
    n_reads reads of the bench's generator: process start + HIP init + parse + pinned H2D (overlapped with the scan) + the three
    stages + links + D2H + FASTA write.  Wall clock of the child process; its own report is kept."""
    import re
    import shutil
    import subprocess
    import tempfile
    import bcalm_amd
    exe = os.path.join(ROOT, "bcalm_amd", "_build", "bcalm")
    tmp = tempfile.mkdtemp(prefix="cdbg_e2e_", dir=os.environ.get("CDBG_E2E_DIR") or None)
    try:
        fa = os.path.join(tmp, "reads.fa")
        g = bcalm_amd.Graph(k, amin, lib=lib, device_id=device_id)
        g.generate_reads(n_reads, read_len, gen_cfg)
        rec = read_len + 1
        with open(fa, "wb") as f:
            step = rec * 1_000_000
            for off in range(0, n_reads * rec, step):
                chunk = g.read_text(off, min(step, n_reads * rec - off))
                f.write(b">r\n" + chunk[:-1].replace(b"\n", b"\n>r\n") + b"\n")
        g.close(); g.release_cached()
        nbytes = os.path.getsize(fa)
        t0 = time.perf_counter()
        p = subprocess.run([exe, "-in", fa, "-kmer-size", str(k), "-abundance-min", str(amin), "-out", os.path.join(tmp, "e2e")],
                           capture_output=True, text=True, timeout=900, cwd=tmp,
                           env={x: y for x, y in os.environ.items() if x != "CDBG_FORCE_MULTI"})   # (--force-dist sets it for THIS process: the CLI would take it as its own test hook)
        wall = time.perf_counter() - t0
        ufa = os.path.join(tmp, "e2e.unitigs.fa")
        m = re.search(r"graph: (\d+) pieces -> (\d+) unitigs", p.stdout)
        rep = [line for line in p.stdout.split("\n") if line.startswith(("input:", "host:", "GPU:"))]
        split = {}
        for name, pat in (("init_s", r"host: init ([\d.]+) s"), ("ingest_s", r"ingest ([\d.]+) s"), ("ingest_GB_per_s", r"ingest [\d.]+ s = ([\d.]+) GB/s"),
                          ("stages_s", r"stages ([\d.]+) s"), ("links_d2h_s", r"links \+ D2H ([\d.]+) s"), ("write_s", r"write ([\d.]+) s ="),
                          ("write_GB_per_s", r"write [\d.]+ s = ([\d.]+) GB/s"), ("threads", r"\((\d+) threads\)")):
            mm = re.search(pat, p.stdout)
            if mm:
                split[name] = float(mm.group(1))
        return {"ok": p.returncode == 0 and os.path.exists(ufa) and os.path.getsize(ufa) > 0, "wall_s": wall, "fasta_bytes": nbytes, "reads": n_reads,
                "unitigs": int(m.group(2)) if m else None, "unitig_file_bytes": os.path.getsize(ufa) if os.path.exists(ufa) else 0,
                "input_GB_per_s": nbytes / wall / 1e9, "split": split, "cli_report": rep,
                "what": "wall clock of `bcalm -in reads.fa -kmer-size %d -abundance-min %d` on a %.2f GB FASTA of the same generator: process start, HIP init, "
                        "parse + pinned H2D overlapped with the scan, count, compact, glue, links, D2H, FASTA write" % (k, amin, nbytes / 1e9)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--cfg", type=int, default=3, choices=[3, 4, 5],
                    help="BASELINE config: 3 = 100 M x 150 bp, k = 31 (the metric's config, fits one GPU); 4 = k = 55 and 5 = k = 127 are 8-GPU "
                         "configs: at N = 1 the bench runs the share of ONE of the eight GPUs (125 M x 150 bp / 6.25 M x 1 kbp)")
    ap.add_argument("--reads", type=int, default=None, help="reads per GPU; default: the config's (smaller values are dev runs, not the metric)")
    ap.add_argument("--read-len", type=int, default=None)
    ap.add_argument("--k", type=int, default=None)
    ap.add_argument("--abundance-min", type=int, default=2)
    ap.add_argument("--cpu-sample-reads", type=int, default=None, help="reads of the CPU baseline's sample (default: 10 M for k <= 31, 6 M for k <= 63 -- the multithreaded restatement; 400 K per process of the scalar port beyond)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end leg (the bcalm CLI on a FASTA dump: t_e2e_s)")
    ap.add_argument("--e2e-reads", type=int, default=None, help="reads of the end-to-end FASTA (default: >= 1 GB of sequence)")
    ap.add_argument("--cpu-worker", action="store_true", help=argparse.SUPPRESS)   # child of cpu_baseline: one scalar run, prints 'distinct seconds'
    ap.add_argument("--gen-cfg", type=int, default=None, help=argparse.SUPPRESS)   # (child only: generator seed of its own sample)
    ap.add_argument("--mode", choices=["sharded", "independent"], default="sharded",
                    help="N>1: sharded = ONE graph over --reads reads: reads and minimizer partitions split over the ranks, records and glue "
                         "data exchanged over RCCL inside libcdbg (strong scaling); independent = one read set per rank, no collective (weak scaling)")
    ap.add_argument("--skewed", action="store_true",
                    help="the HOSTILE generator (cfg | 0x100, include/cdbg.h): two-letter low-complexity blocks, up to 1000 copies of a 5 kbp repeat, "
                         "50 homopolymer runs, ~20x coverage skew -- what the overflow tiers (spill repair, second count tier, multi-pass, HBM tables) cost; "
                         "a robustness line, not the metric")
    ap.add_argument("--force-dist", action="store_true", help="run the sharded (collective) code path even with one rank (testing)")
    ap.add_argument("--read-placement", dest="read_placement", choices=["auto", "replicated", "sharded"], default="auto",
                    help="N>1, --mode sharded: replicated = every rank generates ALL reads and scans them for its own partitions, no record "
                         "exchange (SURVEY.md 8e X0); sharded = every rank holds 1/N of the reads, super-k-mer records all-to-all-v'd to the "
                         "partition owners (X1); auto = replicated on 2-4 GPUs (one xGMI link per peer would carry 1/N^2 of 25.6 GB), sharded on 8")
    a = ap.parse_args()
    CFG = {3: dict(k=31, read_len=150, reads=100_000_000, name="BASELINE config 3"),
           4: dict(k=55, read_len=150, reads=125_000_000, name="BASELINE config 4, the share of one of its 8 GPUs (1 G reads / 8)"),
           5: dict(k=127, read_len=1000, reads=6_250_000, name="BASELINE config 5, the share of one of its 8 GPUs (50 M reads / 8)")}[a.cfg]
    a.k = a.k or CFG["k"]; a.read_len = a.read_len or CFG["read_len"]
    a.reads = a.reads or int(os.environ.get("CDBG_BENCH_READS", CFG["reads"]))
    a.cpu_sample_reads = a.cpu_sample_reads or (10_000_000 if a.k <= 31 else 6_000_000 if a.k <= 63 else 400_000)
    # end-to-end leg: >= 4.5 GB of FASTA by default (the full read set of the config with --e2e-reads = --reads: 15.4 GB at config 3; the
    # default keeps the whole bench within minutes -- writing the FASTA takes longer than the CLI needs for it)
    a.e2e_reads = a.e2e_reads or min(a.reads, (int(os.environ.get("CDBG_E2E_BYTES", 4_500_000_000)) + a.read_len) // (a.read_len + 4))
    gen_cfg = a.cfg | (0x100 if a.skewed else 0)           # generator seed / mode (cdbg_generate_reads)
    if a.cpu_worker:
        d, dt = _cpu_worker((a.k, a.abundance_min, a.read_len, a.gen_cfg if a.gen_cfg is not None else a.cfg, a.cpu_sample_reads))
        print(d, dt)
        return

    # the one JSON line must be the only thing on stdout: RCCL / HIP runtime banners written to fd 1 by native
    # code go to stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    json_out = os.fdopen(json_fd, "w")

    import torch
    import bcalm_amd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1 or a.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(local_rank)
    assert a.gpus == world, f"--gpus {a.gpus} but WORLD_SIZE={world}"

    lib = bcalm_amd.load()                      # raises without the HIP extension: no fallback
    sharded = (world > 1 or a.force_dist) and a.mode == "sharded"
    if sharded:
        # ONE graph over `--reads` reads, entirely inside libcdbg (include/cdbg.h "Multi-GPU"): every rank holds a
        # contiguous 1/N shard of the read set, scans it, all-to-all-v's the super-k-mer records to the partition owners
        # (RCCL ncclSend/ncclRecv), counts + compacts its partitions, all-gathers pieces + junction log, joins sharded,
        # ranks, and emits the unitigs whose first piece it owns.  torch.distributed only carries the ncclUniqueId.
        from bcalm_amd import dist as cdist
        if a.force_dist and world == 1:
            os.environ["CDBG_FORCE_MULTI"] = "1"
        replicated = a.read_placement == "replicated" or (a.read_placement == "auto" and 1 < world <= 4)
        g = bcalm_amd.Graph(a.k, a.abundance_min, lib=lib, device_id=local_rank, world_size=world, rank=rank, reads_replicated=replicated)
        cdist.init_rccl(g, dist, device=torch.device("cuda", local_rank))
        if replicated:
            g.generate_reads(a.reads, a.read_len, gen_cfg, first_read=0, total_reads=a.reads)
        else:
            share = (a.reads + world - 1) // world
            first = rank * share
            g.generate_reads(max(0, min(share, a.reads - first)), a.read_len, gen_cfg, first_read=first, total_reads=a.reads)

        def step():
            g.run()
            return {"reads": "replicated on every rank (X0: no record exchange)" if replicated else "sharded (X1: records all-to-all-v'd to the partition owners)",
                    "transport": "RCCL inside libcdbg.so: all-to-all-v only -- junction records to their key owners, joined pairs to the end owners, "
                                 "ranking queries / replies per round, every piece once to the owner of its unitig's head",
                    "ranking_rounds": g.stats()["n_glue_rounds"],
                    "bytes_sent_plus_received_by_rank0": g.comm_bytes()}
    else:
        # N == 1, or --mode independent: every rank runs the full path on its own read set
        g = bcalm_amd.Graph(a.k, a.abundance_min, lib=lib, device_id=local_rank)
        g.generate_reads(a.reads, a.read_len, gen_cfg + 16 * rank)

        def step():
            g.run()
            return None

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
        g.reset()
    sync()
    t0 = time.perf_counter()
    st = None
    acc = {x: 0.0 for x in ("ms_scan_hist", "ms_scan_emit", "ms_count", "ms_place", "ms_compact", "ms_glue", "ms_total", "ms_exchange")}
    xinfo = None
    for i in range(a.steps):
        xinfo = step()
        st = g.stats()
        for x in acc:
            acc[x] += st[x]
        if i + 1 < a.steps:
            g.reset()
    sync()
    dt = time.perf_counter() - t0
    n_distinct = st["n_distinct"]
    # ---- the bench checks its own output (outside the timed region): size-independent invariants at full size ----
    dig_last = g.digest()
    checks = {}
    if not sharded:
        g.reset(); step()
        checks["set_digest_equal_across_two_steps"] = g.digest()["set_digest"] == dig_last["set_digest"]
    tot = {"n_occurrences": st["n_occurrences"], "n_solid": st["n_solid"], "n_unitigs": st["n_unitigs"], "unitig_bases": st["unitig_bases"],
           "kc_sum": dig_last["kc_sum"], "solid_count_sum": dig_last["solid_count_sum"], "kmers_in_unitigs": dig_last["kmers_in_unitigs"],
           "set_digest": dig_last["set_digest"]}
    if sharded and dist is not None:
        # every rank holds the unitigs it owns: sums (and the additive set digest) over the ranks describe the whole graph
        keys = sorted(tot)
        v = torch.tensor([(tot[x] + (1 << 63)) % (1 << 64) - (1 << 63) for x in keys], device="cuda", dtype=torch.int64)   # two's complement: wraps like uint64
        dist.all_reduce(v, op=dist.ReduceOp.SUM)
        tot = {x: int(v[i].item()) % (1 << 64) for i, x in enumerate(keys)}
    exp_occ = a.reads * (a.read_len - a.k + 1)                   # synthetic reads carry no N: every position is a k-mer
    if sharded or world == 1:
        checks["n_occurrences == R*(L-k+1)"] = tot["n_occurrences"] == exp_occ
    checks["sum(LN-k+1) == n_solid"] = tot["kmers_in_unitigs"] == tot["n_solid"]
    checks["sum(KC) == sum(solid counts)"] = tot["kc_sum"] == tot["solid_count_sum"]
    checks["unitig_bases == n_solid + U*(k-1)"] = tot["unitig_bases"] == tot["kmers_in_unitigs"] + tot["n_unitigs"] * (a.k - 1)
    dig_last = dict(dig_last, set_digest=tot["set_digest"], kc_sum=tot["kc_sum"], kmers_in_unitigs=tot["kmers_in_unitigs"])

    # ---- the unitig DEFINITION at full size, without the oracle (cdbg_verify, bcalm_amd/csrc/k_verify.h): the canonical k-mers
    # spelled by the unitigs are exactly the solid set, each once (position count + two commutative 64-bit sums); no two unitig
    # ends are each other's only link (maximality; single graph per rank only: a rank of a sharded job holds no link table) ----
    vr = g.verify()
    vu, vs = list(vr["unitig_kmers"]), list(vr["solid_kmers"])
    if sharded and dist is not None:
        v = torch.tensor([(x + (1 << 63)) % (1 << 64) - (1 << 63) for x in vu + vs], device="cuda", dtype=torch.int64)
        dist.all_reduce(v, op=dist.ReduceOp.SUM)
        allv = [int(x.item()) % (1 << 64) for x in v]
        vu, vs = allv[:3], allv[3:]
    checks["k-mers spelled by the unitigs == solid set (count, 2 x 64-bit commutative sums)"] = vu == vs and vu[0] == tot["n_solid"]
    if vr["mergeable_ends"] is not None:
        checks["no two unitig ends are each other's only link (every unitig maximal)"] = vr["mergeable_ends"] == 0
    # edge conservation (cdbg_verify_edges; bidirected-graphs-in-bcalm2.md:85): the edges of the solid k-mer graph, counted from the
    # count stage's keys in one global (k-1)-mer table, are exactly the inner adjacencies of the unitigs plus the links between
    # their ends -- a unitig that runs THROUGH a branching junction (over-compaction) leaves a strictly positive shortfall
    ve = vr["edges"]
    if ve is not None:
        checks["edges of the solid graph == links + 2 x inner adjacencies (every inner junction 1-in / 1-out)"] = ve["graph"] == ve["links"] + ve["inner"]
    verify_info = {"unitig_kmers": [vu[0], "%016x" % vu[1], "%016x" % vu[2]], "solid_kmers": [vs[0], "%016x" % vs[1], "%016x" % vs[2]],
                   "mergeable_ends": vr["mergeable_ends"], "closed_chains_cut": vr["closed_chains"], "edges": ve}
    if vr.get("edges_error"):                              # (the edge pass could not run -- table too large for the card: said, not hidden)
        verify_info["edges_error"] = vr["edges_error"]

    if dist is not None:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        n = torch.tensor([n_distinct], device="cuda", dtype=torch.int64)
        dist.all_reduce(n, op=dist.ReduceOp.SUM)
        total_distinct = int(n.item())
    else:
        total_distinct = n_distinct

    if rank == 0:
        per_kernel, alg_total = alg_bytes(a.k, st, a.reads, a.read_len)
        ms = {"k_scan<hist>": acc["ms_scan_hist"], "k_scan<emit>": acc["ms_scan_emit"], "k_count_fast": acc["ms_count"],
              "k_compact_wave": acc["ms_compact"], GLUE_LABEL: acc["ms_glue"]}
        # Deferred record placement (count_slices > 1): part of the records is scattered by k_place on a second HIP stream WHILE
        # k_count_fast counts the slice before, so "the scan" and "the count" are no longer two kernels one after the other.  They are rated as ONE entry:
        # the algorithmic bytes of both (A1 + A2 + A3/2) over the WALL of the pair -- the scan kernel plus the count stage, which contains every wait for
        # the placement; the placement stream's busy span is listed beside it and is not a summand.
        PAIR = "k_scan<emit> + (k_place || k_count_fast)"
        deferred = st.get("count_slices", 1) > 1
        if deferred:
            per_kernel[PAIR] = per_kernel["k_scan<emit>"] + per_kernel["k_count_fast"]
            ms = {"k_scan<hist>": ms["k_scan<hist>"], PAIR: ms["k_scan<emit>"] + ms["k_count_fast"], "k_compact_wave": ms["k_compact_wave"], GLUE_LABEL: ms[GLUE_LABEL]}
        gpu_ms = acc["ms_total"] / a.steps
        Wk = W_OF(a.k)
        pmc_keys = {"k_count_fast": "k_count_fast<%d, %d, " % (Wk, {1: 4096}.get(Wk, 2048)), "k_compact_wave": "k_compact_wave<",
                    "k_scan<emit>": "k_scan_fast<%d, 2" % Wk if a.k <= 127 else "k_scan<%d, 2" % Wk,
                    "k_scan<hist>": "k_scan_fast<%d, 0" % Wk if a.k <= 127 else "k_scan<%d, 0" % Wk,
                    GLUE_LABEL: ["k_join_bucket", "k_walk_", "k_rank8", "k_unitig_heads", "k_emit"]}   # (k_rank8 / k_unitig_heads / k_emit: only when the walk handed over to the ranking)
        # every stage kernel with its own roofline numbers; the PMC table is quoted only while it was taken from THESE kernels
        # (its header carries the content hash of the sources libcdbg.so was built from)
        cur_hash = lib_srchash()
        kernels = []
        stamp = None
        for name, tot_ms in ms.items():
            if tot_ms <= 0:
                continue
            t_ms = tot_ms / a.steps
            if name == "k_scan<hist>" and tot_ms < 0.25 * ms.get("k_scan<emit>", ms.get(PAIR, 0.0)):
                # single-pass record layout: the histogram launch scans a 1/64 SAMPLE of the tiles to size the partition regions -- not a pass over A1
                kernels.append({"kernel": name, "avg_launch_ms": t_ms, "note": "sampled histogram (1 tile in 64) that sizes the partition regions; not a full pass: no roofline figure"})
                continue
            ach = per_kernel[name] / (t_ms * 1e-3) / 1e9
            if name == PAIR:
                # (all launches of the three kernels in one step: the table's per-launch averages x launches / the 2 profiled steps)
                traffic, stamp_k = pmc_traffic([pmc_keys["k_scan<emit>"], "k_place<", pmc_keys["k_count_fast"]], a.cfg)
                fresh = bool(stamp_k and cur_hash and ("srchash=%s" % cur_hash) in stamp_k) and not a.skewed and traffic is not None
                stamp = stamp or stamp_k
                kernels.append({"kernel": name, "alg_bytes_per_launch": per_kernel[name], "avg_launch_ms": t_ms, "achieved": ach, "frac": ach / HBM_PEAK_GBS,
                                "traffic": traffic if fresh else None,
                                "parts_ms": {"k_scan<emit> (alone on the chip)": acc["ms_scan_emit"] / a.steps,
                                             "count stage, wall (k_count_fast once per slice; waits for the placement inside)": acc["ms_count"] / a.steps,
                                             "k_place, busy span of the second stream (overlapped with the count stage: not a summand)": acc["ms_place"] / a.steps},
                                "slices": st["count_slices"], "deferred_records": st["n_deferred_records"]})
                continue
            traffic, stamp_k = pmc_traffic(pmc_keys[name], a.cfg)
            stamp = stamp or stamp_k
            fresh = bool(stamp_k and cur_hash and ("srchash=%s" % cur_hash) in stamp_k) and not a.skewed
            kernels.append({"kernel": name, "alg_bytes_per_launch": per_kernel[name], "avg_launch_ms": t_ms, "achieved": ach,
                            "frac": ach / HBM_PEAK_GBS, "traffic": traffic if fresh else None})
        # headline: of the kernels within 5 % of the longest one (scan and count are 0.2 ms apart at config 3 and swap places from
        # run to run) the one FURTHEST from its roofline
        rated = [x for x in kernels if "frac" in x]
        longest = max(x["avg_launch_ms"] for x in rated)
        head = min((x for x in rated if x["avg_launch_ms"] >= 0.95 * longest), key=lambda x: x["frac"])
        fresh_any = any(x["traffic"] is not None for x in rated)
        traffic_src = ("profiles/%s (separate rocprofv3 --pmc passes of this bench at this source hash; %s)" % (os.path.basename(pmc_csv(a.cfg)), stamp)
                       if fresh_any else "none quoted: profiles/%s was taken from other kernel sources (%s; loaded library: srchash=%s) -- re-run bench_micro/pmc_bench.sh"
                       % (os.path.basename(pmc_csv(a.cfg)), stamp or "absent", cur_hash))
        out = {
            "metric": "distinct k-mers/s reads->unitigs k=%d" % a.k,
            "value": total_distinct * a.steps / dt,
            "unit": "kmers/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if sharded else "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": "%s: synthetic %d x %d bp reads %s, k=%d, abundance-min %d, 1%% substitutions, 30x coverage%s"
                                   % (CFG["name"], a.reads, a.read_len, "in total (one graph)" if sharded else "per GPU", a.k, a.abundance_min,
                                      " -- HOSTILE genome (--skewed: low-complexity blocks, 1000 x 5 kbp repeat, homopolymer runs, 20x coverage skew): a robustness line, not the metric" if a.skewed else ""),
                       "timing_boundary": "reads resident in HBM (ASCII) -> unitigs + KC resident in HBM; includes the stages' host syncs",
                       "multi_gpu": ("single GPU" if world == 1 else
                                     "sharded: minimizer partitions p mod N; reads replicated (2-4 GPUs) or split with an all-to-all-v of super-k-mer records (8 GPUs); glue sharded by owner -- junction records, joined pairs, ranking queries and pieces each travel once over RCCL all-to-all-v (one graph; set_digest comparable with the N=1 line)"
                                     if sharded else "independent read sets per rank (no collective)"),
                       "exchange": xinfo,
                       "minimizer_size": st["minimizer_size"], "log2_partitions": st["log2_partitions"]},
            "counts": {x: st[x] for x in ("n_occurrences", "n_distinct", "n_solid", "n_pieces", "n_unitigs", "n_records", "n_big_partitions", "n_multipass_partitions", "n_cycles", "n_walked_unitigs")},
            "checks": checks, "checks_passed": all(checks.values()),
            "digest": {"set_digest": "%016x" % dig_last["set_digest"], "kc_sum": dig_last["kc_sum"], "kmers_in_unitigs": dig_last["kmers_in_unitigs"]},
            "verify": verify_info,
            "stage_ms": {x: acc[x] / a.steps for x in acc},
            "roofline": {"bound": "hbm", "kernel": head["kernel"], "achieved": head["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": head["frac"],
                         "traffic": head["traffic"], "traffic_source": traffic_src,
                         "alg_bytes_per_launch": head["alg_bytes_per_launch"], "avg_launch_ms": head["avg_launch_ms"],
                         "headline_rule": "of the stage kernels within 5 % of the longest: the lowest fraction",
                         "kernels": kernels,
                         "pipeline": {"alg_bytes": alg_total, "gpu_ms": gpu_ms,
                                      "achieved": alg_total / (gpu_ms * 1e-3) / 1e9,
                                      "frac": alg_total / (gpu_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}},
        }
    g.close(); g.release_cached()                            # (close() hands the buffers to the process's pool; release_cached() gives the HBM back to the driver: the
                                                             #  end-to-end leg is a child process and cannot drain this process's pool)
    if rank == 0:
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cb = cpu_baseline(a.k, a.abundance_min, a.read_len, gen_cfg, a.cpu_sample_reads)
            if "unitig_sets_equal" in cb:
                out["checks"]["reference binary and GPU agree on the baseline sample (canonical unitig sets)"] = cb["unitig_sets_equal"]
            if "set_digest" in cb:
                # the same sample through the GPU path: CPU restatement and HIP kernels must agree on the whole unitig set
                g2 = bcalm_amd.Graph(a.k, a.abundance_min, lib=lib, device_id=local_rank)
                g2.generate_reads(a.cpu_sample_reads, a.read_len, gen_cfg)
                g2.run()
                d2 = g2.digest(); s2 = g2.stats(); g2.close(); g2.release_cached()
                cb["gpu_same_sample"] = {"set_digest": "%016x" % d2["set_digest"], "distinct": s2["n_distinct"], "unitigs": s2["n_unitigs"], "gpu_ms": s2["ms_total"]}
                cb["cpu_gpu_sets_equal"] = cb["set_digest"] == cb["gpu_same_sample"]["set_digest"] and cb["distinct"] == s2["n_distinct"] and cb["unitigs"] == s2["n_unitigs"]
                out["checks"]["cpu restatement and GPU agree on the baseline sample (set digest)"] = cb["cpu_gpu_sets_equal"]
        if world == 1 and not a.no_e2e and not a.skewed:
            # T_e2e (file -> file, reported separately, PCIe / host bound): the bcalm CLI on a FASTA dump of the same generator
            try:
                out["e2e"] = e2e = cli_end_to_end(lib, a.k, a.abundance_min, a.read_len, gen_cfg, a.e2e_reads, local_rank)
                out["t_e2e_s"] = e2e["wall_s"]
                out["checks"]["CLI end to end: exit 0 and a unitig file"] = e2e["ok"]
            except Exception as ex:                          # (no scratch space, no CLI binary: the line still stands)
                out["e2e"] = {"error": repr(ex)}; out["t_e2e_s"] = None
        out["checks_passed"] = all(out["checks"].values())
        json_out.write(json.dumps(out) + "\n"); json_out.flush()
        if not out["checks_passed"]:
            sys.stderr.write("bench.py: OUTPUT CHECK FAILED: %r\n" % out["checks"])
            sys.exit(3)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
