"""dev tool: the CPU restatement (oracle/cpu_mt.cpp) on 4 M reads with 16 / 32 / 64 / 256 threads: which thread count is the fair baseline under a CPU quota?"""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import oracle_lib
orc = oracle_lib.load()
text = orc.synth_reads(4000000, 150, 3)
for th in (16, 32, 64, 256):
    r = oracle_lib.cpu_mt_run(text, 31, 2, th)
    print(th, "threads: %.2f s  %.1f M distinct k-mers/s" % (r["s_total"], r["distinct"] / r["s_total"] / 1e6), "count %.2f solid %.2f unitigs %.2f" % (r["s_count"], r["s_solid"], r["s_unitigs"]), flush=True)
