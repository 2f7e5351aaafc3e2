"""dev tool: the sifting tier's 'many_members' test case (tests/test_gpu_parity.py::test_count_sift_tier_gpu), repeated: which tier takes the partition"""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bcalm_amd
lib = bcalm_amd.load()
def case(k, noise_n=55, copies=16):
    rng = random.Random(k * 7 + len("many_members"))
    rnd = lambda n: "".join(rng.choice("ACGT") for _ in range(n))
    g = rnd(300 + k)
    reads = [g, g, g[5:], g[::-1].translate(str.maketrans("ACGT", "TGCA"))] + [rnd(k + 99) for _ in range(noise_n)]
    reads += [g] * copies
    reads.append(g[:k + 10] + rnd(1) + g[k + 11:2 * k + 30])
    return ("\n".join(reads) + "\n").encode()
for k in (64, 127, 160):
    for noise_n in (55, 45):
        res = []
        for rep in range(8):
            g = bcalm_amd.Graph(k, 2, lib=lib, log2_partitions=0)
            g.push_text(case(k, noise_n)); g.run(); st = g.stats(); g.close()
            res.append((st["n_multipass_partitions"], st["n_distinct"], st["n_solid"], st["n_records"]))
        print(k, noise_n, os.environ.get("CDBG_EXACT_NO_CUR32"), res, flush=True)
