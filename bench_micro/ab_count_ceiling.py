"""dev tool: what would the count stage cost without the reverse complement / canonical compare of every
member k-mer?  Variants libcdbg_fwd.so (-DCDBG_AB_GEN_FWD: the generator draws every read from the forward strand, so the stored
form of a k-mer is the same in every read) and libcdbg_fwd_norc.so (the same + -DCDBG_AB_NO_RC: the count kernels hash the stored
bits as they are -- the table load is unchanged, the keys are not canonical: count stage only, results not used)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bcalm_amd
SHAPES = {3: (31, 150, 100_000_000), 4: (55, 150, 125_000_000), 5: (127, 1000, 6_250_000)}
for cfg in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "3,4,5").split(",")]:
    k, L, n = SHAPES[cfg]
    for name in ("fwd", "fwd_norc", "fwd", "fwd_norc"):
        lib = bcalm_amd.load(os.path.join(ROOT, "bench_micro", "variants", "libcdbg_%s.so" % name))
        g = bcalm_amd.Graph(k, 2, lib=lib)
        g.generate_reads(n, L, cfg)
        best = None
        for rep in range(4):
            g.count(); st = g.stats(); g.reset()
            if rep and (best is None or st["ms_count"] < best["ms_count"]): best = st
        g.close(); lib.cdbg_release_cached()
        print(json.dumps({"cfg": cfg, "lib": name, "n_distinct": best["n_distinct"], "n_solid": best["n_solid"], "multipass": best["n_multipass_partitions"],
                          **{x: round(best[x], 2) for x in ("ms_scan_emit", "ms_count")}}), flush=True)
