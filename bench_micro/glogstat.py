import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bcalm_amd
lib = bcalm_amd.load()
for (n, k, L, cfg, amin) in [(100000000, 31, 150, 3, 2), (20000000, 31, 150, 3, 1), (125000000, 55, 150, 4, 2), (6250000, 127, 1000, 5, 2)]:
    g = bcalm_amd.Graph(k, amin, lib=lib)
    g.generate_reads(n, L, cfg); g.run(); st = g.stats(); g.close()
    print(json.dumps({x: st[x] for x in ("n_solid", "n_solid_travellers", "n_pieces", "n_glue_open_ends", "n_glue_joined", "n_unitigs")} | {"k": k, "amin": amin}))
