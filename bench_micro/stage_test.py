import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bcalm_amd
lib = bcalm_amd.load()
n = int(sys.argv[1]); m = int(sys.argv[2]); lnp = int(sys.argv[3]) if len(sys.argv) > 3 else -1
g = bcalm_amd.Graph(31, 2, lib=lib, minimizer_size=m, log2_partitions=lnp)
g.generate_reads(n, 150, 3); print("gen ok", flush=True)
g.count(); st = g.stats(); print("count ok", st["n_records"], st["n_solid"], st["n_solid_travellers"], st["n_big_partitions"], flush=True)
g.compact(); st = g.stats(); print("compact ok", st["n_pieces"], st["n_glue_open_ends"], st["n_big_partitions"], flush=True)
g.glue(); st = g.stats(); print("glue ok", st["n_unitigs"], flush=True)
g.close()
