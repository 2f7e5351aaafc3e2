import sys, os, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bcalm_amd
lib = bcalm_amd.load(os.environ.get("CDBG_LIB") or None)
g = bcalm_amd.Graph(55, 2, lib=lib)
g.generate_reads(125000000, 150, 4)
g.run(); st = g.stats()
print({k: st[k] for k in ("n_records", "n_member_kmers", "n_occurrences", "n_distinct", "n_solid", "n_solid_travellers", "n_pieces", "n_unitigs", "n_multipass_partitions", "n_big_partitions", "n_split_buckets")})
g.close()
