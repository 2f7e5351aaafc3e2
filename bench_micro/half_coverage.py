"""dev experiment: stage times when every partition holds half the records (15x instead of 30x coverage of the same genome):
what k_count_fast would cost if duplicate super-k-mer records were merged before counting"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bcalm_amd
lib = bcalm_amd.load()
for n, tot in [(100_000_000, 100_000_000), (50_000_000, 100_000_000)]:
    g = bcalm_amd.Graph(31, 2, lib=lib, log2_partitions=22, minimizer_size=16)
    g.generate_reads(n, 150, 3, first_read=0, total_reads=tot)
    for rep in range(2):
        g.run(); st = g.stats(); g.reset()
    print(json.dumps({"reads": n, "genome_for": tot, **{x: (round(st[x], 2) if isinstance(st[x], float) else st[x]) for x in ("n_records", "n_distinct", "n_solid", "n_multipass_partitions", "ms_scan_emit", "ms_count", "ms_compact", "ms_glue")}}))
    g.close()
