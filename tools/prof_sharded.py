#!/usr/bin/env python3
"""host-side profile of Index.run() in the genome-sharded mode on one GPU (passes):  python tools/prof_sharded.py [blocks]"""
import cProfile, os, pstats, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _synth as po
from panagram_amd import index as pidx
blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 8
G, L, k = 8, 100_000_000, 21
gen = po.synth_genomes(G, [L // 5] * 5, 0.01, 1234)
with tempfile.TemporaryDirectory() as d:
    rows = ["name\tfasta"]
    for g in range(G):
        fa = os.path.join(d, f"g{g}.fa")
        open(fa, "wb").write(po.fasta_text([f"chr{c+1}" for c in range(5)], [po.codes_to_ascii(c) for c in gen[g]]))
        rows.append(f"g{g}\t{fa}")
    open(os.path.join(d, "samples.tsv"), "w").write("\n".join(rows) + "\n")
    idx = pidx.Index(os.path.join(d, "samples.tsv"), prefix=os.path.join(d, "idx"), k=k, shard="genome", genome_blocks=blocks)
    pr = cProfile.Profile(); pr.enable(); idx.run(); pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
