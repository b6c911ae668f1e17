#!/usr/bin/env python3
"""Index.run() as WORLD ranks one after the other on one GPU (the contig-sharded partition by pieces of homology classes):
seconds per rank, on whole chromosomes or fragmented assemblies.  python tools/e2e_ranks.py --genomes 4 --mb 40 --contigs 4000 --world 2"""
import argparse, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _synth as po
from panagram_amd import index as pidx
ap = argparse.ArgumentParser()
ap.add_argument("--genomes", type=int, default=4); ap.add_argument("--mb", type=float, default=40.0)
ap.add_argument("--contigs", type=int, default=5); ap.add_argument("--world", type=int, default=2); ap.add_argument("--shard", default=None); ap.add_argument("--blocks", type=int, default=0); ap.add_argument("--profile", action="store_true", help="cProfile of the LAST rank of the multi-rank run (it assembles)")
a = ap.parse_args()
L, G, C, k = int(a.mb * 1e6), a.genomes, a.contigs, 21
gen = po.synth_genomes(G, [L // C] * C, 0.01, 1234)
genomes = [[po.codes_to_ascii(c) for c in g] for g in gen]
with tempfile.TemporaryDirectory() as d:
    rows = ["name\tfasta"]
    for g in range(G):
        fa = os.path.join(d, f"g{g}.fa")
        open(fa, "wb").write(po.fasta_text([f"chr{c+1}" for c in range(C)], genomes[g]))
        rows.append(f"g{g}\t{fa}")
    open(os.path.join(d, "samples.tsv"), "w").write("\n".join(rows) + "\n")
    for world in (1, a.world):
        out = os.path.join(d, f"idx{world}")
        for r in range(world):
            t0 = time.perf_counter()
            idx = pidx.Index(os.path.join(d, "samples.tsv"), prefix=out, k=k, rank=r, world=world, **(dict(shard=a.shard, genome_blocks=a.blocks) if a.shard else {}))
            if a.profile and r == world - 1 and (world > 1 or a.world == 1):
                import cProfile, pstats
                pr = cProfile.Profile(); pr.enable(); idx.run(); pr.disable()
                pstats.Stats(pr).sort_stats("cumulative").print_stats(30)
            else:
                idx.run()
            print(f"world {world} rank {r}: {time.perf_counter() - t0:.2f} s", flush=True)
