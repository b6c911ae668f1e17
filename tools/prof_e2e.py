import cProfile, pstats, os, sys, tempfile, time
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), 'tools'))
import _synth as po
from panagram_amd import engine, index as pidx
import argparse
ap = argparse.ArgumentParser(); ap.add_argument('--genomes', type=int, default=8); ap.add_argument('--mb', type=float, default=100.0); ap.add_argument('--contigs', type=int, default=5)
a = ap.parse_args()
L, G, k = int(a.mb * 1e6), a.genomes, 21
C = a.contigs
gen = po.synth_genomes(G, [L // C] * C, 0.01, 1234)
genomes = [[po.codes_to_ascii(c) for c in g] for g in gen]
with tempfile.TemporaryDirectory() as d:
    rows = ["name\tfasta"]
    for g in range(G):
        fa = os.path.join(d, f"g{g}.fa")
        open(fa, "wb").write(po.fasta_text([f"chr{c+1}" for c in range(C)], genomes[g]))
        rows.append(f"g{g}\t{fa}")
    open(os.path.join(d, "samples.tsv"), "w").write("\n".join(rows) + "\n")
    idx = pidx.Index(os.path.join(d, "samples.tsv"), prefix=os.path.join(d, "idx"), k=k, cores=32)
    pr = cProfile.Profile(); pr.enable()
    t0 = time.perf_counter(); idx.run(); dt = time.perf_counter() - t0
    pr.disable()
    print("Index.run", dt)
    pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
