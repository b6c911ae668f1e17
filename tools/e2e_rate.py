#!/usr/bin/env python3
"""PCIe / file-inclusive rates (never bench.py's `value`): host ASCII in, host rows out, and the
full Index.run() (FASTA on disk -> BGZF files on disk).   python tools/e2e_rate.py [--mb 100]"""
import argparse, os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _synth as po
from panagram_amd import engine, index as pidx

ap = argparse.ArgumentParser(); ap.add_argument("--mb", type=float, default=100.0); ap.add_argument("--genomes", type=int, default=8); ap.add_argument("--only-index", action="store_true", help="skip the host-buffer legs (big shapes)")
a = ap.parse_args()
L, G, k = int(a.mb * 1e6), a.genomes, 21
gen = po.synth_genomes(G, [L // 5] * 5, 0.01, 1234)
genomes = [[po.codes_to_ascii(c) for c in g] for g in gen]
npos = sum(len(c) - k + 1 for g in genomes for c in g)
if not a.only_index:
    ctx = engine.Context(0)
    tbl = engine.PanTable(ctx, k, G, expected_keys=int(L * 2.5))
    t0 = time.perf_counter()
    for g in range(G):
        ss = engine.SeqSet.from_host(ctx, genomes[g]); tbl.insert_seqset(g, ss); ss.close()
    print(f"table build from host ASCII (H2D + pack + insert), {G} x {a.mb:g} Mb: {time.perf_counter()-t0:.2f} s")
    tbl.rehash(2.0)
    t0 = time.perf_counter(); npos = 0
    for g in range(G):
        ss = engine.SeqSet.from_host(ctx, genomes[g])
        res = engine.AnchorResult(tbl, ss, colsums=True); res.run()
        for c in range(5):
            rows, rows100, bins, info = res.download(c); npos += len(rows)
        res.colsums(); res.close(); ss.close()
    dt = time.perf_counter() - t0
    print(f"host-buffer anchoring (pageable H2D, pack, probe, epilogue, D2H): {npos/dt/1e9:.2f} G k-mers/s ({dt:.2f} s for {npos} positions)")
    tbl.close(); ctx.close()  # the Index below builds its own table: this one must not share the HBM with it
with tempfile.TemporaryDirectory() as d:
    rows = ["name\tfasta"]
    for g in range(G):
        fa = os.path.join(d, f"g{g}.fa")
        open(fa, "wb").write(po.fasta_text([f"chr{c+1}" for c in range(5)], genomes[g]))
        rows.append(f"g{g}\t{fa}")
    open(os.path.join(d, "samples.tsv"), "w").write("\n".join(rows) + "\n")
    t0 = time.perf_counter()
    idx = pidx.Index(os.path.join(d, "samples.tsv"), prefix=os.path.join(d, "idx"), k=k, cores=32)
    idx.run()
    dt = time.perf_counter() - t0
    print(f"Index.run(): FASTA files -> table -> {G} anchors -> BGZF/.gzi/TSV files: {dt:.1f} s = {npos/dt/1e6:.0f} M k-mers/s end to end ({engine.usable_cpus()} usable host cores)")
