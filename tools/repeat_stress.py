#!/usr/bin/env python3
"""Stress: repeat families (many diverged copies of one element) make huge minimizer groups.
Builds the table, anchors, checks that the anchor holds all of its own k-mers, reports times."""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _synth as po
from panagram_amd import engine

ap = argparse.ArgumentParser()
ap.add_argument("--copies", type=int, default=2000)
ap.add_argument("--elem", type=int, default=3000)
ap.add_argument("--div", type=float, default=0.03)
ap.add_argument("--k", type=int, default=21)
ap.add_argument("--genomes", type=int, default=2)
ap.add_argument("--background-mb", type=float, default=0.0, help="unique sequence (shared by the genomes at 1 %% SNP divergence) appended to every genome")
ap.add_argument("--minimizer", type=int, default=-1, help="pin the table's minimizer length (default: the library's choice)")
a = ap.parse_args()
rng = np.random.default_rng(1)
elem = rng.integers(0, 4, a.elem, dtype=np.uint8)
genomes = []
for g in range(a.genomes):
    parts = []
    for c in range(a.copies):
        e = elem.copy()
        mut = rng.random(a.elem) < a.div
        e[mut] = (e[mut] + rng.integers(1, 4, int(mut.sum()), dtype=np.uint8)) % 4
        parts.append(e)
        parts.append(rng.integers(0, 4, 500, dtype=np.uint8))
    if a.background_mb > 0:
        if g == 0:
            bg = rng.integers(0, 4, int(a.background_mb * 1e6), dtype=np.uint8)
        b = bg.copy()
        mut = rng.random(len(b)) < 0.01
        b[mut] = (b[mut] + rng.integers(1, 4, int(mut.sum()), dtype=np.uint8)) % 4
        parts.append(b)
    genomes.append([po.codes_to_ascii(np.concatenate(parts))])
L = len(genomes[0][0])
print(f"{a.genomes} genomes x {L/1e6:.1f} Mb, {a.copies} copies of a {a.elem} bp element at {a.div:.0%} divergence")
ctx = engine.Context(0)
tbl = engine.PanTable(ctx, a.k, a.genomes)
if a.minimizer >= 0:
    tbl.set_minimizer(a.minimizer)
t0 = time.perf_counter()
for g in range(a.genomes):
    ss = engine.SeqSet.from_host(ctx, genomes[g]); tbl.insert_seqset(g, ss); ss.close()
ctx.synchronize()
print(f"build {time.perf_counter()-t0:.2f} s, stats {tbl.stats()}")
ss = engine.SeqSet.from_host(ctx, genomes[0])
res = engine.AnchorResult(tbl, ss)
res.run(); ctx.synchronize()
t0 = time.perf_counter(); res.run(); ctx.synchronize(); dt = time.perf_counter() - t0
print(f"anchor {L/dt/1e9:.2f} G k-mers/s ({dt*1e3:.1f} ms), m = {tbl.minimizer}")
rows = res.download(0)[0]
assert (rows[:, 0] & 1).all(), "anchor genome must contain all its k-mers"
print("own-bit check ok; rows with both bits:", int(((rows[:, 0] & 3) == 3).sum()))
tbl.rehash(3.0)  # (the spill fraction is counted at a re-hash: keys outside their minimizer's home line / keys)
print(f"after a re-hash to 3 keys per line: spill fraction {tbl.spill()[0]:.3f}, m = {tbl.minimizer}")
