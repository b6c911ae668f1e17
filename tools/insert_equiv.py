#!/usr/bin/env python3
"""The wave-cooperative table build (k_insert_tile: distinct slots claimed in one round, later copies retired) against
the one-thread-per-k-mer build (PG_INSERT_PER_THREAD=1) on repeat-rich genomes: exported (key, mask) sets must be equal.
Inputs: identical copies and diverged copies of elements, tandem arrays of several periods, near-identical genomes — equal
new k-mers inside a 64-lane batch, in neighbouring tiles and far apart, built by different waves at the same time."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _synth as po
from panagram_amd import engine

rng = np.random.default_rng(7)
k, G = int(sys.argv[1]) if len(sys.argv) > 1 else 21, 4
elems = [rng.integers(0, 4, n, dtype=np.uint8) for n in (300, 1100, 5000)]
genomes = []
for g in range(G):
    parts = []
    for _ in range(3000):
        kind = rng.integers(0, 4)
        if kind == 0:
            parts.append(elems[int(rng.integers(0, 3))])  # identical copy
        elif kind == 1:
            e = elems[int(rng.integers(0, 3))].copy()
            mut = rng.random(len(e)) < 0.02
            e[mut] = (e[mut] + rng.integers(1, 4, int(mut.sum()), dtype=np.uint8)) % 4
            parts.append(e)
        elif kind == 2:
            period = int(rng.choice([2, 5, 13, 40, 150, 700]))
            parts.append(np.tile(rng.integers(0, 4, period, dtype=np.uint8), int(rng.integers(200, 4000)) // period + 2))
        else:
            parts.append(rng.integers(0, 4, int(rng.integers(100, 3000)), dtype=np.uint8))
    genomes.append([po.codes_to_ascii(np.concatenate(parts))])
print(f"{G} genomes x {len(genomes[0][0]) / 1e6:.1f} Mb, k={k}")
ctx = engine.Context(0)


def build():
    tbl = engine.PanTable(ctx, k, G)
    for g in range(G):
        ss = engine.SeqSet.from_host(ctx, genomes[g])
        tbl.insert_seqset(g, ss)
        ss.close()
    keys, vals = tbl.export(0)
    o = np.argsort(keys, kind="stable")
    st = tbl.stats()
    tbl.close()
    return keys[o], vals[o], st


for rep in range(3):
    os.environ.pop("PG_INSERT_PER_THREAD", None)
    ka, va, sa = build()
    os.environ["PG_INSERT_PER_THREAD"] = "1"
    kb, vb, sb = build()
    assert len(np.unique(ka)) == len(ka), "a key was exported twice"
    assert np.array_equal(ka, kb) and np.array_equal(va, vb), "the two builds differ"
    assert sa["nkeys"] == sb["nkeys"] == len(ka), (sa, sb, len(ka))
    print(f"round {rep}: {len(ka)} keys, equal")
