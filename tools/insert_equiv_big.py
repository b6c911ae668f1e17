#!/usr/bin/env python3
"""Config 2's table (8 x 100 Mb, 2.3 x 10^8 keys) built by k_insert_tile and by the one-thread-per-k-mer kernel: the
exported (key, mask) sets must be equal (sorted and compared on the GPU)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import bench
from panagram_amd import engine

dev = torch.device("cuda", 0)
ctx = engine.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
G, L = 8, [20_000_000] * 5
genomes = bench.synth_genomes_device(G, L, 0.01, 1234, dev)
sets = []
for g in range(G):
    ss = engine.SeqSet(ctx, L)
    for c, t in enumerate(genomes[g]):
        ss.load_dev(c, t.data_ptr(), t.numel())
    sets.append(ss)
ctx.synchronize()


def build():
    tbl = engine.PanTable(ctx, 21, G, expected_keys=235_000_000)
    for g in range(G):
        tbl.insert_seqset(g, sets[g])
    keys, vals = tbl.export(0)
    nk = tbl.stats()["nkeys"]
    tbl.close()
    ctx.trim()
    k = torch.from_numpy(keys.view(np.int64)).to(dev)
    v = torch.from_numpy(vals.view(np.int32)).to(dev)
    k, o = torch.sort(k)
    return k, v[o], nk


os.environ.pop("PG_INSERT_PER_THREAD", None)
ka, va, na = build()
os.environ["PG_INSERT_PER_THREAD"] = "1"
kb, vb, nb = build()
assert na == nb == ka.numel() == kb.numel(), (na, nb, ka.numel(), kb.numel())
assert bool((ka[1:] != ka[:-1]).all()), "a key was exported twice"
assert torch.equal(ka, kb) and torch.equal(va, vb), "the two builds differ"
print(f"{na} keys: the wave-cooperative and the per-thread build agree")
