#!/usr/bin/env python3
"""k_insert_tile's found path alone: a genome already in the table inserted again (every key found, its bit present:
no store, no queue), and the same genome as a SECOND genome bit (every key found, one mask store each)."""
import sys
import time

import torch

sys.path.insert(0, ".")
import bench
from panagram_amd import engine

dev = torch.device("cuda", 0)
ctx = engine.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
L = [20_000_000] * 5
g = bench.synth_genomes_device(1, L, 0.01, 1234, dev)[0]
ss = engine.SeqSet(ctx, L)
for c, t in enumerate(g):
    ss.load_dev(c, t.data_ptr(), t.numel())
ctx.synchronize()
tbl = engine.PanTable(ctx, 21, 8, expected_keys=230_000_000)


def timed(label, gi):
    ctx.synchronize()
    t = time.perf_counter()
    tbl.insert_seqset(gi, ss)
    ctx.synchronize()
    print(f"{label}: {1e3 * (time.perf_counter() - t):.2f} ms")


timed("genome bit 0, all keys new", 0)
timed("genome bit 0 again (found, bit present)", 0)
timed("genome bit 0 again", 0)
timed("genome bit 1 (found, one mask store per key)", 1)
timed("genome bit 1 again", 1)
