#!/usr/bin/env python3
"""Wall time of each pg_table_insert_seqset call of the default bench's build (8 x 100 Mb, k=21), to set beside the
kernel trace's k_insert_tile durations: what the host side of a call costs.  python tools/insert_calls.py [tag]"""
import sys
import time

import torch

sys.path.insert(0, ".")
import bench
from panagram_amd import engine

tag = sys.argv[1] if len(sys.argv) > 1 else ""
dev = torch.device("cuda", 0)
ctx = engine.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
G, L, k, d = 8, [20_000_000] * 5, 21, 0.01
novel = 1.0 - (1.0 - d) ** k
est = int(sum(L) * (1 + (G - 1) * novel) * 1.05)
for rep in range(2):
    genomes = bench.synth_genomes_device(G, L, d, 1234, dev)
    sets = []
    for g in range(G):
        ss = engine.SeqSet(ctx, L)
        for c, t in enumerate(genomes[g]):
            ss.load_dev(c, t.data_ptr(), t.numel())
        sets.append(ss)
    tbl = engine.PanTable(ctx, k, G, expected_keys=est)
    ctx.synchronize()
    torch.cuda.synchronize()
    per = []
    t0 = time.perf_counter()
    for g in range(G):
        t = time.perf_counter()
        tbl.insert_seqset(g, sets[g])
        per.append(1e3 * (time.perf_counter() - t))
    ctx.synchronize()
    total = 1e3 * (time.perf_counter() - t0)
    print(f"[{tag}] rep {rep}: build {total:.2f} ms; per call " + " ".join(f"{x:.2f}" for x in per))
    for ss in sets:
        ss.close()
    tbl.close()
    del genomes
    torch.cuda.empty_cache()
