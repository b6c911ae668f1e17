#!/usr/bin/env python3
"""GPU BGZF writer (k_row_deflate) by row width: rows of one anchor genome resident in HBM -> bitmap.1.gz on disk.
   python tools/deflate_rate.py [N ...]      (DR_MB: genome length, default 40)"""
import os, sys, time, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import bench
from panagram_amd import engine
MB = float(os.environ.get("DR_MB", "40"))
dev = torch.device("cuda", 0)
ctx = engine.Context(0)
for N in [int(x) for x in sys.argv[1:]] or [8, 16, 27, 64, 128]:
    L = int(MB * 1e6)
    lens = [L // 5] * 5
    genomes = bench.synth_genomes_device(N, lens, float(os.environ.get("DR_D", "0.01")), 1234, dev)
    tbl = engine.PanTable(ctx, 21, N, expected_keys=int(L * (1 + (N - 1) * 0.19) * 1.05))
    first = None
    for g in range(N):
        ss = engine.SeqSet(ctx, lens)
        for c, t in enumerate(genomes[g]):
            ss.load_dev(c, t.data_ptr(), t.numel())
        tbl.insert_seqset(g, ss)
        if g == 0:
            first = ss
        else:
            ss.close()
    res = engine.AnchorResult(tbl, first, colsums=True)
    res.run()
    ctx.synchronize()
    nbytes = (N + 7) // 8
    rows = first.total_kmers(21) * nbytes
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "x.gz")
        best = 1e9
        for rep in range(4):
            t0 = time.perf_counter()
            res.write_bgzf(1, p, p + "i", level=-2, threads=16)
            best = min(best, time.perf_counter() - t0)
        out = os.path.getsize(p)
    print(f"N={N:4d} rows of {nbytes:2d} B: {rows/1e9:.2f} GB in {best*1e3:.1f} ms = {rows/best/1e9:.1f} GB/s, ratio {rows/out:.1f}", flush=True)
    res.close(); first.close(); tbl.close()
    del genomes
    torch.cuda.empty_cache()
