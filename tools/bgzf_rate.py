#!/usr/bin/env python3
"""BGZF writer throughput on real bitmap rows (host side only): threads x level."""
import os, sys, time, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _synth as po
from panagram_amd import engine
L, G, k = int(os.environ.get("BR_L", 50_000_000)), int(os.environ.get("BR_G", 8)), 21
gen = po.synth_genomes(G, [L], 0.01, 1234)
genomes = [[po.codes_to_ascii(c) for c in g] for g in gen]
ctx = engine.Context(0)
tbl = engine.PanTable(ctx, k, G, expected_keys=int(L * 2.5))
for g in range(G):
    ss = engine.SeqSet.from_host(ctx, genomes[g]); tbl.insert_seqset(g, ss); ss.close()
ss = engine.SeqSet.from_host(ctx, genomes[0])
res = engine.AnchorResult(tbl, ss); res.run()
rows = res.download(0)[0]
print("rows", rows.shape, "host cores", os.cpu_count())
with tempfile.TemporaryDirectory() as d:
    for level in (6, 4, 1):
        for th in (16, 32, 64, 128):
            p = os.path.join(d, "x.gz")
            t0 = time.perf_counter()
            w = engine.BgzfWriter(p, level=level, threads=th); w.write(rows); w.close(p + "i")
            dt = time.perf_counter() - t0
            print(f"level {level} threads {th:3d}: {rows.nbytes/dt/1e6:7.0f} MB/s  ratio {rows.nbytes/os.path.getsize(p):.2f}")
    t0 = time.perf_counter(); res.write_bgzf(1, p, p + "i", level=6, threads=64); dt = time.perf_counter() - t0
    print(f"write_bgzf from HBM level 6 threads 64: {rows.nbytes/dt/1e6:.0f} MB/s")
    for lvl, nm in ((-2, "GPU deflate"), (6, "host, 16 threads")):
        t0 = time.perf_counter(); res.write_bgzf(1, p, p + "i", level=lvl, threads=16); dt = time.perf_counter() - t0
        print(f"write_bgzf from HBM, {nm}: {rows.nbytes/dt/1e6:.0f} MB/s  ratio {rows.nbytes/os.path.getsize(p):.2f}")
