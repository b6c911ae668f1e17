#!/usr/bin/env python3
"""zlib strategies on real bitmap rows (single thread, per 65280-byte block as BGZF does)."""
import os, sys, time, zlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _synth as po
from panagram_amd import engine
for G, L in ((8, 20_000_000), (27, 8_000_000), (64, 4_000_000)):
    gen = po.synth_genomes(G, [L], 0.01, 1234)
    genomes = [[po.codes_to_ascii(c) for c in g] for g in gen]
    ctx = engine.Context(0)
    tbl = engine.PanTable(ctx, 21, G, expected_keys=int(L * 2.5))
    for g in range(G):
        ss = engine.SeqSet.from_host(ctx, genomes[g]); tbl.insert_seqset(g, ss); ss.close()
    ss = engine.SeqSet.from_host(ctx, genomes[0])
    res = engine.AnchorResult(tbl, ss); res.run()
    rows = res.download(0)[0].tobytes()[: 40 * 65280]
    print(f"N={G}: {len(rows)/1e6:.1f} MB sample")
    for name, level, strat in (("default-6", 6, zlib.Z_DEFAULT_STRATEGY), ("default-1", 1, zlib.Z_DEFAULT_STRATEGY),
                               ("rle-6", 6, zlib.Z_RLE), ("filtered-6", 6, zlib.Z_FILTERED), ("huffman-only", 6, zlib.Z_HUFFMAN_ONLY),
                               ("fixed-6", 6, zlib.Z_FIXED)):
        t0 = time.perf_counter(); out = 0
        for i in range(0, len(rows), 65280):
            c = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strat)
            out += len(c.compress(rows[i:i + 65280])) + len(c.flush())
        dt = time.perf_counter() - t0
        print(f"   {name:13s} {len(rows)/dt/1e6:7.1f} MB/s  ratio {len(rows)/out:.2f}")
    import tempfile
    nb = (G + 7) // 8
    with tempfile.TemporaryDirectory() as d:
        for name, level in (("writer zlib-6", 6), ("writer rle", 6 | engine.BgzfWriter.RLE), ("writer rows", 6 | engine.BgzfWriter.ROWS(nb))):
            if nb == 1 and "rows" in name:
                continue
            p = os.path.join(d, "x.gz")
            t0 = time.perf_counter()
            w = engine.BgzfWriter(p, level=level, threads=1); w.write(rows); w.close()
            dt = time.perf_counter() - t0
            print(f"   {name:13s} {len(rows)/dt/1e6:7.1f} MB/s  ratio {len(rows)/os.path.getsize(p):.2f}")
    res.close(); ss.close(); tbl.close(); ctx.close()
