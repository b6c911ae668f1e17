#!/usr/bin/env python3
"""Times k_probe alone (rows-only results), k_epilogue alone (pg_rows_epilogue on finished rows) and
the overlapped product path on one bench shape:  python tools/split_time.py [--genomes G --genome-mb M --k K]"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from panagram_amd import engine

ap = argparse.ArgumentParser()
ap.add_argument("--genomes", type=int, default=8)
ap.add_argument("--genome-mb", type=float, default=100.0)
ap.add_argument("--contigs", type=int, default=8)
ap.add_argument("--k", type=int, default=21)
ap.add_argument("--d", type=float, default=0.01)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--no-colsums", action="store_true")
ap.add_argument("--cosched", action="store_true", help="one co-scheduled launch over all genomes (the bench default)")
ap.add_argument("--minimizer", type=int, default=-1)
ap.add_argument("--n-blocks", type=int, default=0, help="1 kb blocks of N per contig")
ap.add_argument("--keys-per-bucket", type=float, default=2.0)
a = ap.parse_args()
dev = torch.device("cuda", 0)
ctx = engine.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
L = int(a.genome_mb * 1e6)
lens = [L // a.contigs] * a.contigs
genomes = bench.synth_genomes_device(a.genomes, lens, a.d, 1234, dev)
if a.n_blocks:
    gen = torch.Generator(device=dev); gen.manual_seed(5)
    for g in range(a.genomes):
        for t in genomes[g]:
            for p in torch.randint(0, t.numel() - 1000, (a.n_blocks,), device=dev, generator=gen).tolist():
                t[p:p + 1000] = ord('N')
seqsets = []
for g in range(a.genomes):
    ss = engine.SeqSet(ctx, lens)
    for c, t in enumerate(genomes[g]):
        ss.load_dev(c, t.data_ptr(), t.numel())
    seqsets.append(ss)
tbl = engine.PanTable(ctx, a.k, a.genomes, expected_keys=int(L * (1 + (a.genomes - 1) * (1 - (1 - a.d) ** a.k)) * 1.05))
if a.minimizer >= 0:
    tbl.set_minimizer(a.minimizer)
for g in range(a.genomes):
    tbl.insert_seqset(g, seqsets[g])
tbl.rehash(a.keys_per_bucket)
npos = sum(s.total_kmers(a.k) for s in seqsets)


def timed(fn):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / a.reps


if a.cosched:
    import numpy as np
    merged = engine.SeqSet.concat(ctx, seqsets)
    grp = np.repeat(np.arange(a.genomes), a.contigs)
    seqsets = [merged]


def mk(**kw):
    rs = [engine.AnchorResult(tbl, s, colsums=not a.no_colsums, **kw) for s in seqsets]
    if a.cosched:
        rs[0].coschedule(grp, 64)
    return rs


full = mk()
t_full = timed(lambda: [r.run() for r in full])
for r in full:
    r.close()
rows = mk(rows_only=True)
t_probe = timed(lambda: [r.run() for r in rows])
t_epi = timed(lambda: [r.rows_epilogue() for r in rows])
print(f"m={tbl.minimizer} positions/step={npos}  full {t_full*1e3:.2f} ms ({npos/t_full/1e9:.1f} G/s)  "
      f"probe only {t_probe*1e3:.2f} ms ({npos/t_probe/1e9:.1f} G/s)  epilogue only {t_epi*1e3:.2f} ms ({npos/t_epi/1e9:.1f} G/s)")
