#!/usr/bin/env python3
"""Experiment: does co-scheduling homologous regions of all anchor genomes make the table lines hit
in L2 / Infinity Cache?  Cuts every genome into pieces and anchors them (rows only) in
(contig, piece, genome) order versus genome-major order."""
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
ap.add_argument("--piece", type=int, default=32768)
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
dev = torch.device("cuda", 0)
ctx = engine.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
L = int(a.genome_mb * 1e6)
lens = [L // a.contigs] * a.contigs
genomes = bench.synth_genomes_device(a.genomes, lens, a.d, 1234, dev)
seqsets = []
for g in range(a.genomes):
    ss = engine.SeqSet(ctx, lens)
    for c, t in enumerate(genomes[g]):
        ss.load_dev(c, t.data_ptr(), t.numel())
    seqsets.append(ss)
tbl = engine.PanTable(ctx, a.k, a.genomes, expected_keys=int(L * (1 + (a.genomes - 1) * (1 - (1 - a.d) ** a.k)) * 1.05))
for g in range(a.genomes):
    tbl.insert_seqset(g, seqsets[g])
tbl.rehash(2.0)


def pieces(order):
    out = []
    P = a.piece
    for key in order:
        g, c, j = key
        t = genomes[g][c]
        lo = j * P
        hi = min(t.numel(), lo + P + a.k - 1)
        out.append((t.data_ptr() + lo, hi - lo))
    return out


npieces = (lens[0] - a.k + 1 + a.piece - 1) // a.piece
orders = {
    "genome-major": [(g, c, j) for g in range(a.genomes) for c in range(a.contigs) for j in range(npieces)],
    "co-scheduled": [(g, c, j) for c in range(a.contigs) for j in range(npieces) for g in range(a.genomes)],
}
for name, order in orders.items():
    pcs = pieces(order)
    ss = engine.SeqSet(ctx, [n for _, n in pcs])
    for i, (p, n) in enumerate(pcs):
        ss.load_dev(i, p, n)
    r = engine.AnchorResult(tbl, ss, colsums=False, rows_only=True)
    r.run(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        r.run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.reps
    npos = ss.total_kmers(a.k)
    print(f"{name}: {len(pcs)} pieces, {npos} positions, {dt*1e3:.2f} ms, {npos/dt/1e9:.1f} G k-mers/s")
    r.close(); ss.close()
