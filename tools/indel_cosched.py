#!/usr/bin/env python3
"""How much of the co-scheduling gain survives indels and structural variants?  Derived genomes get substitutions plus
insertions/deletions (so homologous positions drift apart along the contig), and optionally inversions,
translocations between contigs and a shuffled contig order; anchoring all genomes in one co-scheduled launch is
compared with one launch per genome."""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from panagram_amd import engine

ap = argparse.ArgumentParser()
ap.add_argument("--genomes", type=int, default=8)
ap.add_argument("--contig-mb", type=float, default=10.0)
ap.add_argument("--contigs", type=int, default=5)
ap.add_argument("--d", type=float, default=0.01)
ap.add_argument("--indel-rate", type=float, default=2e-4, help="events per base")
ap.add_argument("--indel-mean", type=float, default=200.0, help="mean event length (geometric), capped at 50 kb")
ap.add_argument("--big-rate", type=float, default=0.0, help="additional rare events of 100-500 kb per base")
ap.add_argument("--inversions", type=int, default=0, help="per derived genome and contig: segments of 0.2-2 Mb reverse-complemented in place")
ap.add_argument("--translocations", type=int, default=0, help="per derived genome: segments of 0.5-3 Mb moved to a random place of another contig")
ap.add_argument("--shuffle-contigs", action="store_true", help="derived genomes list their contigs in a random order")
ap.add_argument("--k", type=int, default=21)
a = ap.parse_args()
rng = np.random.default_rng(1)
L = int(a.contig_mb * 1e6)
ACGT = np.frombuffer(b"ACGT", np.uint8)
base = [rng.integers(0, 4, L, dtype=np.uint8) for _ in range(a.contigs)]


def derive(b, g):
    r = np.random.default_rng(1000 + g)
    x = b.copy()
    mut = r.random(len(x)) < a.d
    x[mut] = (x[mut] + r.integers(1, 4, int(mut.sum()), dtype=np.uint8)) % 4
    n_ev = r.poisson(a.indel_rate * len(x))
    pos = np.sort(r.integers(0, len(x), n_ev))
    lens = np.minimum(r.geometric(1.0 / a.indel_mean, n_ev), 50000)
    if a.big_rate:
        nb = r.poisson(a.big_rate * len(x))
        pos = np.concatenate([pos, r.integers(0, len(x), nb)])
        lens = np.concatenate([lens, r.integers(100000, 500000, nb)])
        o = np.argsort(pos)
        pos, lens = pos[o], lens[o]
    parts, cur = [], 0
    for p, ln in zip(pos, lens):
        if p < cur:
            continue
        parts.append(x[cur:p])
        if r.random() < 0.5:
            cur = min(len(x), p + ln)           # deletion
        else:
            parts.append(r.integers(0, 4, ln, dtype=np.uint8))  # insertion
            cur = p
    parts.append(x[cur:])
    return np.concatenate(parts)


def rearrange(contigs, g):
    """structural variants: inversions (reverse complement in place), translocations (cut and paste between contigs),
    contig order"""
    r = np.random.default_rng(5000 + g)
    contigs = [c.copy() for c in contigs]
    for ci in range(len(contigs)):
        for _ in range(a.inversions):
            ln = int(r.integers(200_000, 2_000_000))
            if ln >= len(contigs[ci]):
                continue
            p = int(r.integers(0, len(contigs[ci]) - ln))
            contigs[ci][p:p + ln] = (3 - contigs[ci][p:p + ln])[::-1]
    for _ in range(a.translocations):
        src, dst = (int(x) for x in r.choice(len(contigs), 2, replace=len(contigs) < 2))
        ln = int(r.integers(500_000, 3_000_000))
        if ln >= len(contigs[src]):
            continue
        p = int(r.integers(0, len(contigs[src]) - ln))
        seg = contigs[src][p:p + ln].copy()
        contigs[src] = np.concatenate([contigs[src][:p], contigs[src][p + ln:]])
        q = int(r.integers(0, len(contigs[dst])))
        contigs[dst] = np.concatenate([contigs[dst][:q], seg, contigs[dst][q:]])
    order = np.arange(len(contigs))
    if a.shuffle_contigs:
        order = r.permutation(len(contigs))
        contigs = [contigs[i] for i in order]
    return contigs, order


rearr = [rearrange([derive(c, g) for c in base], g) for g in range(1, a.genomes)]
orders = [np.arange(a.contigs)] + [o for _, o in rearr]
genomes = [[ACGT[c] for c in base]] + [[ACGT[c] for c in cs] for cs, _ in rearr]
sizes = [sum(len(c) for c in g) for g in genomes]
print(f"{a.genomes} genomes, {a.contigs} contigs, sizes {min(sizes)/1e6:.1f}-{max(sizes)/1e6:.1f} Mb, indel rate {a.indel_rate}, mean {a.indel_mean}, big {a.big_rate}, "
      f"inversions {a.inversions}/contig, translocations {a.translocations}/genome, contig order {'shuffled' if a.shuffle_contigs else 'kept'}")
ctx = engine.Context(0)
tbl = engine.PanTable(ctx, a.k, a.genomes, expected_keys=int(sizes[0] * 2.5))
sets = []
for g in range(a.genomes):
    ss = engine.SeqSet.from_host(ctx, genomes[g])
    tbl.insert_seqset(g, ss)
    sets.append(ss)
tbl.rehash(2.0)
npos = sum(s.total_kmers(a.k) for s in sets)


def timed(results, reps=5):
    for r in results: r.run()
    ctx.synchronize()
    for r in results: r.timing()
    t0 = time.perf_counter()
    for _ in range(reps):
        for r in results: r.run()
    for r in results: r.timing()   # synchronises on each result's last event
    return (time.perf_counter() - t0) / reps


per = [engine.AnchorResult(tbl, s) for s in sets]
t_per = timed(per)
for r in per: r.close()
merged = engine.SeqSet.concat(ctx, sets)
res = engine.AnchorResult(tbl, merged)
res.coschedule(np.repeat(np.arange(a.genomes), [len(g) for g in genomes]))
t_co = timed([res])
# ... and with the contigs matched by name (the rearranged genomes keep each contig's identity: class = base contig)
res.coschedule(np.repeat(np.arange(a.genomes), [len(g) for g in genomes]), contig_class=np.concatenate(orders))
t_cls = timed([res])
print(f"one launch per genome: {npos/t_per/1e9:.1f} G k-mers/s   co-scheduled by contig order: {npos/t_co/1e9:.1f}   "
      f"co-scheduled by contig identity: {npos/t_cls/1e9:.1f} G k-mers/s")
