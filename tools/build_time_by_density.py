#!/usr/bin/env python3
"""Table build time by density and shape: the genomes of a pangenome inserted one after the other into a table created at a given
number of keys per line (0 = the library's 3).   python tools/build_time_by_density.py --genomes 64 --mb 200 --k 31 --d 0.005 --kpl 1.25,3"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from panagram_amd import engine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--genomes", type=int, default=64)
ap.add_argument("--mb", type=float, default=200.0)
ap.add_argument("--contigs", type=int, default=10)
ap.add_argument("--k", type=int, default=31)
ap.add_argument("--d", type=float, default=0.005)
ap.add_argument("--kpl", default="1.25,2,3")
a = ap.parse_args()
dev = torch.device("cuda:0")
ctx = engine.Context(0)
G = a.genomes
lens = [int(a.mb * 1e6) // a.contigs] * a.contigs
genomes = bench.synth_genomes_device(G, lens, a.d, 1238, dev)
seqsets = []
for g in range(G):
    ss = engine.SeqSet(ctx, lens)
    for c, t in enumerate(genomes[g]):
        ss.load_dev(c, t.data_ptr(), t.numel())
    seqsets.append(ss)
ctx.synchronize()
del genomes
torch.cuda.empty_cache()
sk = engine.KmerSketch(ctx, a.k)
for ss in seqsets:
    sk.add(ss)
est = sk.estimate()
est += est // 32
sk.close()
print(f"{G} x {a.mb:g} Mb, k={a.k}, d={a.d}: {est / 1e6:.0f} M keys expected", flush=True)
for kpl in [float(x) for x in a.kpl.split(",")]:
    for rep in range(2):
        tbl = engine.PanTable(ctx, a.k, G, expected_keys=est, coscheduled=G, keys_per_line=kpl)
        ctx.synchronize()
        t0 = time.perf_counter()
        per = []
        for g in range(G):
            t1 = time.perf_counter()
            tbl.insert_seqset(g, seqsets[g])
            ctx.synchronize()
            per.append(time.perf_counter() - t1)
        dt = time.perf_counter() - t0
        st = tbl.stats()
        print(f"  {kpl:g} keys per line (rep {rep}): build {dt:.3f} s; first genome {per[0] * 1e3:.0f} ms, 2nd {per[1] * 1e3:.0f}, last {per[-1] * 1e3:.0f}, max {max(per) * 1e3:.0f} (genome {per.index(max(per))}); "
              f"{st}", flush=True)
        tbl.close()
