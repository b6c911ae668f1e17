"""where Index.load_inputs' time goes (FASTA files of tools/e2e_fresh.py --write-only DIR): python tools/attic/r6_load_inputs_breakdown.py DIR"""
import os, sys, time
from concurrent.futures import ThreadPoolExecutor
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import numpy as np
from panagram_amd import engine
d = sys.argv[1]
files = sorted(f for f in os.listdir(d) if f.endswith(".fa"))
paths = [os.path.join(d, f) for f in files]
tot = sum(os.path.getsize(p) for p in paths)
for nthreads in (1, 6, 12):
    t0 = time.perf_counter()
    with ThreadPoolExecutor(nthreads) as ex:
        imgs = list(ex.map(lambda p: np.fromfile(p, dtype=np.uint8), paths[:24]))
    dt = time.perf_counter() - t0
    n = sum(i.nbytes for i in imgs)
    print(f"np.fromfile, {nthreads} threads: {n / dt / 1e9:.1f} GB/s ({dt:.2f} s for {n / 1e9:.1f} GB)", flush=True)
    del imgs
ctx = engine.Context(0)
imgs = [np.fromfile(p, dtype=np.uint8) for p in paths[:16]]
for rep in range(2):
    t0 = time.perf_counter()
    sets = [engine.SeqSet.from_fasta(ctx, im) for im in imgs]
    ctx.synchronize()
    dt = time.perf_counter() - t0
    n = sum(i.nbytes for i in imgs)
    print(f"SeqSet.from_fasta of images in (pageable) memory: {n / dt / 1e9:.1f} GB/s ({dt * 1e3 / len(imgs):.1f} ms per file of {imgs[0].nbytes / 1e6:.0f} MB)", flush=True)
    for s in sets:
        s.close()
sk = engine.KmerSketch(ctx, 31)
sets = [engine.SeqSet.from_fasta(ctx, im) for im in imgs[:8]]
ctx.synchronize()
t0 = time.perf_counter()
for s in sets:
    sk.reset(); sk.add(s); sk.registers()
dt = time.perf_counter() - t0
print(f"sketch + registers: {dt * 1e3 / len(sets):.1f} ms per genome")
