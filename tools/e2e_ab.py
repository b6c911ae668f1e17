#!/usr/bin/env python3
"""Index.run() end to end under several environment settings on the SAME inputs and box:
   python tools/e2e_ab.py --genomes 27 --mb 135 --variants "" "PG_WRITERS=2" --reps 2"""
import argparse, os, shutil, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _synth as po
from panagram_amd import index as pidx

ap = argparse.ArgumentParser()
ap.add_argument("--mb", type=float, default=100.0); ap.add_argument("--genomes", type=int, default=8)
ap.add_argument("--variants", nargs="*", default=[""]); ap.add_argument("--reps", type=int, default=2)
a = ap.parse_args()
L, G, k = int(a.mb * 1e6), a.genomes, 21
gen = po.synth_genomes(G, [L // 5] * 5, 0.01, 1234)
with tempfile.TemporaryDirectory() as d:
    rows = ["name\tfasta"]
    for g in range(G):
        fa = os.path.join(d, f"g{g}.fa")
        open(fa, "wb").write(po.fasta_text([f"chr{c+1}" for c in range(5)], [po.codes_to_ascii(c) for c in gen[g]]))
        rows.append(f"g{g}\t{fa}")
    gen = None
    open(os.path.join(d, "samples.tsv"), "w").write("\n".join(rows) + "\n")
    for rep in range(a.reps + 1):  # (the first round warms the HIP runtime and the page cache)
        for v in a.variants:
            keys = [kv.split("=")[0] for kv in v.split() if "=" in kv]
            for kv in v.split():
                if "=" in kv:
                    os.environ[kv.split("=")[0]] = kv.split("=", 1)[1]
            out = os.path.join(d, "idx")
            t0 = time.perf_counter()
            pidx.Index(os.path.join(d, "samples.tsv"), prefix=out, k=k, cores=32).run()
            dt = time.perf_counter() - t0
            shutil.rmtree(out)
            for kk in keys:
                os.environ.pop(kk, None)
            print(f"round {rep} [{v}] {dt:.2f} s" + ("  (warm-up)" if rep == 0 else ""), flush=True)
