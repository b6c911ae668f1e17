#!/usr/bin/env python3
"""Index.run() end to end at human scale on ONE GPU: G genomes of M Mb in C contigs at divergence d (default
8 x 3000 Mb, 24 contigs, d = 0.1 %: the shape of BASELINE config 5 at a divergence whose table fits 288 GB).
The FASTA files are synthesised on the GPU and written to a temporary directory first (not timed).
   python tools/e2e_human.py [--genomes 8 --mb 3000 --contigs 24 --d 0.001]"""
import argparse, os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from panagram_amd import index as pidx

ap = argparse.ArgumentParser()
ap.add_argument("--genomes", type=int, default=8); ap.add_argument("--mb", type=float, default=3000.0)
ap.add_argument("--contigs", type=int, default=24); ap.add_argument("--d", type=float, default=0.001)
ap.add_argument("--k", type=int, default=21); ap.add_argument("--write-only", default=None, help=argparse.SUPPRESS); ap.add_argument("--profile", action="store_true", help="host-side profile of Index.run()")
a = ap.parse_args()
G, L = a.genomes, int(a.mb * 1e6)
lens = [L // a.contigs] * a.contigs


def write_inputs(d):
    """synthesised on the GPU (torch) in a process of its own, so that the allocator state of the timed process
    is that of a fresh run (freeing tens of GB of torch tensors is paid by the next big hipMalloc)"""
    import torch
    dev = torch.device("cuda", 0)
    genomes = bench.synth_genomes_device(G, lens, a.d, 1234, dev)
    rows = ["name\tfasta"]
    nl = torch.full((1,), 10, dtype=torch.uint8, device=dev)
    for g in range(G):
        fa = os.path.join(d, f"g{g}.fa")
        with open(fa, "wb") as f:
            for c, t in enumerate(genomes[g]):
                f.write(f">chr{c + 1} synthetic\n".encode())
                n80 = (t.numel() // 80) * 80
                body = torch.cat([t[:n80].view(-1, 80), nl.expand(n80 // 80, 1)], dim=1).flatten()
                f.write(body.cpu().numpy().tobytes())
                if n80 < t.numel():
                    f.write(t[n80:].cpu().numpy().tobytes() + b"\n")
        genomes[g] = None
        rows.append(f"g{g}\t{fa}")
    open(os.path.join(d, "samples.tsv"), "w").write("\n".join(rows) + "\n")


if a.write_only:
    write_inputs(a.write_only)
    sys.exit(0)
with tempfile.TemporaryDirectory() as d:
    t0 = time.perf_counter()
    subprocess.run([sys.executable, os.path.abspath(__file__), "--write-only", d, "--genomes", str(G), "--mb", str(a.mb),
                    "--contigs", str(a.contigs), "--d", str(a.d)], check=True)
    print(f"inputs written in {time.perf_counter() - t0:.0f} s", flush=True)
    npos = G * sum(x - a.k + 1 for x in lens)
    t0 = time.perf_counter()
    idx = pidx.Index(os.path.join(d, "samples.tsv"), prefix=os.path.join(d, "idx"), k=a.k, cores=32)
    if a.profile:
        import cProfile, pstats
        pr = cProfile.Profile(); pr.enable(); idx.run(); pr.disable()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
    else:
        idx.run()
    dt = time.perf_counter() - t0
    # sanity at full size: every anchor holds all of its own k-mers; the .gzi geometry matches the payload length
    import numpy as np, pandas as pd
    per = sum(x - a.k + 1 for x in lens)
    for g in (0, G - 1):
        adir = os.path.join(d, "idx", "anchor", f"g{g}")
        tp = pd.read_csv(os.path.join(adir, "total_paircounts.csv"), index_col="name")
        assert int(tp.loc[f"g{g}", "count"]) == per, (g, int(tp.loc[f"g{g}", "count"]), per)
        assert (tp["count"] <= per).all() and (tp["count"] > 0).all()
        gzi = np.fromfile(os.path.join(adir, "bitmap.1.gzi"), "<u8")
        assert int(gzi[0]) == (per * ((G + 7) // 8) + 65279) // 65280 - 1
    out = sum(os.path.getsize(os.path.join(r, f)) for r, _, fs in os.walk(os.path.join(d, "idx")) for f in fs)
    print(f"Index.run(): {G} x {a.mb:g} Mb FASTA files -> table -> {G} anchors -> BGZF/.gzi/TSV files: {dt:.1f} s = "
          f"{npos / dt / 1e6:.0f} M k-mers/s end to end; {npos} positions, {out / 1e9:.1f} GB written")
