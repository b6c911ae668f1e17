#!/usr/bin/env python3
"""bench.py — anchored k-mers/s building the pan-kmer bitmap on MI355X.

One "step" = one pass of the anchor hot path over one batch of synthetic input:
every k-mer position of all G anchor genomes is looked up in the GPU-resident
pan-kmer table and its presence row / 1-in-100 row / bin histogram are written
(device-resident inputs and outputs; BASELINE.json configs[1]: 8 synthetic 100 Mb
genomes, k=21, one GPU, all tables resident).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Multi-GPU: the path shards by anchor contig with a replicated table and NO data-path
collective; every rank anchors an equal-sized shard ("scaling": "weak").
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12  # B/s, MI355X_MICROARCH.md


def synth_genomes_device(ngenomes, contig_lens, d, seed, device):
    """SURVEY §8d generator (i.i.d. base genome; genome g>0 = per-base substitution at rate d,
    new base != old), drawn with torch on the GPU so that no PCIe traffic is involved.
    Returns [genome][contig] uint8 ASCII tensors."""
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    base = [torch.randint(0, 4, (L,), dtype=torch.uint8, device=device, generator=gen) for L in contig_lens]
    out = [[acgt[b.long()] for b in base]]
    for g in range(1, ngenomes):
        gg = torch.Generator(device=device)
        gg.manual_seed(seed + g)
        contigs = []
        for b in base:
            mut = torch.rand(b.shape, device=device, generator=gg) < d
            shift = torch.randint(1, 4, b.shape, dtype=torch.uint8, device=device, generator=gg)
            contigs.append(acgt[torch.where(mut, (b + shift) & 3, b).long()])
        out.append(contigs)
    return out


def cpu_baseline(tbl, genomes_dev, k, ngenomes, sample_bases, nthreads, check_rows=None):
    """Time the oracle's C restatement of the reference CPU algorithm (prefix LUT + binary
    search over sorted records + byte scatter + histogram; oracle/anchor_oracle.c) on a
    bounded sample: thread t anchors the first `sample_bases` of genome t — the reference's
    only parallel axis is one thread per anchor FASTA (cpp/anchor.cpp:217-223)."""
    from oracle import coracle
    keys, masks = tbl.export(0)
    kt = torch.from_numpy(keys.view(np.int64)).cuda()
    ks, order = torch.sort(kt)
    ms = torch.from_numpy(masks.view(np.int32)).cuda()[order]
    keys_s = ks.cpu().numpy().view(np.uint64)
    masks_s = ms.cpu().numpy().view(np.uint32)
    del kt, ks, ms, order
    db = coracle.OracleDB.from_arrays(keys_s, masks_s, k)
    samples = [genomes_dev[t % len(genomes_dev)][0][:sample_bases].cpu().numpy() for t in range(nthreads)]
    results = [None] * nthreads

    def work(t):
        results[t] = coracle.write_bits([db], ngenomes, samples[t], k)

    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(t,)) for t in range(nthreads)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    dt = time.perf_counter() - t0
    npos = sum(len(r[0]) for r in results)
    ok = None
    if check_rows is not None:  # full-size parity spot check: CPU sample rows == GPU rows
        ok = all(np.array_equal(results[t][0], check_rows(t, len(results[t][0]))) for t in range(nthreads))
    db.close()
    return npos / dt, dt, npos, ok


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--genomes", type=int, default=8)
    ap.add_argument("--genome-mb", type=float, default=100.0)
    ap.add_argument("--contigs", type=int, default=5)
    ap.add_argument("--k", type=int, default=21)
    ap.add_argument("--d", type=float, default=0.01)
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--keys-per-bucket", type=float, default=2.0)
    ap.add_argument("--minimizer", type=int, default=-1, help="pin the table's minimizer length (tuning; default: library's choice)")
    ap.add_argument("--no-colsums", action="store_true")
    ap.add_argument("--no-rehash", action="store_true",
                    help="keep the table as created from the expected key count (a re-hash holds the table twice in HBM)")
    ap.add_argument("--per-genome-launches", action="store_true",
                    help="one launch per anchor genome instead of one co-scheduled launch over all of them")
    ap.add_argument("--piece-tiles", type=int, default=0, help="co-scheduling granularity in 512-position tiles (0: library default)")
    ap.add_argument("--no-compare", action="store_true", help="skip the untimed one-launch-per-genome comparison run")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-mb", type=float, default=20.0, help="bases per thread of the CPU baseline leg (about 12 s of CPU work)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: no GPU visible (there is no CPU fallback)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from panagram_amd import engine
    ctx = engine.Context(local)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)

    G, k = args.genomes, args.k
    L = int(args.genome_mb * 1e6)
    contig_lens = [L // args.contigs] * args.contigs
    # every rank holds the same pangenome (replicated table) and anchors its own equal shard
    genomes = synth_genomes_device(G, contig_lens, args.d, args.seed, dev)
    torch.cuda.synchronize()

    seqsets = []
    for g in range(G):
        ss = engine.SeqSet(ctx, contig_lens)
        for c, t in enumerate(genomes[g]):
            ss.load_dev(c, t.data_ptr(), t.numel())
        seqsets.append(ss)
    torch.cuda.synchronize()
    if args.no_cpu_baseline and G * L > 4_000_000_000:  # big inputs: the ASCII copies are only needed by the CPU leg
        genomes = None
        torch.cuda.empty_cache()

    # ---- k-mer set construction on the GPU (replaces kmc + kmc_tools; timed separately) ----
    novel = 1.0 - (1.0 - args.d) ** k
    est_keys = int(L * (1 + (G - 1) * novel) * 1.05)
    t0 = time.perf_counter()
    tbl = engine.PanTable(ctx, k, G, expected_keys=est_keys)
    if args.minimizer >= 0:
        tbl.set_minimizer(args.minimizer)
    for g in range(G):
        tbl.insert_seqset(g, seqsets[g])
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t0
    if not args.no_rehash:
        tbl.rehash(args.keys_per_bucket)
    torch.cuda.synchronize()
    st = tbl.stats()

    pos_per_genome = [seqsets[g].total_kmers(k) for g in range(G)]
    pos_per_step = sum(pos_per_genome)
    C = args.contigs

    def make_results(per_genome):
        """per_genome: one result (= one k_probe launch) per anchor genome, the statistics pass of
        genome g overlapping the probes of genome g+1.  Default: ONE result over all G genomes,
        tiles co-scheduled so that homologous regions share their table lines in L2
        (pg_result_coschedule) — the reference anchors its FASTAs in parallel threads too
        (cpp/anchor.cpp:217-223)."""
        if per_genome:
            return [engine.AnchorResult(tbl, seqsets[g], colsums=not args.no_colsums) for g in range(G)], None
        merged = engine.SeqSet.concat(ctx, seqsets)
        r = engine.AnchorResult(tbl, merged, colsums=not args.no_colsums)
        r.coschedule(np.repeat(np.arange(G), C), args.piece_tiles)
        return [r], merged

    def timed(results, steps, warmup):
        for _ in range(warmup):
            for r in results:
                r.run()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            for r in results:
                r.run()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt

    results, merged = make_results(args.per_genome_launches)
    elapsed = timed(results, args.steps, args.warmup)
    # per-launch kernel durations come from HIP events recorded by the library on the stream the
    # kernels run on (pg_result_timing): events of the last timed step, read after the timed region
    probe_ms, epi_ms = zip(*[r.timing() for r in results])
    avg_launch_s = float(np.mean(probe_ms)) / 1e3       # dominant kernel: k_probe
    avg_epi_s = float(np.mean(epi_ms)) / 1e3
    pos_per_launch = pos_per_step / len(results)

    def genome_rows(g, n):  # first n rows of genome g's first contig
        r, ci = (results[g], 0) if args.per_genome_launches else (results[0], g * C)
        return r.download(ci)[0][:n]

    # ---- invariants at full size (cheap): anchor g contains all of its own k-mers ----
    if not args.no_colsums:
        cs = results[0].colsums() if args.per_genome_launches else results[0].contig_colsums(0, C).sum(axis=0)
        assert int(cs[0]) == pos_per_genome[0], "anchor genome 0 must contain every one of its k-mers"

    # HBM traffic of the dominant kernel: PMC counters cannot be read from inside this process;
    # the figure comes from the committed rocprofv3 --pmc passes of this same command
    # (profiles/traffic.json, corrected as MI355X_MICROARCH.md prescribes) and is only quoted
    # when the workload is the one that was profiled
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            tj = json.load(f)
        if abs(tj["positions_per_launch"] - pos_per_launch) < 1 and k == 21 and G == 8:
            traffic = tj["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        pass
    nbytes = (G + 7) // 8
    P = (G + 63) // 64  # table probes per position in this design (one wide-mask sub-table per 64 genomes)
    B = 0.25 + 64.0 * P + 1.01 * nbytes
    per_launch_bytes = pos_per_launch * B
    achieved = per_launch_bytes / avg_launch_s
    value = world * pos_per_step * args.steps / elapsed

    shape = (G, round(args.genome_mb), k)
    baseline_config = {(8, 100, 21): "BASELINE.json configs[1]", (27, 135, 21): "BASELINE.json configs[2] at full size",
                       (64, 200, 31): "BASELINE.json configs[3] at full size, all 64 genomes anchored",
                       (8, 3000, 21): "the shape of BASELINE.json configs[4] on ONE GPU, at a divergence whose table fits"
                       }.get(shape, "not a BASELINE.json config")
    out = {
        "metric": "anchored k-mers/sec building pan-kmer bitmap",
        "value": value,
        "unit": "k-mers/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u64",
        "data": "synthetic",
        "config": {
            "workload": f"{G} synthetic {args.genome_mb:g} Mb genomes ({args.contigs} contigs each), k={k}, "
                        f"d={args.d}, all {G} genomes anchored per step, table resident in one GPU's HBM "
                        f"({baseline_config})",
            "positions_per_step_per_gpu": pos_per_step,
            "table_keys": st["nkeys"], "table_bytes": st["bytes"], "keys_per_128B_line": args.keys_per_bucket,
            "table_build_s": build_s, "table_spill_fraction": tbl.spill()[0], "table_slots_per_line": tbl.spill()[1], "probes_per_position": P, "nbytes": nbytes,
            "colsums": not args.no_colsums,
            "launches_per_step": len(results),
            "schedule": "one launch per anchor genome" if args.per_genome_launches else
                        "one launch over all anchor genomes, tiles co-scheduled (homologous regions side by side)",
            "parallelism": f"contig-sharded x{world}, replicated table, no collective",
        },
        "roofline": {
            "bound": "hbm", "kernel": "k_probe",
            "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
            "frac": achieved / HBM_PEAK,
            "algorithmic_bytes_per_position": B,
            "avg_launch_ms": avg_launch_s * 1e3,
            "epilogue_kernel_ms": avg_epi_s * 1e3,
            "whole_run_frac": (value / world) * B / HBM_PEAK,
            "hbm_read_frac": (pos_per_launch * (0.25 + 64.0 * P) / avg_launch_s) / HBM_PEAK,
            "traffic": traffic,
            "algorithmic_bytes_per_launch": per_launch_bytes,
            "note": "algorithmic bytes price one 64-byte table fetch per position (SURVEY 8d); minimizer-keyed "
                    "lines serve runs of positions and, co-scheduled, all anchor genomes share them in L2, so "
                    "the measured HBM bytes (traffic) are far fewer and frac can exceed 1; past that point "
                    "k_probe is VALU-issue bound (DESIGN.md section 4)",
        },
    }

    if world == 1 and not args.per_genome_launches and not args.no_compare:
        # for comparison only (outside the timed region): the same work as one launch per genome
        alt, _ = make_results(True)
        dt = timed(alt, 3, 1)
        out["config"]["per_genome_launches_value"] = pos_per_step * 3 / dt
        for r in alt:
            r.close()

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        nthreads = min(G, engine.usable_cpus())
        sample = int(args.cpu_sample_mb * 1e6)
        cache = {}

        def gpu_rows(t, n):
            g = t % G
            if g not in cache:
                cache[g] = genome_rows(g, n)
            return cache[g][:n]

        v, dt, npos, ok = cpu_baseline(tbl, genomes, k, G, sample, nthreads, gpu_rows)
        out["cpu_baseline"] = {
            "value": v, "unit": "k-mers/s", "cores": nthreads, "kind": "port",
            "sample": f"{nthreads} threads x first {args.cpu_sample_mb:g} Mb of a genome each = {npos} positions "
                      f"in {dt:.1f} s against the full {st['nkeys']}-key DB (prefix LUT + binary search, "
                      f"oracle/anchor_oracle.c); host shows {os.cpu_count()} hardware threads, "
                      f"{engine.usable_cpus()} usable under its CPU quota",
            "rows_equal_gpu": ok,
        }
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
