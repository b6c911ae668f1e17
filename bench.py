#!/usr/bin/env python3
"""bench.py — anchored k-mers/s building the pan-kmer bitmap on MI355X.

One "step" = one pass of the anchor hot path over one batch of synthetic input:
every k-mer position of the anchor genomes is looked up in the GPU-resident
pan-kmer table and its presence row / 1-in-100 row / bin histogram / column sums are written
(device-resident inputs and outputs; BASELINE.json configs[1]: 8 synthetic 100 Mb
genomes, k=21, one GPU, all tables resident).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Multi-GPU (one process per GPU):
  --mode contig-sharded (default)  the path shards by anchor contig with a replicated table and NO data-path
        collective.  Weak scaling: ONE pangenome whose genomes are N times longer (N x 5 contigs of
        20 Mb each), its contig groups dealt to the ranks — rank r anchors contigs 5r..5r+4 of every genome against
        its replica of the table of the WHOLE pangenome.  No rank repeats another rank's work.
  --mode genome-sharded            the table is cut into genome blocks, rank r holds block r; every rank probes every
        anchor position; the blocks' bit columns are all-gathered over RCCL/xGMI and merged on the anchor's writer
        (panagram_amd.distributed.ShardedAnchoring — the product's pipeline), all inside the timed region.
  With N > 1 the default mode also times the genome-sharded pipeline once, untimed-region-outside, and reports it
  as config.genome_sharded_leg (collective bytes included).
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

HBM_PEAK = 8.0e12      # B/s, MI355X_MICROARCH.md
SIMDS = 256 * 4        # 256 CUs x 4 SIMDs
CLOCK_HZ = 2.4e9       # max shader clock, MI355X_MICROARCH.md
VALU_CYCLES = 4.0      # issue cycles of a wave64 VALU instruction of k_probe's mix (tools/valu_rate.hip: 4.2-4.6 measured)


LEG_TIMEOUT_S = 240  # watchdog of the further legs when there is more than one rank (they take seconds)


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


def cpu_baseline(dbs, samples, k, ngenomes, check_rows=None):
    """Time the oracle's C restatement of the reference CPU algorithm (prefix LUT + binary
    search over sorted records + byte scatter + histogram; oracle/anchor_oracle.c) on a
    bounded sample: thread t anchors samples[t] — the reference's only parallel axis is one thread per
    anchor FASTA (cpp/anchor.cpp:217-223).  ``dbs``: [(sorted keys, masks)] per 32-genome group."""
    from oracle import coracle
    odbs = [coracle.OracleDB.from_arrays(kk, mm, k) for kk, mm in dbs]
    nthreads = len(samples)
    results = [None] * nthreads

    def work(t):
        results[t] = coracle.write_bits(odbs, ngenomes, samples[t], k)

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
    for o in odbs:
        o.close()
    return npos / dt, dt, npos, ok


def sorted_db_from_table(tbl, db_idx):
    keys, masks = tbl.export(db_idx)
    kt = torch.from_numpy(keys.view(np.int64)).cuda()
    ks, order = torch.sort(kt)
    ms = torch.from_numpy(masks.view(np.int32)).cuda()[order]
    out = ks.cpu().numpy().view(np.uint64), ms.cpu().numpy().view(np.uint32)
    del kt, ks, ms, order
    return out


def torch_canonical_kmers(ascii_t, k):
    """canonical k-mer values (int64 holding the 2k-bit integer, first base most significant) of an ACGT-only
    uint8 ASCII tensor — plain torch integer ops, independent of the HIP kernels (bench check only)"""
    c = ((ascii_t >> 1) & 3).long()
    c = c ^ (c >> 1)  # A0 C1 G2 T3
    n = c.numel() - k + 1
    fwd = torch.zeros(n, dtype=torch.int64, device=c.device)
    rev = torch.zeros(n, dtype=torch.int64, device=c.device)
    for j in range(k):
        w = c[j:j + n]
        fwd = (fwd << 2) | w
        rev = rev | ((3 - w) << (2 * j))
    # k <= 31 here: values are below 2^62, signed comparison is the unsigned one
    return torch.minimum(fwd, rev)


def sample_db_by_brute_force(genomes, samples, k, ngenomes):
    """The k-mer DB restricted to the canonical k-mers of ``samples`` (ASCII tensors), built WITHOUT the HIP
    library: every genome's k-mers (torch) are searched in the sorted sample keys; a hit sets the genome's bit.
    Returns [(sorted keys u64, masks u32)] per 32-genome group — what the CPU oracle needs to anchor the samples."""
    keys = torch.unique(torch.cat([torch_canonical_kmers(s, k) for s in samples]))  # sorted
    ndbs = (ngenomes + 31) // 32
    masks = [torch.zeros(keys.numel(), dtype=torch.int64, device=keys.device) for _ in range(ndbs)]
    for g in range(ngenomes):
        bit = 1 << (g % 32)
        m = masks[g // 32]
        for contig in genomes[g]:
            for s0 in range(0, contig.numel() - k + 1, 1 << 24):  # 16 M k-mers at a time
                kk = torch_canonical_kmers(contig[s0:s0 + (1 << 24) + k - 1], k)
                idx = torch.searchsorted(keys, kk).clamp_(max=keys.numel() - 1)
                sel = idx[keys[idx] == kk]
                m[sel] = m[sel] | bit
                del kk, idx, sel
    out = []
    for d in range(ndbs):
        keep = masks[d] != 0
        out.append((keys[keep].cpu().numpy().view(np.uint64), masks[d][keep].cpu().numpy().astype(np.uint32)))
    return out


class Pangenome:
    """synthetic pangenome resident in HBM: packed sequences of this rank's contig group, the table of ALL groups"""

    def __init__(self, ctx, dev, G, contig_lens, d, seed, k, groups=1, my_group=0, keep_ascii=True, minimizer=-1,
                 rehash_kpb=None, block=None):
        from panagram_amd import engine
        self.G, self.k, self.contig_lens = G, k, list(contig_lens)
        L = sum(contig_lens)
        novel = 1.0 - (1.0 - d) ** k
        g_lo, g_hi = (0, G) if block is None else block  # genome block of the table (genome-sharded mode)
        # a rank's table holds the k-mers of ITS contig group (what its anchoring can ask for); the other groups' sequences
        # only set their bits in it (pg_table_update_seqset) — Index.build_table does the same with the genomes a rank anchors
        self.filtered = groups > 1 and os.environ.get("PG_FULL_TABLE", "") in ("", "0")
        est = int(L * (1 + max(0, g_hi - g_lo - 1) * novel) * 1.05) * (1 if self.filtered else groups)
        t0 = time.perf_counter()
        self.table = engine.PanTable(ctx, k, g_hi - g_lo, expected_keys=est)
        if minimizer >= 0:
            self.table.set_minimizer(minimizer)
        self.seqsets, self.ascii = None, None
        self.build_s = 0.0
        for j in ([my_group] + [x for x in range(groups) if x != my_group]):
            genomes = synth_genomes_device(G, contig_lens, d, seed + 7919 * j, dev)
            torch.cuda.synchronize()
            seqsets = []
            for g in range(G):
                ss = engine.SeqSet(ctx, contig_lens)
                for c, t in enumerate(genomes[g]):
                    ss.load_dev(c, t.data_ptr(), t.numel())
                seqsets.append(ss)
            ctx.synchronize()
            tb = time.perf_counter()
            for g in range(g_lo, g_hi):
                if self.filtered and j != my_group:
                    self.table.update_seqset(g - g_lo, seqsets[g])
                else:
                    self.table.insert_seqset(g - g_lo, seqsets[g])
            ctx.synchronize()
            self.build_s += time.perf_counter() - tb
            if j == my_group:
                self.seqsets, self.ascii = seqsets, (genomes if keep_ascii else None)
            else:
                for ss in seqsets:
                    ss.close()
            del genomes
            torch.cuda.empty_cache()
        if rehash_kpb:
            self.table.rehash(rehash_kpb)
        ctx.synchronize()
        self.setup_s = time.perf_counter() - t0
        self.stats = self.table.stats()
        self.pos_per_genome = [self.seqsets[g].total_kmers(k) for g in range(G)]

    def close(self):
        for ss in self.seqsets or []:
            ss.close()
        self.table.close()
        self.ascii = None
        torch.cuda.empty_cache()


def timed_steps(run_step, steps, warmup, world, dev, dist):
    for _ in range(warmup):
        run_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        run_step()
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


def make_results(ctx, pg, colsums, per_genome, piece_tiles):
    """per_genome: one result (= one k_probe launch) per anchor genome, the statistics pass of
    genome g overlapping the probes of genome g+1.  Default: ONE result over all G genomes,
    tiles co-scheduled so that homologous regions share their table lines in L2
    (pg_result_coschedule) — the reference anchors its FASTAs in parallel threads too
    (cpp/anchor.cpp:217-223)."""
    from panagram_amd import engine
    if per_genome:
        return [engine.AnchorResult(pg.table, pg.seqsets[g], colsums=colsums) for g in range(pg.G)], None
    merged = engine.SeqSet.concat(ctx, pg.seqsets)
    r = engine.AnchorResult(pg.table, merged, colsums=colsums)
    r.coschedule(np.repeat(np.arange(pg.G), len(pg.contig_lens)), piece_tiles)
    return [r], merged


def load_counters(pos_per_launch, k, G):
    """PMC counters cannot be read from inside this process: the figures come from the committed
    rocprofv3 --pmc passes of this same command (profiles/traffic.json, corrected as MI355X_MICROARCH.md
    prescribes) and are only quoted when the workload is the one that was profiled"""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            tj = json.load(f)
        if abs(tj["positions_per_launch"] - pos_per_launch) < 1 and k == 21 and G == 8:
            return tj
    except (OSError, KeyError, ValueError):
        pass
    return None


def roofline_block(pos_per_launch, avg_launch_s, avg_epi_s, G, value_per_gpu, counters, nruns):
    """The dominant kernel (k_probe) against the two ceilings that can bind it.
    contract: algorithmic bytes = one 64-byte bucket fetch per table probe + sequence + row;
    the design needs far fewer bytes (minimizer-keyed lines serve runs of positions, co-scheduled genomes share
    them in L2), so that figure can exceed the peak and is NOT the bound; what binds is instruction issue:
    frac = VALU wave-instructions x issue cycles / (SIMDs x clock x launch time)."""
    nbytes = (G + 7) // 8
    P = (G + 63) // 64  # table probes per position in this design (one wide-mask sub-table per 64 genomes)
    B = 0.25 + 64.0 * P + 1.01 * nbytes
    contract_bytes = pos_per_launch * B
    out = {
        "kernel": "k_probe",
        "avg_launch_ms": avg_launch_s * 1e3, "launches_averaged": nruns,
        "epilogue_kernel_ms": avg_epi_s * 1e3,
        "algorithmic_bytes_per_position": B, "algorithmic_bytes_per_launch": contract_bytes,
        "contract_hbm_GBps": contract_bytes / avg_launch_s / 1e9,
        "contract_hbm_frac": contract_bytes / avg_launch_s / HBM_PEAK,
        "contract_whole_run_frac": value_per_gpu * B / HBM_PEAK,
        "traffic": None, "hbm_counter_frac": None, "valu_frac": None,
    }
    if counters is not None:
        if counters.get("hbm_bytes_per_launch"):
            traffic = counters["hbm_bytes_per_launch"]
            out["traffic"] = traffic
            out["hbm_counter_frac"] = traffic / avg_launch_s / HBM_PEAK
        if counters.get("SQ_INSTS_VALU"):
            out["valu_wave_instructions_per_launch"] = counters["SQ_INSTS_VALU"]
            out["valu_wave_instructions_per_position"] = counters["SQ_INSTS_VALU"] / pos_per_launch
            out["valu_frac"] = counters["SQ_INSTS_VALU"] * VALU_CYCLES / (SIMDS * CLOCK_HZ * avg_launch_s)
    if out["valu_frac"] is not None and out["valu_frac"] >= (out["hbm_counter_frac"] or 0):
        out.update(bound="valu", achieved=out["valu_wave_instructions_per_launch"] * VALU_CYCLES / avg_launch_s / 1e12,
                   peak=SIMDS * CLOCK_HZ / 1e12, unit="T issue-cycles/s", frac=out["valu_frac"])
    elif out["hbm_counter_frac"] is not None:
        out.update(bound="hbm", achieved=out["traffic"] / avg_launch_s / 1e9, peak=HBM_PEAK / 1e9, unit="GB/s",
                   frac=out["hbm_counter_frac"])
    else:  # no committed counter pass for this workload: no fraction is claimed (contract_* above is not a bound)
        out.update(bound="unmeasured (no committed --pmc pass for this workload)", achieved=None, peak=None, unit=None, frac=None)
    out["note"] = ("frac is the binding ceiling: VALU issue (SQ_INSTS_VALU of the committed rocprofv3 --pmc pass x 4 issue "
                   "cycles / (1024 SIMDs x 2.4 GHz x launch time)) or counter-measured HBM bytes / 8 TB/s; contract_* price one "
                   "64-byte table fetch per position (SURVEY 8d) — the kernel moves far fewer bytes (traffic), so that "
                   "figure is not a bound (DESIGN.md section 4)")
    return out


def north_star_leg(ctx, dev, args):
    """The north star's target shape on ONE GPU — 64 synthetic 200 Mb genomes, k=21, all 64 anchored — as a second
    measured leg: value, launch times, and rows checked against the CPU oracle on a sample whose k-mer DB is built
    by brute force with torch (no HIP kernel involved in the expected rows)."""
    from panagram_amd import engine
    G, k, C, L = 64, 21, 10, 200_000_000
    contig_lens = [L // C] * C
    pg = Pangenome(ctx, dev, G, contig_lens, 0.01, args.seed + 1, k, keep_ascii=True)
    sample_n = 1_000_000
    picks = [0, 37]
    samples = [pg.ascii[g][0][:sample_n] for g in picks]
    t0 = time.perf_counter()
    dbs = sample_db_by_brute_force(pg.ascii, samples, k, G)
    db_s = time.perf_counter() - t0
    samples_host = [s.cpu().numpy() for s in samples]
    pg.ascii = None
    torch.cuda.empty_cache()
    results, merged = make_results(ctx, pg, True, False, 0)
    steps, warmup = 3, 1

    def step():
        for r in results:
            r.run()
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    results[0].timing_reset()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    p_ms, e_ms, nruns = results[0].timing_mean()
    pos = sum(pg.pos_per_genome)
    cs = results[0].contig_colsums(0, C).sum(axis=0)
    assert int(cs[0]) == pg.pos_per_genome[0], "anchor genome 0 must contain every one of its k-mers"
    v, cdt, npos, ok = cpu_baseline(dbs, samples_host, k, G,
                                    lambda t, n: results[0].download(picks[t] * C, want_bitmap100=False)[0][:n])
    out = {
        "workload": "64 synthetic 200 Mb genomes (10 contigs each), k=21, d=0.01, all 64 anchored per step, one GPU "
                    "(the north star's target shape; BASELINE.json's target is 1e9 k-mers/s on 8 GPUs)",
        "value": pos * steps / dt, "unit": "k-mers/s", "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * dt / steps,
        "positions_per_step": pos, "k_probe_ms": p_ms, "k_epilogue_ms": e_ms, "launches_averaged": nruns,
        "table_keys": pg.stats["nkeys"], "table_bytes": pg.stats["bytes"], "table_build_s": pg.build_s,
        "rows_equal_gpu": ok,
        "rows_check": f"first {sample_n} positions of genomes {picks}: CPU oracle rows (k-mer DB of the sample built by "
                      f"brute force with torch in {db_s:.1f} s, {sum(len(kk) for kk, _ in dbs)} keys) == GPU rows of the timed result",
    }
    for r in results:
        r.close()
    if merged is not None:
        merged.close()
    pg.close()
    ctx.trim()
    return out


def sharded_leg(ctx, dev, args, rank, world, dist, steps, warmup, nblocks=None):
    """The genome-sharded pipeline (panagram_amd.distributed.ShardedAnchoring — what Index.run() uses when the
    table exceeds one GPU) on the configs[1] pangenome: rank r holds the table of genome block r only, every rank
    probes all anchor positions; extract + all-gather (RCCL over xGMI) + merge + statistics are all timed."""
    from panagram_amd import distributed as pdist
    from panagram_amd import engine
    G, k = args.genomes, args.k
    L = int(args.genome_mb * 1e6)
    contig_lens = [L // args.contigs] * args.contigs
    nblocks = min(G, nblocks or world)
    per = (G + nblocks - 1) // nblocks
    nblocks = (G + per - 1) // per
    emulated = world == 1 and nblocks > 1  # one GPU plays rank 0 of `nblocks`: its block's table, no collective
    if nblocks > world and not emulated:
        raise SystemExit("bench.py times one pass: --blocks must not exceed the number of GPUs")
    blk = (rank * per, min(G, (rank + 1) * per)) if rank < nblocks else None
    pg = Pangenome(ctx, dev, G, contig_lens, args.d, args.seed, k, keep_ascii=False,
                   block=blk if blk is not None else (0, 1))
    names = [f"g{g}" for g in range(G)]
    seqs = dict(zip(names, pg.seqsets))
    writer = {a: i % world for i, a in enumerate(names)}
    sh = pdist.ShardedAnchoring(engine, ctx, k, G, per, rank, world, seqs, writer, None, None)
    table = pg.table if blk is not None else None

    def step():
        sh.run_pass(table, 0, 1 if emulated else nblocks, False, lambda a, res: res.rows_epilogue())
    dt = timed_steps(step, steps, warmup, world, dev, dist)
    pos = sum(pg.pos_per_genome)
    mine = [a for a in names if writer[a] == rank]
    if mine and not emulated:  # the writer's completed rows: the anchor holds all of its own k-mers
        cs = sh.full[mine[0]].colsums()
        assert int(cs[names.index(mine[0])]) == pg.pos_per_genome[names.index(mine[0])]
    out = {
        "value": pos * steps / dt, "unit": "k-mers/s", "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * dt / steps,
        "positions_per_step": pos, "genome_blocks": nblocks, "genomes_per_block": per,
        "block_table_keys": pg.stats["nkeys"], "block_table_bytes": pg.stats["bytes"],
        "chunk_groups_per_step": len(sh.groups),
        "collective": "all_gather_into_tensor of bit columns (RCCL over xGMI)" if world > 1 else "none (one rank)",
        "collective_bytes_received_per_rank_per_step": sh.bytes_received / max(1, steps + warmup),
        "parallelism": (f"EMULATED rank 0 of {nblocks}: the table of genome block 0 ({per} genome(s)) only, every position probed, "
                        f"columns extracted and merged, no collective" if emulated else
                        f"genome-sharded x{world}: {nblocks} genome blocks of {per}, every rank probes every position, "
                        f"columns all-gathered, anchors' rows merged + statistics on their writer rank"),
        "columns_from_the_probe": bool(getattr(sh, "_direct", False)),
    }
    sh.close()
    pg.close()
    ctx.trim()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--mode", choices=["contig-sharded", "genome-sharded"], default="contig-sharded")
    ap.add_argument("--blocks", type=int, default=0, help="genome blocks of the genome-sharded mode (default: one per GPU)")
    ap.add_argument("--genomes", type=int, default=8)
    ap.add_argument("--genome-mb", type=float, default=100.0)
    ap.add_argument("--contigs", type=int, default=5)
    ap.add_argument("--k", type=int, default=21)
    ap.add_argument("--d", type=float, default=0.01)
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--keys-per-bucket", type=float, default=0.0,
                    help="re-hash the built table to this many keys per 128-byte line (tuning experiments; default 0: keep the "
                         "table as the library builds it from its expected key count, as Index.run() does — in insertion "
                         "order the keys that most genomes share sit in their minimizer's home line, a re-hash scatters them: "
                         "167 vs 155 G k-mers/s)")
    ap.add_argument("--minimizer", type=int, default=-1, help="pin the table's minimizer length (tuning; default: library's choice)")
    ap.add_argument("--no-colsums", action="store_true")
    ap.add_argument("--no-rehash", action="store_true",  # (the default now; kept for older command lines)
                    help="keep the table as created from the expected key count (a re-hash holds the table twice in HBM)")
    ap.add_argument("--per-genome-launches", action="store_true",
                    help="one launch per anchor genome instead of one co-scheduled launch over all of them")
    ap.add_argument("--piece-tiles", type=int, default=0, help="co-scheduling granularity in 512-position tiles (0: library default)")
    ap.add_argument("--no-compare", action="store_true", help="skip the untimed one-launch-per-genome comparison run")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-shapes", action="store_true", help="skip the north-star-shape leg (64 x 200 Mb, k=21)")
    ap.add_argument("--no-sharded-leg", action="store_true", help="skip the genome-sharded pipeline leg")
    ap.add_argument("--cpu-sample-mb", type=float, default=20.0, help="bases per thread of the CPU baseline leg (about 12 s of CPU work)")
    ap.add_argument("--emulate-rank", type=str, default="", metavar="R/N",
                    help="one process plays rank R of an N-rank contig-sharded run (no collective): the N x longer "
                         "pangenome, the replicated table and rank R's contig group, for checking the multi-GPU "
                         "set-up on a one-GPU box; the reported value is this rank's alone")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: no GPU visible (there is no CPU fallback)")
    # (test knobs, for checking the N > 1 code path on a ONE-GPU box: all ranks on device 0 and a gloo process group —
    # RCCL refuses two ranks on one device.  The numbers of such a run mean nothing.)
    one_device = os.environ.get("PG_BENCH_ONE_DEVICE", "") not in ("", "0")
    backend = os.environ.get("PG_BENCH_BACKEND", "nccl")
    if one_device:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from panagram_amd import engine
    ctx = engine.Context(local)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    G, k, C = args.genomes, args.k, args.contigs
    L = int(args.genome_mb * 1e6)
    contig_lens = [L // C] * C
    groups, my_group = world, rank
    if args.emulate_rank and world == 1:
        my_group, groups = (int(x) for x in args.emulate_rank.split("/"))
    default_shape = (G, round(args.genome_mb), k, C, args.d) == (8, 100, 21, 5, 0.01)

    if args.mode == "genome-sharded":
        leg = sharded_leg(ctx, dev, args, rank, world, dist, args.steps, args.warmup, args.blocks)
        out = {"metric": "anchored k-mers/sec building pan-kmer bitmap", "value": leg["value"], "unit": "k-mers/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": leg["ms_per_step"],
               "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
               "config": dict(leg, workload=f"{G} synthetic {args.genome_mb:g} Mb genomes ({C} contigs each), k={k}, d={args.d}, "
                                            f"all {G} anchored per step, genome-sharded over {world} GPU(s) "
                                            "(BASELINE.json configs[4]'s mode on configs[1]'s pangenome)"),
               "roofline": None, "cpu_baseline": None}
        if rank == 0:
            print(json.dumps(out))
        if world > 1:
            dist.destroy_process_group()
        return

    # ---- k-mer set construction on the GPU (replaces kmc + kmc_tools; timed separately) ----
    keep_ascii = rank == 0 and world == 1 and not args.no_cpu_baseline
    pg = Pangenome(ctx, dev, G, contig_lens, args.d, args.seed, k, groups=groups, my_group=my_group, keep_ascii=keep_ascii,
                   minimizer=args.minimizer, rehash_kpb=None if (args.no_rehash or groups > 1 or args.keys_per_bucket <= 0) else args.keys_per_bucket)
    st = pg.stats
    pg_rehashed = not (args.no_rehash or groups > 1 or args.keys_per_bucket <= 0)
    pos_per_step = sum(pg.pos_per_genome)

    results, merged = make_results(ctx, pg, not args.no_colsums, args.per_genome_launches, args.piece_tiles)

    def step():
        for r in results:
            r.run()
    # per-launch kernel durations come from HIP events recorded by the library on the stream the
    # kernels run on, one event set per run: the mean over ALL timed steps' launches (pg_result_timing_mean)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    for r in results:
        r.timing_reset()
    elapsed = timed_steps(step, args.steps, 0, world, dev, dist)
    tm = [r.timing_mean() for r in results]
    avg_launch_s = float(np.mean([t[0] for t in tm])) / 1e3       # dominant kernel: k_probe
    avg_epi_s = float(np.mean([t[1] for t in tm])) / 1e3
    nruns = int(sum(t[2] for t in tm))
    pos_per_launch = pos_per_step / len(results)

    def genome_rows(g, n):  # first n rows of genome g's first contig
        r, ci = (results[g], 0) if args.per_genome_launches else (results[0], g * C)
        return r.download(ci)[0][:n]

    # ---- invariants at full size (cheap): anchor g contains all of its own k-mers ----
    if not args.no_colsums:
        cs = results[0].colsums() if args.per_genome_launches else results[0].contig_colsums(0, C).sum(axis=0)
        assert int(cs[0]) == pg.pos_per_genome[0], "anchor genome 0 must contain every one of its k-mers"

    value = world * pos_per_step * args.steps / elapsed
    counters = load_counters(pos_per_launch, k, G) if not args.per_genome_launches else None
    if counters is not None and groups > 1:
        # a rank of a multi-GPU run launches the profiled kernel over as many positions, against an N x larger table:
        # the instruction count per position carries over, the HBM traffic of the one-GPU profile does not
        counters = {"SQ_INSTS_VALU": counters.get("SQ_INSTS_VALU"), "profiled_on": "one GPU (profiles/traffic.json)"}
    shape = (G, round(args.genome_mb), k)
    baseline_config = {(8, 100, 21): "BASELINE.json configs[1]", (27, 135, 21): "BASELINE.json configs[2] at full size",
                       (64, 200, 31): "BASELINE.json configs[3] at full size, all 64 genomes anchored",
                       (8, 3000, 21): "the shape of BASELINE.json configs[4] on ONE GPU, at a divergence whose table fits"
                       }.get(shape, "not a BASELINE.json config")
    if groups > 1 and world == 1:
        tbl_txt = ("the table of the k-mers of this rank's contigs (the rest of the pangenome only sets bits in it)"
                   if pg.filtered else "table of all of it")
        workload = (f"EMULATED rank {my_group} of {groups}: {G} synthetic {args.genome_mb * groups:g} Mb genomes, {tbl_txt}, "
                    f"this rank's {C} contigs of every genome anchored")
        parallelism = f"one GPU playing rank {my_group} of a contig-sharded x{groups} run"
    elif world == 1:
        workload = (f"{G} synthetic {args.genome_mb:g} Mb genomes ({C} contigs each), k={k}, d={args.d}, all {G} genomes "
                    f"anchored per step, table resident in one GPU's HBM ({baseline_config})")
        parallelism = "one GPU"
    else:
        tbl_txt = ("every GPU's table built from the contigs it anchors, the rest of the pangenome only setting bits in it"
                   if pg.filtered else "one table of all of it replicated on every GPU")
        workload = (f"{G} synthetic {args.genome_mb * world:g} Mb genomes ({C * world} contigs of {L // C / 1e6:g} Mb each), k={k}, "
                    f"d={args.d}: the configs[1] pangenome made {world}x longer, {tbl_txt}, "
                    f"the {C * world} contig groups dealt to the {world} ranks ({C} contigs of every genome each)")
        parallelism = f"contig-sharded x{world}: disjoint contigs per rank, a table per rank, no data-path collective"
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
            "workload": workload,
            "positions_per_step_per_gpu": pos_per_step,
            "table_keys": st["nkeys"], "table_bytes": st["bytes"], "keys_per_128B_line": round(st["nkeys"] / max(1, st["nbuckets"]), 3),
            "table_rehashed": bool(pg_rehashed),
            "table_build_s": pg.build_s, "table_spill_fraction": pg.table.spill()[0], "table_slots_per_line": pg.table.spill()[1],
            "probes_per_position": (G + 63) // 64, "nbytes": (G + 7) // 8,
            "colsums": not args.no_colsums,
            "launches_per_step": len(results),
            "schedule": "one launch per anchor genome" if args.per_genome_launches else
                        "one launch over all anchor genomes, tiles co-scheduled (homologous regions side by side)",
            "parallelism": parallelism,
        },
        "roofline": roofline_block(pos_per_launch, avg_launch_s, avg_epi_s, G, value / world, counters, nruns),
    }

    if world == 1 and not args.per_genome_launches and not args.no_compare:
        # for comparison only (outside the timed region): the same work as one launch per genome
        alt, _ = make_results(ctx, pg, not args.no_colsums, True, 0)

        def alt_step():
            for r in alt:
                r.run()
        for r in alt:
            r.run()
        torch.cuda.synchronize()
        for r in alt:
            r.timing_reset()
        dt = timed_steps(alt_step, 3, 0, 1, dev, None)
        tma = [r.timing_mean() for r in alt]
        pgl = {"value": pos_per_step * 3 / dt, "k_probe_ms_per_launch": float(np.mean([t[0] for t in tma])),
               "positions_per_launch": pos_per_step / len(alt)}
        if counters is not None and counters.get("per_genome_launches"):
            c2 = counters["per_genome_launches"]
            s2 = pgl["k_probe_ms_per_launch"] / 1e3
            pgl["hbm_counter_frac"] = c2["hbm_bytes_per_launch"] / s2 / HBM_PEAK
            if c2.get("SQ_INSTS_VALU"):
                pgl["valu_frac"] = c2["SQ_INSTS_VALU"] * VALU_CYCLES / (SIMDS * CLOCK_HZ * s2)
            pgl["traffic"] = c2["hbm_bytes_per_launch"]
        out["config"]["per_genome_launches"] = pgl
        out["config"]["per_genome_launches_value"] = pgl["value"]
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

        dbs = [sorted_db_from_table(pg.table, i) for i in range((G + 31) // 32)]
        samples = [pg.ascii[t % G][0][:sample].cpu().numpy() for t in range(nthreads)]
        v, dt, npos, ok = cpu_baseline(dbs, samples, k, G, gpu_rows)
        out["cpu_baseline"] = {
            "value": v, "unit": "k-mers/s", "cores": nthreads, "kind": "port",
            "sample": f"{nthreads} threads x first {args.cpu_sample_mb:g} Mb of a genome each = {npos} positions "
                      f"in {dt:.1f} s against the full {st['nkeys']}-key DB (prefix LUT + binary search, "
                      f"oracle/anchor_oracle.c); host shows {os.cpu_count()} hardware threads, "
                      f"{engine.usable_cpus()} usable under its CPU quota",
            "rows_equal_gpu": ok,
        }
        del dbs, samples
    for r in results:
        r.close()
    if merged is not None:
        merged.close()
    pg.close()
    ctx.trim()

    # ---- further legs, outside the timed region of `value` ----
    # (the line so far goes to stderr first: should a further leg die, the measured value is still on record; stdout
    # carries the ONE complete line at the end)
    if rank == 0:
        print("[bench] timed region done, before the further legs: " + json.dumps(out), file=sys.stderr, flush=True)
    # With more than one rank the further leg has a collective in it, and a rank that dies or stalls there would leave
    # the others waiting inside RCCL for ever — and the measured line unprinted.  A watchdog per rank: if the leg has not
    # come back after LEG_TIMEOUT_S, rank 0 prints the line as it stands (the leg marked as timed out) and every rank
    # leaves with exit code 0.
    watchdog = None
    if world > 1:
        def give_up():
            if rank == 0:
                out["config"].setdefault("genome_sharded_leg", {"error": f"no result after {LEG_TIMEOUT_S} s (watchdog): the measured line stands"})
                print(json.dumps(out), flush=True)
            print(f"[bench] rank {rank}: further leg timed out, leaving", file=sys.stderr, flush=True)
            os._exit(0)
        watchdog = threading.Timer(LEG_TIMEOUT_S, give_up)
        watchdog.daemon = True
        watchdog.start()
    if not args.no_sharded_leg and (world > 1 or default_shape):
        # the one mode with a data-path collective (BASELINE.json configs[4]); with one rank: the pipeline's own cost
        try:
            out["config"]["genome_sharded_leg"] = sharded_leg(ctx, dev, args, rank, world, dist, 3, 1, args.blocks)
        except Exception as e:  # a failed extra leg must not take the measured line with it
            out["config"]["genome_sharded_leg"] = {"error": f"{type(e).__name__}: {e}"}
    if world == 1 and default_shape and not args.no_other_shapes:
        try:
            out["config"]["other_shapes"] = [north_star_leg(ctx, dev, args)]
        except Exception as e:
            out["config"]["other_shapes"] = [{"error": f"{type(e).__name__}: {e}"}]
    if watchdog is not None:
        watchdog.cancel()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        bye = threading.Timer(60.0, lambda: os._exit(0))  # (the line is out: a stuck teardown must not turn into a failure)
        bye.daemon = True
        bye.start()
        dist.destroy_process_group()
        bye.cancel()


if __name__ == "__main__":
    main()
