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
# (the files -> files child of baseline_config_legs runs WITHOUT torch, as `python -m panagram_amd index` does: with torch in the
# process the library shares torch's bundled HIP runtime, whose large allocations cost 24 ms per GB — the 90 GB table of
# BASELINE configs[3] then takes 2.2 s to create instead of 0.2)
_TORCHLESS_CHILD = any(a.startswith("run:") for a in sys.argv[1:])
if not _TORCHLESS_CHILD:
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
    # PG_BENCH_N_RUNS=n (experiments): n assembly gaps of 500 N per contig and genome, at the genome's own places — real
    # chromosomes carry gaps, and a contig with an N takes k_probe's masked path (an extra LDS read per batch)
    n_runs = int(os.environ.get("PG_BENCH_N_RUNS", "0"))
    if n_runs:
        for g, contigs in enumerate(out):
            gn = torch.Generator(device=device)
            gn.manual_seed(seed + 1000003 * (g + 1))
            for t in contigs:
                if t.numel() > 10000:
                    for p in torch.randint(0, t.numel() - 600, (n_runs,), device=device, generator=gn).tolist():
                        t[p:p + 500] = ord("N")
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
                 rehash_kpb=None, block=None, pieces_of=None, coscheduled=0):
        """``pieces_of=(rank, world)`` (strong scaling): the pangenome is the SAME whatever the GPU count; this rank
        anchors its pieces of homology classes (panagram_amd.distributed.plan_class_pieces — the partition
        Index.run() uses with several ranks) against a table built from those pieces, every genome then only setting
        its bits in it."""
        from panagram_amd import engine
        self.G, self.k, self.contig_lens = G, k, list(contig_lens)
        self.pieces = None
        # (how the table will be probed decides its minimizer window with the key count, as Index.build_table tells it:
        # the anchor genomes of a launch, 1 = one launch per genome)
        self.coscheduled = coscheduled or G
        if pieces_of is not None:
            self._init_pieces(ctx, dev, d, seed, pieces_of, minimizer)
            return
        L = sum(contig_lens)
        novel = 1.0 - (1.0 - d) ** k
        g_lo, g_hi = (0, G) if block is None else block  # genome block of the table (genome-sharded mode)
        # a rank's table holds the k-mers of ITS contig group (what its anchoring can ask for); the other groups' sequences
        # only set their bits in it (pg_table_update_seqset) — Index.build_table does the same with the genomes a rank anchors
        self.filtered = groups > 1 and os.environ.get("PG_FULL_TABLE", "") in ("", "0")
        est = int(L * (1 + max(0, g_hi - g_lo - 1) * novel) * 1.05) * (1 if self.filtered else groups)
        t0 = time.perf_counter()
        # (as Index.build_table creates its table: sparser than the library's 3 keys per line where HBM is plentiful — it has to
        # leave room for this shape's rows, packed sequences and, in the legs that check rows, the ASCII copy)
        pos_est = G * sum(contig_lens)
        self.keys_per_line = engine.PanTable.roomy_density(ctx, k, g_hi - g_lo, est, pos_est * ((G + 7) // 8) * 21 // 20 + pos_est * (2 if keep_ascii else 1))
        self.table = engine.PanTable(ctx, k, g_hi - g_lo, expected_keys=est, coscheduled=self.coscheduled, keys_per_line=self.keys_per_line)
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

    def _init_pieces(self, ctx, dev, d, seed, pieces_of, minimizer):
        from panagram_amd import distributed as pdist
        from panagram_amd import engine
        rank, world = pieces_of
        G, k, contig_lens = self.G, self.k, self.contig_lens
        C = len(contig_lens)
        genomes = synth_genomes_device(G, contig_lens, d, seed, dev)
        torch.cuda.synchronize()
        full = []
        for g in range(G):
            ss = engine.SeqSet(ctx, contig_lens)
            for c, t in enumerate(genomes[g]):
                ss.load_dev(c, t.data_ptr(), t.numel())
            full.append(ss)
        ctx.synchronize()
        del genomes
        torch.cuda.empty_cache()
        contigs = [(g, c, contig_lens[c] - k + 1, c) for g in range(G) for c in range(C)]  # class = chromosome number
        plan = pdist.plan_class_pieces(contigs, world)
        mine = sorted(plan[rank], key=lambda p: (p[4], p[5], p[0]))
        t0 = time.perf_counter()
        # this rank's pieces, cut out of the packed genomes in HBM with k - 1 bases of overlap
        self.pieces = [[p for p in mine if p[0] == g] for g in range(G)]
        self.seqsets = [full[g].slice([(p[1], p[2], p[3] + k - 1) for p in self.pieces[g]]) for g in range(G)]
        sketch = engine.KmerSketch(ctx, k)
        for ss in self.seqsets:
            sketch.add(ss)
        est = sketch.estimate()
        sketch.close()
        expected = est + est // 32 + 1024
        self.keys_per_line = engine.PanTable.roomy_density(ctx, k, G, expected, 0)  # (as Index.build_table: the rows of this rank's pieces are a fraction of the pangenome's)
        self.table = engine.PanTable(ctx, k, G, expected_keys=expected, coscheduled=self.coscheduled, keys_per_line=self.keys_per_line)
        if minimizer >= 0:
            self.table.set_minimizer(minimizer)
        ctx.synchronize()
        tb = time.perf_counter()
        for g in range(G):
            self.table.insert_seqset(g, self.seqsets[g])
        if world > 1:
            for g in range(G):
                self.table.update_seqset(g, full[g])
        ctx.synchronize()
        self.build_s = time.perf_counter() - tb
        self.filtered = world > 1
        for ss in full:
            ss.close()
        self.ascii = None
        self.setup_s = time.perf_counter() - t0
        self.stats = self.table.stats()
        self.pos_per_genome = [self.seqsets[g].total_kmers(k) for g in range(G)]
        self.total_positions = sum(n for _, _, n, _ in contigs)  # of the whole pangenome, all ranks together
        self.plan_loads = [sum(p[3] for p in sh) for sh in plan]
        self.plan_class_pieces = len({(p[4], p[5]) for sh in plan for p in sh})

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
    if pg.pieces is not None:  # a rank's pieces of homology classes: co-scheduled by (class, piece), as run_index_sharded does
        ids = {key: i for i, key in enumerate(sorted({(p[4], p[5]) for ps in pg.pieces for p in ps}))}
        r.coschedule(np.concatenate([np.full(len(ps), g, np.uint32) for g, ps in enumerate(pg.pieces)]), piece_tiles,
                     contig_class=np.array([ids[(p[4], p[5])] for ps in pg.pieces for p in ps], np.uint32))
    else:
        r.coschedule(np.repeat(np.arange(pg.G), len(pg.contig_lens)), piece_tiles)
    return [r], merged


KERNEL_SOURCES = ("pg_anchor.hip", "pg_device.h", "pg_kernels.h", "pg_kernels.hip", "pg_api.hip")  # what k_probe / k_epilogue are compiled from, and what picks the table layout and window they run with (geom_for, k_table_init)


def kernel_sources_sha():
    """sha256 over the kernel sources (first 16 hex digits): profiles/traffic.json records the one its --pmc passes ran on
    (tools/make_traffic.py), so that counters of an older kernel cannot pass for the current one's"""
    import hashlib
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "panagram_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def load_counters(pos_per_launch, k, G):
    """PMC counters cannot be read from inside this process: the figures come from the committed
    rocprofv3 --pmc passes of this same command (profiles/traffic.json, corrected as MI355X_MICROARCH.md
    prescribes) and are only quoted when the workload is the one that was profiled.  The line says which file they came
    from (its sha256) and whether the kernel sources are still the ones that were profiled."""
    import hashlib
    try:
        path = os.path.join(ROOT, "profiles", "traffic.json")
        with open(path, "rb") as f:
            raw = f.read()
        tj = json.loads(raw)
        if abs(tj["positions_per_launch"] - pos_per_launch) < 1 and k == 21 and G == 8:
            tj["file_sha16"] = hashlib.sha256(raw).hexdigest()[:16]
            tj["sources_match"] = tj.get("kernel_sources_sha16") == kernel_sources_sha()
            return tj
    except (OSError, KeyError, ValueError):
        pass
    return None


def roofline_block(pos_per_launch, avg_launch_s, avg_epi_s, G, value_per_gpu, counters, nruns, bound="issue"):
    """The dominant kernel (k_probe) against the HBM roofline (8 TB/s, MI355X_MICROARCH.md).

    ``achieved`` = ALGORITHMIC bytes per launch / the launch's mean duration, the algorithmic bytes being the design's
    useful-payload floor SURVEY §8d defines: B_min = 0.25 (2-bit base) + 12 P (one 8-byte key + one 4-byte mask word per
    table probed; P = 1: ONE table and one probe per position whatever the genome count) + 1.01 nbytes (the bitmap.1
    row + 1/100 bitmap.100 row) bytes per position.  ``frac`` = achieved / peak <= 1 by construction, recomputable
    from DESIGN.md §4 + profiles/.  ``traffic`` = HBM bytes per launch by the PMC counters of the committed rocprofv3
    passes of this same command (profiles/traffic.json: FETCH_SIZE doubled per the guide's gfx950 correction, +
    WRITE_SIZE), quoted only for the workload they were collected on; traffic / algorithmic = re-read overhead.
    Secondary, named for what they are: ``contract_*`` = SURVEY §8d's 64-byte-bucket-per-probe figure (the kernel
    does not move those bytes — minimizer-keyed lines serve runs of positions, co-scheduled genomes share them in L2 — so
    it exceeds 1 and is no bound); ``valu_issue_*`` = the instruction-issue ceiling at the architectural 2 cycles per
    wave64 VALU instruction and at the 4 cycles measured for this kernel's instruction mix (tools/valu_rate.hip)."""
    nbytes = (G + 7) // 8
    P = 1
    B_min = 0.25 + 12.0 * P + 1.01 * nbytes
    algorithmic = pos_per_launch * B_min
    B_contract = 0.25 + 64.0 * ((G + 63) // 64) + 1.01 * nbytes
    out = {
        # what binds this launch: co-scheduled genomes take most table lines from L2 and the kernel runs into VALU instruction
        # issue ("issue"); one launch per genome / diverged genomes run into HBM's random-line rate ("hbm").  achieved / peak /
        # frac are priced against the HBM roofline either way (priced_against), as the contract asks.
        "bound": bound, "priced_against": "hbm", "kernel": "k_probe",
        "achieved": algorithmic / avg_launch_s / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
        "frac": algorithmic / avg_launch_s / HBM_PEAK,
        "traffic": None,
        "avg_launch_ms": avg_launch_s * 1e3, "launches_averaged": nruns, "epilogue_kernel_ms": avg_epi_s * 1e3,
        "algorithmic_bytes_per_position": B_min, "algorithmic_bytes_per_launch": algorithmic,
        "table_probes_per_position": P,
        "hbm_counter_frac": None, "traffic_over_algorithmic": None,
        "contract_bytes_per_position": B_contract, "contract_hbm_frac": pos_per_launch * B_contract / avg_launch_s / HBM_PEAK,
        "valu_issue_frac_2cycle": None, "valu_issue_frac_4cycle": None,
    }
    if counters is not None:
        if counters.get("hbm_bytes_per_launch"):
            out["traffic"] = counters["hbm_bytes_per_launch"]
            out["hbm_counter_frac"] = out["traffic"] / avg_launch_s / HBM_PEAK
            out["traffic_over_algorithmic"] = out["traffic"] / algorithmic
        if counters.get("SQ_INSTS_VALU"):
            out["valu_wave_instructions_per_launch"] = counters["SQ_INSTS_VALU"]
            out["valu_wave_instructions_per_position"] = counters["SQ_INSTS_VALU"] / pos_per_launch
            out["valu_issue_frac_2cycle"] = counters["SQ_INSTS_VALU"] * 2.0 / (SIMDS * CLOCK_HZ * avg_launch_s)
            out["valu_issue_frac_4cycle"] = counters["SQ_INSTS_VALU"] * VALU_CYCLES / (SIMDS * CLOCK_HZ * avg_launch_s)
        out["counters_from"] = counters.get("profiled_on", "profiles/traffic.json (committed rocprofv3 --pmc passes of this command)")
        out["counters_round"] = counters.get("round")
        out["counters_file_sha16"] = counters.get("file_sha16")
        # false: the kernel sources changed after the --pmc passes — traffic and instruction counts are the OLD kernel's
        out["counters_kernel_sources_match"] = counters.get("sources_match")
    out["note"] = ("frac = algorithmic bytes (B_min of SURVEY 8d x positions per launch) / mean k_probe launch time / 8 TB/s; traffic = "
                   "PMC-measured HBM bytes of the same launch (null for workloads without a committed --pmc pass); the kernel is "
                   "not HBM-bound — valu_issue_frac_* say how close instruction issue is to its ceiling (DESIGN.md section 4)")
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
        "launch_note": "a run of this size goes out in 16 chunks, the statistics pass of a chunk beside the next chunk's probe "
                       "(DESIGN.md 7.2): k_probe_ms / k_epilogue_ms are the two streams' spans, not a sum",
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


def baseline_config_leg(ctx, dev, args, label, G, contig_lens, k, d, seed, picks, sample_n, steps=3, warmup=1):
    """One of BASELINE.json's wider configs at FULL size on one GPU, as the headline is measured: every genome anchored per
    step in one co-scheduled result + its statistics, inputs and outputs in HBM.  Carries its own roofline block (B_min of
    SURVEY 8d for its row width, the k_probe time from HIP events on the kernels' stream) and an oracle check: the first
    ``sample_n`` rows of the first contig and the last ``sample_n / 2`` rows of the last contig of the ``picks`` genomes
    against the CPU oracle (cpp/anchor.cpp:112-195 restated in oracle/anchor_oracle.c), its k-mer DB built by brute force
    with torch.  Outside the timed region of ``value``."""
    pg = Pangenome(ctx, dev, G, contig_lens, d, seed, k, keep_ascii=True)
    C = len(contig_lens)
    tail_n = sample_n // 2
    where = [(g, 0, 0) for g in picks] + [(g, C - 1, contig_lens[C - 1] - tail_n) for g in picks]
    samples = [pg.ascii[g][ci][s0:s0 + (sample_n if ci == 0 else tail_n)] for g, ci, s0 in where]
    t0 = time.perf_counter()
    dbs = sample_db_by_brute_force(pg.ascii, samples, k, G)
    db_s = time.perf_counter() - t0
    samples_host = [x.cpu().numpy() for x in samples]
    pg.ascii = None
    del samples
    torch.cuda.empty_cache()
    results, merged = make_results(ctx, pg, True, False, 0)
    for _ in range(warmup):
        results[0].run()
    torch.cuda.synchronize()
    results[0].timing_reset()
    t0 = time.perf_counter()
    for _ in range(steps):
        results[0].run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    p_ms, e_ms, nruns = results[0].timing_mean()
    pos = sum(pg.pos_per_genome)
    ccs = results[0].contig_colsums().astype(np.int64)
    own_ok = all(int(ccs[g * C:(g + 1) * C, g].sum()) == pg.pos_per_genome[g] for g in range(G))

    def gpu_rows(t, n):
        g, ci, s0 = where[t]
        rows = results[0].download(g * C + ci, want_bitmap100=False)[0]
        return rows[s0:s0 + n]
    v, cdt, npos, ok = cpu_baseline(dbs, samples_host, k, G, gpu_rows)
    st = pg.stats
    rf = roofline_block(pos, p_ms / 1e3, e_ms / 1e3, G, pos * steps / dt, None, nruns, bound="issue + hbm (see DESIGN.md section 4)")
    out = {"config": label,
           "workload": f"{G} synthetic {sum(contig_lens) / 1e6:g} Mb genomes ({C} contigs each), k={k}, d={d}, all {G} anchored per step, one GPU",
           "value": pos * steps / dt, "unit": "k-mers/s", "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * dt / steps,
           "positions_per_step": pos, "nbytes": (G + 7) // 8, "k_probe_ms": p_ms, "statistics_ms": e_ms, "launches_averaged": nruns,
           "table_keys": st["nkeys"], "table_bytes": st["bytes"], "minimizer_length": pg.table.minimizer, "table_build_s": pg.build_s,
           "row_bytes_per_step": pos * ((G + 7) // 8),
           "roofline": {kk: rf[kk] for kk in ("bound", "priced_against", "kernel", "achieved", "peak", "unit", "frac", "avg_launch_ms",
                                              "algorithmic_bytes_per_position", "algorithmic_bytes_per_launch")},
           "anchors_hold_all_own_kmers": bool(own_ok), "rows_equal_gpu": ok,
           "rows_check": f"genomes {list(picks)}: first {sample_n} rows of the first contig and last {tail_n} rows of the last contig == CPU "
                         f"oracle rows (k-mer DB of the samples by brute force with torch in {db_s:.1f} s)"}
    if nruns and e_ms > 0.5 * p_ms and (G + 7) // 8 >= 8:
        out["launch_note"] = ("a run of this size goes out in chunks, the statistics of a chunk beside the next chunk's probe: k_probe_ms / "
                              "statistics_ms are the two streams' spans, not a sum")
    for r in results:
        r.close()
    if merged is not None:
        merged.close()
    pg.close()
    ctx.trim()
    torch.cuda.empty_cache()
    return out


def baseline_config_legs(ctx, dev, args):
    """BASELINE.json configs[2] (27 Arabidopsis-scale ~135 Mb genomes, k=21) and configs[3] (64 synthetic 200 Mb genomes,
    k=31) at full size on this GPU, each with its roofline block and oracle-checked head / tail rows; and configs[3] files ->
    files through Index.run() (SURVEY 8d(ii) for the widest config: 102 GB of rows through k_row_deflate and to disk)."""
    import shutil
    import tempfile
    legs = []
    for label, G, lens, k, d, seed, picks, sn in (
            ("BASELINE.json configs[2] at full size", 27, [27_000_000] * 5, 21, 0.01, args.seed + 3, (0, 19), 1_000_000),
            ("BASELINE.json configs[3] at full size, all 64 genomes anchored", 64, [20_000_000] * 10, 31, 0.005, args.seed + 4, (0, 41), 500_000)):
        try:
            legs.append(baseline_config_leg(ctx, dev, args, label, G, lens, k, d, seed, picks, sn))
        except Exception as e:  # noqa: BLE001 — a failed extra leg must not take the measured line with it
            legs.append({"config": label, "error": f"{type(e).__name__}: {e}"})
    e2e = None
    if not args.no_e2e:
        try:
            free = shutil.disk_usage(tempfile.gettempdir()).free
            if free < 80e9:
                e2e = {"skipped": f"{free / 1e9:.0f} GB free under {tempfile.gettempdir()}: 13 GB of FASTA + the index need more"}
            else:
                # In a process of its own, as a user's `panagram index` is: this one has allocated and freed hundreds of GB by
                # now, and hipFree's cost is paid inside the NEXT large hipMalloc on this stack (40 ms per GB: the 90 GB table of
                # this leg took 2.5 s to create here, 0.25 s in a fresh process).
                import subprocess
                ctx.trim()
                torch.cuda.empty_cache()
                root = tempfile.mkdtemp(prefix="pg_bench_e2e4_")
                try:
                    recs = []
                    # (the FASTA files by one process, Index.run() in the next — after a pause: a process that starts right behind one
                    # that held 150-190 GB, as this one did until a moment ago, finds its large hipMallocs waiting, 24-40 ms per GB
                    # (2 s for the table, 1-2 s for the rows: tools/e2e_fresh.py runs one behind the other show it every other time,
                    # HISTORY round 6 item 12) — the driver's housekeeping of the memory just released, not this job's work)
                    for mode in ("write", "run"):
                        if mode == "run":
                            shutil.rmtree(os.path.join(root, "idx"), ignore_errors=True)
                            time.sleep(float(os.environ.get("PG_BENCH_E2E_PAUSE_S", "10")))
                        cmd = [sys.executable, os.path.abspath(__file__), "--e2e-config4-child", f"{mode}:{root}", "--seed", str(args.seed)]
                        p = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
                        lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
                        if p.returncode != 0 or not lines:
                            raise RuntimeError(f"child ({mode}) exited {p.returncode}: {p.stderr[-400:]}")
                        recs.append(json.loads(lines[-1]))
                    e2e = recs[1]
                    e2e["fasta_files_written_in_s"] = recs[0].get("fasta_files_written_in_s")
                    e2e["process"] = ("a fresh process without torch, as `python -m panagram_amd index` is (python bench.py --e2e-config4-child "
                                      "run:DIR), the FASTA files written by another")
                finally:
                    shutil.rmtree(root, ignore_errors=True)
        except Exception as e:  # noqa: BLE001
            e2e = {"error": f"{type(e).__name__}: {e}"}
        ctx.trim()
        torch.cuda.empty_cache()
    return legs, e2e


def e2e_config4(dev, args, prewritten=None, write_only=None):
    """BASELINE.json configs[3] files -> files: 64 synthetic 200 Mb genomes, k = 31, d = 0.005, every genome anchored"""
    a4 = argparse.Namespace(**vars(args))
    a4.d, a4.seed = 0.005, args.seed + 4
    e2e = e2e_leg(dev, a4, 64, [20_000_000] * 10, 31, prewritten=prewritten, write_only=write_only)
    if write_only:
        return e2e
    e2e["config"] = "BASELINE.json configs[3]: 64 x 200 Mb, k=31, files -> files"
    e2e["payload_gb_per_s"] = e2e["bitmap_payload_bytes"] / e2e["anchor_and_write_s"] / 1e9
    e2e["index_write_gb_per_s"] = e2e["index_bytes_out"] / e2e["anchor_and_write_s"] / 1e9
    return e2e


def wide_legs(ctx, dev, args, k=21):
    """More than 64 genomes (rows of 9, 12 and 16 bytes): 65 / 96 / 128 synthetic 10 Mb genomes (5 contigs each), d = 0.01, all
    anchored per step in one co-scheduled launch + one statistics pass — the inline table layout (65..96 genomes) and the split
    layout (more) with k_epilogue_w; the first 200 000 rows of two anchors against the CPU oracle (k-mer DB of the samples by brute
    force with torch).  Outside the timed region of ``value``.  The reference's multi-DB loop: cpp/anchor.cpp:138-165."""
    out = []
    for G in (65, 96, 128):
        L, C = 10_000_000, 5  # (the shape of tools/lines.sh's --genomes G --genome-mb 10: what profiles/r5* were measured on)
        contig_lens = [L // C] * C
        pg = Pangenome(ctx, dev, G, contig_lens, 0.01, args.seed + 2, k, keep_ascii=True)
        sample_n = 200_000
        picks = [0, G - 1]
        samples = [pg.ascii[g][0][:sample_n] for g in picks]
        dbs = sample_db_by_brute_force(pg.ascii, samples, k, G)
        samples_host = [x.cpu().numpy() for x in samples]
        pg.ascii = None
        torch.cuda.empty_cache()
        results, merged = make_results(ctx, pg, True, False, 0)
        steps, warmup = 12, 6
        for _ in range(warmup):
            results[0].run()
        torch.cuda.synchronize()
        results[0].timing_reset()
        t0 = time.perf_counter()
        for _ in range(steps):
            results[0].run()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        p_ms, e_ms, nruns = results[0].timing_mean()
        pos = sum(pg.pos_per_genome)
        v, cdt, npos, ok = cpu_baseline(dbs, samples_host, k, G,
                                        lambda t, n: results[0].download(picks[t] * C, want_bitmap100=False)[0][:n])
        st = pg.stats
        out.append({"genomes": G, "genome_mb": L / 1e6, "nbytes": (G + 7) // 8, "value": pos * steps / dt, "unit": "k-mers/s",
                    "ms_per_step": 1e3 * dt / steps, "k_probe_ms": p_ms, "statistics_ms": e_ms, "positions_per_step": pos,
                    "table_keys": st["nkeys"], "table_bytes": st["bytes"], "table_keys_per_line_capacity": st["nslots"] // st["nbuckets"],
                    "minimizer_length": pg.table.minimizer, "table_spill_fraction": pg.table.measure_spill(), "table_build_s": pg.build_s,
                    "rows_equal_gpu": ok, "rows_check": f"first {sample_n} rows of genomes {picks} == CPU oracle rows"})
        for r in results:
            r.close()
        if merged is not None:
            merged.close()
        pg.close()
        ctx.trim()
    return out


def config5_leg(ctx, dev, args, genome_mb=None, contigs=24, sample_n=1_000_000, chunk_positions=None, per=1, keys_per_line=None):
    """BASELINE.json configs[4] AS SPECIFIED — 8 synthetic 3 Gb genomes (24 contigs of 125 Mb each), k=21, d=0.05: 1.7e10
    distinct k-mers, more than one GPU's 288 GB holds — on ONE GPU through the product's own pass mode
    (panagram_amd.distributed.ShardedAnchoring, what Index.run() runs when plan_sharding says "genome"): genome_blocks = 8,
    ONE genome per block; pass p builds the table of genome p in the same allocation (pg_table_clear), probes EVERY anchor
    position of all 8 genomes against it in co-scheduled chunk groups, extracts the block's bit column and ORs it into the
    anchors' full rows (accumulate).  With 8 GPUs the 8 passes run side by side, one per rank, and the columns travel by
    all-gather (the same code, world = 8); here one GPU does the 8 ranks' work one after the other, without a collective.
    The byte layout that must come out: cpp/anchor.cpp:139-164 (one byte per position, bit g = genome g).
    Checked: every anchor's own column counts every one of its positions; the first ``sample_n`` rows of anchor 0 and the
    LAST ``sample_n`` rows of anchor 5's last contig equal the CPU oracle's, whose k-mer DB for the samples is built by brute
    force with torch (no HIP kernel involved in the expected rows).  Outside the timed region of ``value``."""
    from panagram_amd import distributed as pdist
    from panagram_amd import engine
    G, k, d = 8, 21, 0.05
    L = int((genome_mb if genome_mb is not None else float(os.environ.get("PG_BENCH_C5_MB", "3000"))) * 1e6)
    contig_lens = [L // contigs] * contigs
    t_leg = time.perf_counter()
    genomes = synth_genomes_device(G, contig_lens, d, args.seed + 5, dev)
    torch.cuda.synchronize()
    synth_s = time.perf_counter() - t_leg
    # ---- expected rows of two samples: head of anchor 0, tail of anchor 5 (brute-force DB, CPU oracle) ----
    picks = [(0, 0, "head"), (5, contigs - 1, "tail")]
    n = min(sample_n, contig_lens[0] - k + 1)
    samples = [genomes[g][c][:n + k - 1] if where == "head" else genomes[g][c][-(n + k - 1):] for g, c, where in picks]
    t0 = time.perf_counter()
    dbs = sample_db_by_brute_force(genomes, samples, k, G)
    db_s = time.perf_counter() - t0
    samples_host = [x.cpu().numpy() for x in samples]
    # ---- packed sequences (0.375 byte per base); the ASCII goes ----
    seqsets = []
    for g in range(G):
        ss = engine.SeqSet(ctx, contig_lens)
        for c, t in enumerate(genomes[g]):
            ss.load_dev(c, t.data_ptr(), t.numel())
        seqsets.append(ss)
    ctx.synchronize()
    del genomes, samples
    torch.cuda.empty_cache()
    names = [f"g{g}" for g in range(G)]
    seqs = dict(zip(names, seqsets))
    pos_per_genome = [ss.total_kmers(k) for ss in seqsets]
    pos = sum(pos_per_genome)
    # ---- ONE table allocation for all blocks, sized from the genomes' sketches (run_genome_sharded does the same) ----
    sketch = engine.KmerSketch(ctx, k)
    est = 0
    for b0 in range(0, G, per):  # distinct k-mers of the largest block (the union of its genomes' sketches)
        sketch.reset()
        for ss in seqsets[b0:b0 + per]:
            sketch.add(ss)
        est = max(est, sketch.estimate())
    sketch.close()
    old_chunk = pdist.CHUNK_POSITIONS
    if chunk_positions:
        pdist.CHUNK_POSITIONS = chunk_positions
    try:
        # (blocks of two genomes in a dense table: the probe emits the block's bit columns itself, as Index.plan_sharding asks — no narrow rows)
        direct = True if (per == 2 and keys_per_line == "auto") else None
        sh = pdist.ShardedAnchoring(engine, ctx, k, G, per, 0, 1, seqs, {a: 0 for a in names}, None, None, direct_columns=direct)
    finally:
        pdist.CHUNK_POSITIONS = old_chunk
    # (``per`` genomes per block: the block's table holds the union of their k-mers — at d = 0.05 two genomes share a ninth of
    # theirs — created at ``keys_per_line`` keys per 128-byte line so that it fits beside the rows: fewer, denser passes)
    nblocks = (G + per - 1) // per
    expect = est + est // 32 + 1024
    if keys_per_line == "auto":
        # as Index.plan_sharding sizes a block table: what is free next to the full rows (one byte per anchor position at 8
        # genomes), the narrow result's rows and the library's reserve, at no fewer than the library's 3 keys per line
        torch.cuda.synchronize()
        free = torch.cuda.mem_get_info()[0]
        room = free - (8 << 30) - pos * ((G + 7) // 8) - (0 if direct else pos * ((per + 7) // 8))
        keys_per_line = max(3.0, round(expect * 128.0 * 1.02 / max(room, 1) + 0.05, 1))
        if keys_per_line > 6.2:
            raise RuntimeError(f"blocks of {per} genomes do not fit this GPU at any usable density ({keys_per_line} keys per line)")
        if keys_per_line <= 3.0:
            keys_per_line = None
    tbl = engine.PanTable(ctx, k, per, expected_keys=expect, coscheduled=G, **({"keys_per_line": float(keys_per_line)} if keys_per_line else {}))
    passes, done = [], []
    total_build = total_pass = 0.0
    for p in range(nblocks):
        if p:
            tbl.clear()
        ctx.synchronize()
        t0 = time.perf_counter()
        for j, g in enumerate(range(p * per, min(G, (p + 1) * per))):
            tbl.insert_seqset(j, seqsets[g])
        ctx.synchronize()
        build_s = time.perf_counter() - t0
        stt = tbl.stats()
        # (a measuring pass first, synchronised after every stage — probe / extract / merge — then the pass as the product
        # runs it, timed as a whole; OR-ing a block's bits into the rows twice changes nothing)
        ph = {}
        sh.run_pass(tbl, p, 1, True, None, phase_s=ph)
        ctx.synchronize()
        t0 = time.perf_counter()
        sh.run_pass(tbl, p, 1, True, (lambda a, res: (res.rows_epilogue(), done.append(a))) if p == nblocks - 1 else None)
        ctx.synchronize()
        pass_s = time.perf_counter() - t0
        total_build += build_s
        total_pass += pass_s
        passes.append({"genome_block": p, "table_keys": stt["nkeys"], "table_bytes": stt["bytes"], "minimizer_length": tbl.minimizer,
                       "table_build_s": build_s, "pass_ms": 1e3 * pass_s, "value": pos / pass_s,
                       "probe_ms": 1e3 * ph.get("probe", 0.0), "extract_ms": 1e3 * ph.get("extract", 0.0), "merge_ms": 1e3 * ph.get("merge", 0.0)})
    # ---- checks on the completed rows ----
    own_ok = all(int(sh.full[a].colsums()[gi]) == pos_per_genome[gi] for gi, a in enumerate(names))

    def gpu_rows(t, nrows):
        g, c, where = picks[t]
        rows = sh.full[names[g]].download(c, want_bitmap100=False)[0]
        return rows[:nrows] if where == "head" else rows[-nrows:]
    v, cdt, npos, ok = cpu_baseline(dbs, samples_host, k, G, gpu_rows)
    out = {
        "workload": f"BASELINE.json configs[4]: 8 synthetic {L / 1e6:g} Mb genomes ({contigs} contigs each), k=21, d=0.05, genome_blocks={nblocks} "
                    f"({per} genome(s) per block), all 8 anchored, on ONE GPU as {nblocks} passes through distributed.ShardedAnchoring "
                    "(table of block p built in one re-used allocation, every anchor position probed, the block's bit columns extracted and "
                    "OR-ed into the full rows); with 8 GPUs one-genome blocks run side by side in ONE pass and each anchor's columns go to its writer",
        "positions": pos, "genome_blocks": nblocks, "genomes_per_block": per, "table_keys_per_line_at_creation": keys_per_line or 3,
        "columns_straight_from_the_probe": bool(getattr(sh, "_direct", False)),
        "chunk_groups_per_pass": len(sh.groups),
        "value": pos / total_pass, "value_with_table_builds": pos / (total_pass + total_build), "unit": "k-mers/s",
        "value_note": "the whole job on ONE GPU: all positions / the passes' time (each pass probes every position against one "
                      "block's table); with one genome per block per_pass_value_mean is what one rank of an 8-GPU run sustains, the "
                      "job's 8-GPU rate if the exchange hides behind the next group's probe (DESIGN.md section 6)",
        "per_pass_value_mean": float(np.mean([x["value"] for x in passes])),
        "passes_s": total_pass, "table_builds_s": total_build, "passes": passes,
        "union_keys_sum_over_blocks": int(sum(x["table_keys"] for x in passes)),
        "anchors_hold_all_own_kmers": bool(own_ok), "anchors_completed": len(done),
        "rows_equal_gpu": ok,
        "rows_check": f"first {n} rows of genome 0 and last {n} rows of genome 5's last contig: CPU oracle rows (k-mer DB of the "
                      f"samples built by brute force with torch in {db_s:.1f} s, {sum(len(kk) for kk, _ in dbs)} keys) == the rows "
                      f"the {nblocks} passes assembled",
        "synth_s": synth_s, "leg_s": time.perf_counter() - t_leg,
    }
    sh.close()
    tbl.close()
    for ss in seqsets:
        ss.close()
    ctx.trim()
    torch.cuda.empty_cache()
    return out


def write_fasta_from_device(path, names, contigs, width=80):
    """FASTA text of ASCII tensors in HBM: the line breaks are put in on the GPU, the host only writes the bytes"""
    with open(path, "wb") as f:
        for nm, t in zip(names, contigs):
            f.write(b">" + nm.encode() + b"\n")
            n = t.numel()
            full = (n // width) * width
            if full:
                body = torch.empty((n // width, width + 1), dtype=torch.uint8, device=t.device)
                body[:, :width] = t[:full].view(-1, width)
                body[:, width] = 10
                f.write(body.cpu().numpy().tobytes())
            if n > full:
                f.write(t[full:].cpu().numpy().tobytes() + b"\n")


def e2e_leg(dev, args, G, contig_lens, k, prewritten=None, write_only=None):
    """anchor END TO END — FASTA files on disk -> (table) -> every anchor's BGZF / .gzi / TSV files on disk
    through the product's `panagram index` entry, Index.run() (the reference times the whole rule,
    workflow/Snakefile:43-44).  Outside the timed region of `value`; the table build (FASTA read + GPU parse + sketch +
    inserts) is timed on its own.  Checked: every anchor's files exist and its own column of total_paircounts.csv
    counts every one of its positions."""
    import shutil
    import tempfile
    import pandas as pd
    from panagram_amd import index as pidx
    # (prewritten / write_only: the FASTA files written by one process, Index.run() timed in another that has allocated and freed
    # nothing yet — see baseline_config_legs)
    root = prewritten or write_only or tempfile.mkdtemp(prefix="pg_bench_e2e_")
    try:
        nbytes_in, write_s = 0, None
        if prewritten is None:
            genomes = synth_genomes_device(G, contig_lens, args.d, args.seed, dev)
            rows = ["name\tfasta"]
            t0 = time.perf_counter()
            for g in range(G):
                fa = os.path.join(root, f"g{g}.fa")
                write_fasta_from_device(fa, [f"chr{c + 1}" for c in range(len(contig_lens))], genomes[g])
                rows.append(f"g{g}\t{fa}")
            del genomes
            torch.cuda.empty_cache()
            with open(os.path.join(root, "samples.tsv"), "w") as f:
                f.write("\n".join(rows) + "\n")
            write_s = time.perf_counter() - t0
            if write_only:
                return {"fasta_files_written_in_s": write_s}
        nbytes_in = sum(os.path.getsize(os.path.join(root, f"g{g}.fa")) for g in range(G))
        t0 = time.perf_counter()
        import contextlib
        with contextlib.redirect_stdout(sys.stderr):  # (the CLI's progress lines: stdout carries the ONE JSON line)
            idx = pidx.Index(os.path.join(root, "samples.tsv"), prefix=os.path.join(root, "idx"), k=k)
            idx.run()
        t2 = time.perf_counter()
        t1 = t0 + idx.timings.get("load_inputs_s", 0.0) + idx.timings.get("table_build_s", 0.0)
        npos = G * sum(L - k + 1 for L in contig_lens)
        out_bytes = 0
        for g in range(G):
            adir = os.path.join(root, "idx", "anchor", f"g{g}")
            for fn in ("bitmap.1.gz", "bitmap.1.gzi", "bitmap.100.gz", "bitmap.100.gzi", "bitsum.bins.tsv", "chrs.tsv", "total_paircounts.csv"):
                out_bytes += os.path.getsize(os.path.join(adir, fn))
            tp = pd.read_csv(os.path.join(adir, "total_paircounts.csv"), index_col="name")
            assert int(tp.loc[f"g{g}", "count"]) == npos // G, "an anchor's own column must count every one of its positions"
        return {
            "what": "Index.run(): FASTA files on disk -> k-mer table on the GPU -> every genome anchored -> bitmap.{1,100}.gz/.gzi, "
                    "bitsum.bins.tsv, chrs.tsv, total_paircounts.csv on disk (GPU BGZF); table build included in seconds",
            "seconds": t2 - t0, "value": npos / (t2 - t0), "unit": "k-mers/s", "positions": npos,
            "read_parse_sketch_s": idx.timings.get("load_inputs_s"), "table_insert_s": idx.timings.get("table_build_s"),
            "table_build_s": t1 - t0, "anchor_and_write_s": t2 - t1, "anchor_and_write_value": npos / (t2 - t1),
            "fasta_bytes_in": nbytes_in, "index_bytes_out": out_bytes, "bitmap_payload_bytes": npos * ((G + 7) // 8) * 101 // 100,
            "fasta_files_written_in_s": write_s, "tmp_dir_fs": root.rsplit("/", 1)[0],
            "row_batches": idx.timings.get("batches"), "anchor_batches_s": idx.timings.get("anchor_batches_s"), "writers_wait_s": idx.timings.get("writers_wait_s"),
        }
    finally:
        if not (prewritten or write_only):
            shutil.rmtree(root, ignore_errors=True)


def _revcomp_ascii(t):
    lut = torch.zeros(256, dtype=torch.uint8, device=t.device)
    for a, b in zip(b"ACGT", b"TGCA"):
        lut[a] = b
    return lut[t.flip(0).long()]


def robustness_legs(ctx, dev, args, k):
    """The headline is measured on collinear genomes that differ by SNPs; here, at 8 x 50 Mb and outside the timed
    region, the same co-scheduled launch on inputs that are less kind: (a) every derived genome carries inversions and
    lists its chromosomes in an order of its own (scheduled by homology class, as Index.run() pairs contigs by record
    id); (b) a tenth of every genome is one young repeat family (thousands of 3-kb copies at 3 % divergence: huge
    minimizer groups).  Each with its one-launch-per-genome value beside it.  Invariant checked: an anchor holds all of
    its own k-mers."""
    from panagram_amd import engine
    G, C, L = 8, 5, 10_000_000
    out = {}

    def measure(genomes, classes, label, what, minimizer=-1, into=None):
        seqsets = []
        for g in range(G):
            ss = engine.SeqSet(ctx, [int(t.numel()) for t in genomes[g]])
            for c, t in enumerate(genomes[g]):
                ss.load_dev(c, t.data_ptr(), t.numel())
            seqsets.append(ss)
        ctx.synchronize()
        exp_keys = int(sum(t.numel() for t in genomes[0]) * (1 + (G - 1) * 0.25))
        # (as Index.build_table decides the table's density: sparser where HBM is plentiful, unless one genome's sketch says repeat-rich)
        sk = engine.KmerSketch(ctx, k)
        sk.add(seqsets[0])
        distinct = min(1.0, sk.estimate() / max(1, seqsets[0].total_kmers(k)))
        sk.close()
        kpl = engine.PanTable.roomy_density(ctx, k, G, exp_keys, 0, distinct_fraction=distinct)
        tbl = engine.PanTable(ctx, k, G, expected_keys=exp_keys, coscheduled=G, keys_per_line=kpl)
        if minimizer >= 0:
            tbl.set_minimizer(minimizer)
        tb = time.perf_counter()
        for g in range(G):
            tbl.insert_seqset(g, seqsets[g])
        ctx.synchronize()
        build_s = time.perf_counter() - tb
        npos = sum(ss.total_kmers(k) for ss in seqsets)
        merged = engine.SeqSet.concat(ctx, seqsets)
        res = engine.AnchorResult(tbl, merged, colsums=True)
        res.coschedule(np.repeat(np.arange(G), [len(g) for g in genomes]), contig_class=np.asarray(classes, np.uint32))
        res.run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            res.run()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        cs = res.contig_colsums(0, len(genomes[0])).sum(axis=0)
        # (an anchor holds every one of its k-mers; positions whose window holds an N have none)
        own_min = seqsets[0].total_kmers(k) - sum(int((t == ord("N")).sum().item()) for t in genomes[0]) * k
        assert own_min <= int(cs[0]) <= seqsets[0].total_kmers(k), "anchor genome 0 must contain every one of its k-mers"
        res.close()
        merged.close()
        singles = [engine.AnchorResult(tbl, ss, colsums=True) for ss in seqsets]
        for r in singles:
            r.run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            for r in singles:
                r.run()
        torch.cuda.synchronize()
        dt1 = (time.perf_counter() - t0) / 3
        (out if into is None else into)[label] = {
            "what": what, "value": npos / dt, "per_genome_launches_value": npos / dt1, "unit": "k-mers/s",
            "positions_per_step": npos, "table_keys": tbl.stats()["nkeys"], "table_build_s": build_s,
            "minimizer_length": tbl.minimizer, "table_spill_fraction": tbl.measure_spill(),
            "distinct_kmers_per_position_of_one_genome": round(distinct, 3), "table_keys_per_line_at_creation": kpl or 3.0}
        for r in singles:
            r.close()
        for ss in seqsets:
            ss.close()
        tbl.close()
        ctx.trim()
        torch.cuda.empty_cache()

    # (a) inversions + every derived genome's chromosomes in an order of its own
    base = synth_genomes_device(G, [L] * C, args.d, args.seed + 101, dev)
    rng = np.random.default_rng(args.seed + 5)
    genomes, classes = [base[0]], list(range(C))
    for g in range(1, G):
        contigs = []
        for t in base[g]:
            t = t.clone()
            for _ in range(3):
                ln = int(rng.integers(200_000, 2_000_000))
                s0 = int(rng.integers(0, L - ln))
                t[s0:s0 + ln] = _revcomp_ascii(t[s0:s0 + ln])
            contigs.append(t)
        order = rng.permutation(C)
        genomes.append([contigs[i] for i in order])
        classes += [int(i) for i in order]
    del base
    measure(genomes, classes, "inversions_and_shuffled_contig_order",
            "8 x 50 Mb, 1 % SNPs, 3 inversions of 0.2-2 Mb per chromosome, every derived genome listing its chromosomes in its own order")
    del genomes
    torch.cuda.empty_cache()
    # (b) one tenth of the genome is a young repeat family
    elem_n, copies = 3000, (C * L // 10) // 3500
    gen = torch.Generator(device=dev)
    gen.manual_seed(args.seed + 77)
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    elem = torch.randint(0, 4, (elem_n,), dtype=torch.uint8, device=dev, generator=gen)
    uniq = synth_genomes_device(G, [L * 9 // 10] * C, args.d, args.seed + 202, dev)
    genomes = []
    for g in range(G):
        contigs = []
        per = copies // C
        for c in range(C):
            e = elem.repeat(per, 1)
            mut = torch.rand(e.shape, device=dev, generator=gen) < 0.03
            e = torch.where(mut, (e + torch.randint(1, 4, e.shape, dtype=torch.uint8, device=dev, generator=gen)) & 3, e)
            spacer = torch.randint(0, 4, (per, 500), dtype=torch.uint8, device=dev, generator=gen)
            fam = acgt[torch.cat([e, spacer], dim=1).reshape(-1).long()]
            contigs.append(torch.cat([uniq[g][c], fam]))
        genomes.append(contigs)
    del uniq
    measure(genomes, list(range(C)) * G, "repeat_family_10pct",
            f"8 x 50 Mb, 90 % unique sequence at 1 % SNPs + 10 % one repeat family ({copies} copies of a 3-kb element at 3 % divergence per genome)")
    del genomes
    torch.cuda.empty_cache()
    # (c) a plant-like pangenome: half of every genome is transposable-element copies of many families and ages
    genomes, what = plant_like_pangenome(dev, G, C, L, args.d, args.seed + 303)
    leg = {}
    measure(genomes, list(range(C)) * G, "library_choice", what, into=leg)
    measure(genomes, list(range(C)) * G, "window_capped_at_4", "the same, minimizer window capped at 4 (what PG_TABLE_WMAX=4 chooses: m = k - 3)",
            minimizer=k - 3, into=leg)
    a, b = leg["library_choice"], leg["window_capped_at_4"]
    out["plant_like_50pct_repeats"] = dict(a, window_capped_at_4={x: b[x] for x in ("value", "per_genome_launches_value", "minimizer_length",
                                                                                    "table_keys", "table_spill_fraction", "table_build_s")})
    return out


def plant_like_pangenome(dev, G, C, L, d, seed, families=30, repeat_share=0.5):
    """What the reference is used on (README.md: plant pangenomes; panagram/introgressions/run_example.sh:8-22 simulates from
    A. thaliana chr1): an ancestral genome of C chromosomes of L bases in which ``repeat_share`` of the sequence is copies of
    ``families`` transposable-element families — element length 300 b .. 8 kb, copy number 10^2 .. 10^4, copies diverged from
    their family's consensus by the family's AGE, 1 .. 20 % (log-uniform each; copy numbers scaled so that the share comes
    out) — scattered between stretches of unique sequence, plus two 1-Mb tandem arrays of a 178-b satellite unit (copies 2 %
    apart) and three assembly gaps of 500 N per chromosome.  The G genomes are that ancestor with substitutions at rate d
    (genome 0: the ancestor) — the insertions are shared, as most are within a species — and gaps of their own.
    Returns ([genome][chromosome] ASCII tensors, description)."""
    rng = np.random.default_rng(seed)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    total = C * L
    sat_bases = 2 * 1_000_000
    te_target = int(repeat_share * total) - sat_bases
    lens = np.exp(rng.uniform(np.log(300), np.log(8000), families)).astype(np.int64)
    copies = np.exp(rng.uniform(np.log(100), np.log(10000), families))
    copies = np.maximum(100, copies * te_target / float((lens * copies).sum())).astype(np.int64)
    ages = np.exp(rng.uniform(np.log(0.01), np.log(0.20), families))

    def mutated(cons, n, rate):  # n copies of a consensus (codes 0..3), each base substituted with probability `rate`
        e = cons.repeat(n, 1)
        mut = torch.rand(e.shape, device=dev, generator=gen) < rate
        return torch.where(mut, (e + torch.randint(1, 4, e.shape, dtype=torch.uint8, device=dev, generator=gen)) & 3, e)

    pieces = []  # every element copy of every family, as rows of the families' matrices
    for f in range(families):
        cons = torch.randint(0, 4, (int(lens[f]),), dtype=torch.uint8, device=dev, generator=gen)
        m = mutated(cons, int(copies[f]), float(ages[f]))
        pieces += list(m.unbind(0))
    order = rng.permutation(len(pieces))
    te_bases = int((lens * copies).sum())
    sats = []
    for _ in range(2):
        unit = torch.randint(0, 4, (178,), dtype=torch.uint8, device=dev, generator=gen)
        sats.append(mutated(unit, 1_000_000 // 178, 0.02).reshape(-1))
    # chromosome c takes every C-th element copy, each followed by a stretch of unique sequence; the satellites go into the
    # middle of chromosomes 0 and 2
    uniq_total = total - te_bases - sum(int(x.numel()) for x in sats)
    base = []
    for c in range(C):
        mine = [pieces[i] for i in order[c::C]]
        room = L - sum(int(x.numel()) for x in mine) - (int(sats[c // 2].numel()) if c in (0, 2) else 0)
        room = max(room, len(mine))
        cuts = np.sort(rng.integers(0, room, len(mine)))
        gaps = np.diff(np.concatenate([[0], cuts, [room]]))  # unique stretches before / between / after the copies
        u = torch.randint(0, 4, (int(room),), dtype=torch.uint8, device=dev, generator=gen)
        parts, off = [], 0
        for j, x in enumerate(mine):
            parts.append(u[off:off + int(gaps[j])])
            off += int(gaps[j])
            parts.append(x)
            if c in (0, 2) and j == len(mine) // 2:
                parts.append(sats[c // 2])
        parts.append(u[off:])
        base.append(torch.cat(parts)[:L].contiguous())
    del pieces, sats
    genomes = []
    for g in range(G):
        contigs = []
        for b in base:
            t = b
            if g:
                mut = torch.rand(b.shape, device=dev, generator=gen) < d
                t = torch.where(mut, (b + torch.randint(1, 4, b.shape, dtype=torch.uint8, device=dev, generator=gen)) & 3, b)
            a = acgt[t.long()]
            for p in rng.integers(1000, int(b.numel()) - 2000, 3).tolist():
                a[p:p + 500] = ord("N")
            contigs.append(a)
        genomes.append(contigs)
    what = (f"{G} x {C * L / 1e6:g} Mb, plant-like: {100 * (te_bases + 2e6) / total:.0f} % of every genome is repeats — {families} transposable-"
            f"element families (elements of {int(lens.min())}-{int(lens.max())} b, {int(copies.min())}-{int(copies.max())} copies each, copies "
            f"{100 * ages.min():.0f}-{100 * ages.max():.0f} % off their consensus, {len(order)} copies in all) scattered through unique sequence, two 1-Mb "
            f"tandem arrays of a 178-b satellite, three 500-N gaps per chromosome; genomes = the ancestor + {100 * d:g} % SNPs")
    return genomes, what


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
        "collective": ("none (one rank)" if world <= 1 else
                       ("each anchor's bit columns to its writer only: all_to_all_single with split sizes (RCCL over xGMI; batched isend / irecv "
                        "on gloo / the host route)" if getattr(sh, "to_writers", False) else "all_gather_into_tensor of bit columns (RCCL over xGMI)")
                       + f" [PG_SHARD_EXCHANGE: {getattr(sh, 'exchange', 'rccl')}]"),
        "collective_bytes_received_per_rank_per_step": sh.bytes_received / max(1, steps + warmup),
        "parallelism": (f"EMULATED rank 0 of {nblocks}: the table of genome block 0 ({per} genome(s)) only, every position probed, "
                        f"columns extracted and merged, no collective" if emulated else
                        f"genome-sharded x{world}: {nblocks} genome blocks of {per}, every rank probes every position, "
                        f"columns sent to the anchors' writers, rows merged + statistics there"),
        "columns_from_the_probe": bool(getattr(sh, "_direct", False)),
    }
    sh.close()
    pg.close()
    ctx.trim()
    return out


def spawn_ranks(n):
    """Re-run this command as ``n`` ranks of one node under torch.distributed.run (one process per GPU, RCCL over
    xGMI), rendezvous on 127.0.0.1 at a free port; the ranks' output passes through.  Returns the exit code."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < n and os.environ.get("PG_BENCH_ONE_DEVICE", "") in ("", "0"):
        print(f"bench.py --gpus {n}: only {have} GPU(s) visible on this node", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("[bench] " + " ".join(cmd), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--mode", choices=["contig-sharded", "genome-sharded"], default="contig-sharded")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="contig-sharded mode with N > 1 GPUs.  weak (default): the configs[1] pangenome made N x longer, "
                         "each rank its own 5 contigs of every genome.  strong: the SAME pangenome whatever N (e.g. "
                         "--genomes 27 --genome-mb 135 = BASELINE configs[2], '1 vs 8 GPUs'), cut into pieces of homology "
                         "classes and dealt to the ranks exactly as Index.run() does (distributed.plan_class_pieces)")
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
    ap.add_argument("--piece-tiles", type=int, default=0, help="co-scheduling granularity in tiles of 1024 positions (0: library default, 64 down to 8 by genome count)")
    ap.add_argument("--no-compare", action="store_true", help="skip the untimed one-launch-per-genome comparison run")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-shapes", action="store_true", help="skip the north-star-shape leg (64 x 200 Mb, k=21)")
    ap.add_argument("--no-sharded-leg", action="store_true", help="skip the genome-sharded pipeline leg")
    ap.add_argument("--no-e2e", action="store_true", help="skip the files-to-files leg (Index.run() on FASTA files of this shape)")
    ap.add_argument("--no-robustness", action="store_true", help="skip the robustness legs (inversions + shuffled contig order; repeat family)")
    ap.add_argument("--no-baseline-configs", action="store_true", help="skip the BASELINE configs[2] / configs[3] legs at full size (and configs[3] files -> files)")
    ap.add_argument("--no-config5", action="store_true", help="skip the BASELINE configs[4] leg (8 x 3 Gb, d=0.05, 8 genome blocks as passes on this GPU)")
    ap.add_argument("--settle-s", type=float, default=0.4, help="seconds of untimed steps before the warm-up steps (the shader clock settles; 0: none)")
    ap.add_argument("--cpu-sample-mb", type=float, default=20.0, help="bases per thread of the CPU baseline leg (about 12 s of CPU work)")
    ap.add_argument("--e2e-config4-child", default="", metavar="write:DIR|run:DIR", help=argparse.SUPPRESS)  # (baseline_config_legs' files -> files leg, in processes of its own)
    ap.add_argument("--emulate-rank", type=str, default="", metavar="R/N",
                    help="one process plays rank R of an N-rank contig-sharded run (no collective): the N x longer "
                         "pangenome, the replicated table and rank R's contig group, for checking the multi-GPU "
                         "set-up on a one-GPU box; the reported value is this rank's alone")
    args = ap.parse_args()

    if args.e2e_config4_child:
        mode, d = args.e2e_config4_child.split(":", 1)
        dev = None if mode == "run" else torch.device("cuda", 0)
        print(json.dumps(e2e_config4(dev, args, prewritten=d if mode == "run" else None, write_only=d if mode == "write" else None)))
        return
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: no GPU visible (there is no CPU fallback)")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: become N ranks (one process per GPU) instead of silently measuring one
        sys.exit(spawn_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(1, args.gpus) and rank == 0:
        print(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}: the line reports the {world} rank(s) that actually run",
              file=sys.stderr, flush=True)
    # (test knobs, for checking the N > 1 code path on a ONE-GPU box: all ranks on device 0 and a gloo process group —
    # RCCL refuses two ranks on one device.  The numbers of such a run mean nothing.)
    one_device = os.environ.get("PG_BENCH_ONE_DEVICE", "") not in ("", "0")
    backend = os.environ.get("PG_BENCH_BACKEND", "nccl")
    if one_device:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    ranks_observed = 1
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        # the rank count the collective library actually sees (one all-reduce of ones), and that no two ranks share a GPU
        ones = torch.ones(1, device=dev if backend == "nccl" else "cpu", dtype=torch.int64)
        dist.all_reduce(ones)
        ranks_observed = int(ones.item())
        gpus = [None] * world
        dist.all_gather_object(gpus, (os.uname().nodename, str(getattr(torch.cuda.get_device_properties(local), "uuid", local))))
        if ranks_observed != world:
            sys.exit(f"bench.py: {ranks_observed} ranks answered the all-reduce, WORLD_SIZE says {world}")
        if len(set(gpus)) != world and not one_device:
            sys.exit(f"bench.py: {world} ranks on {len(set(gpus))} distinct GPU(s) — one process per GPU is the contract")

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

    # ---- files -> files first (outside the timed region of `value`, like every further leg): Index.run() on FASTA files of this
    # shape in a process that has not yet freed tens of GB — hipFree's cost is paid inside the NEXT large hipMalloc on this stack
    # (40 ms per GB), and behind the legs below it landed in this leg's table build (1.5-1.9 s instead of 0.02)
    e2e_first = None
    if world == 1 and (default_shape or os.environ.get("PG_BENCH_E2E_ANY")) and not args.no_e2e:  # (PG_BENCH_E2E_ANY: experiments on other shapes)
        try:
            e2e_first = e2e_leg(dev, args, G, contig_lens, k)
        except Exception as e:
            e2e_first = {"error": f"{type(e).__name__}: {e}"}
        ctx.trim()
        torch.cuda.empty_cache()

    # ---- k-mer set construction on the GPU (replaces kmc + kmc_tools; timed separately) ----
    keep_ascii = rank == 0 and world == 1 and not args.no_cpu_baseline
    strong = args.scaling == "strong" and (world > 1 or bool(args.emulate_rank))
    if strong:
        pg = Pangenome(ctx, dev, G, contig_lens, args.d, args.seed, k, minimizer=args.minimizer, pieces_of=(my_group, groups))
    else:
        pg = Pangenome(ctx, dev, G, contig_lens, args.d, args.seed, k, groups=groups, my_group=my_group, keep_ascii=keep_ascii,
                       minimizer=args.minimizer, rehash_kpb=None if (args.no_rehash or groups > 1 or args.keys_per_bucket <= 0) else args.keys_per_bucket,
                       coscheduled=1 if args.per_genome_launches else 0)  # (the timed mode's table is built for the way it is probed)
    st = pg.stats
    pg_rehashed = not (args.no_rehash or groups > 1 or args.keys_per_bucket <= 0)
    pos_per_step = sum(pg.pos_per_genome)

    results, merged = make_results(ctx, pg, not args.no_colsums, args.per_genome_launches, args.piece_tiles)

    def step():
        for r in results:
            r.run()
    # per-launch kernel durations come from HIP events recorded by the library on the stream the
    # kernels run on, one event set per run: the mean over ALL timed steps' launches (pg_result_timing_mean)
    # (untimed, before the W warm-up steps: the step repeated for --settle-s seconds, so that the timed region runs at the
    # shader clock a long job has — the first tenths of a second after the table build run 3-5 % slower, profiles/r4c_clock.txt;
    # reported as clock_settle_s)
    t_settle = time.perf_counter()
    while args.settle_s > 0 and time.perf_counter() - t_settle < args.settle_s:
        for _ in range(8):
            step()
        torch.cuda.synchronize()
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
        nc0 = len(pg.pieces[0]) if pg.pieces is not None else C
        cs = results[0].colsums() if args.per_genome_launches else results[0].contig_colsums(0, nc0).sum(axis=0)
        gaps = int(os.environ.get("PG_BENCH_N_RUNS", "0")) * nc0 * (500 + args.k)  # (experiments: positions whose window holds an N have no k-mer)
        assert pg.pos_per_genome[0] - gaps <= int(cs[0]) <= pg.pos_per_genome[0], "anchor genome 0 must contain every one of its k-mers"

    if strong and world > 1:  # every rank a different share of ONE pangenome: the units all ranks processed
        tot = torch.tensor([pos_per_step], device=dev if backend == "nccl" else "cpu", dtype=torch.int64)
        dist.all_reduce(tot)
        assert int(tot.item()) == pg.total_positions, "the ranks' pieces must cover the pangenome exactly once"
        value = pg.total_positions * args.steps / elapsed
    else:
        value = world * pos_per_step * args.steps / elapsed
    counters = load_counters(pos_per_launch, k, G) if not args.per_genome_launches else None
    if counters is not None and groups > 1:
        # a rank of a multi-GPU run launches the profiled kernel over as many positions, against an N x larger table:
        # the instruction count per position carries over, the HBM traffic of the one-GPU profile does not
        counters = {"SQ_INSTS_VALU": counters.get("SQ_INSTS_VALU"), "profiled_on": "one GPU (profiles/traffic.json)",
                    "round": counters.get("round"), "file_sha16": counters.get("file_sha16"), "sources_match": counters.get("sources_match")}
    shape = (G, round(args.genome_mb), k)
    baseline_config = {(8, 100, 21): "BASELINE.json configs[1]", (27, 135, 21): "BASELINE.json configs[2] at full size",
                       (64, 200, 31): "BASELINE.json configs[3] at full size, all 64 genomes anchored",
                       (8, 3000, 21): "the shape of BASELINE.json configs[4] on ONE GPU, at a divergence whose table fits"
                       }.get(shape, "not a BASELINE.json config")
    if groups > 1 and world == 1 and not strong:
        tbl_txt = ("the table of the k-mers of this rank's contigs (the rest of the pangenome only sets bits in it)"
                   if pg.filtered else "table of all of it")
        workload = (f"EMULATED rank {my_group} of {groups}: {G} synthetic {args.genome_mb * groups:g} Mb genomes, {tbl_txt}, "
                    f"this rank's {C} contigs of every genome anchored")
        parallelism = f"one GPU playing rank {my_group} of a contig-sharded x{groups} run"
    elif strong:
        workload = (("" if world > 1 else f"EMULATED rank {my_group} of {groups} (this rank's value alone): ") + f"{G} synthetic {args.genome_mb:g} Mb genomes ({C} contigs each), k={k}, d={args.d} ({baseline_config}): ONE "
                    f"pangenome whatever the GPU count, cut into {pg.plan_class_pieces} pieces of homology classes dealt to "
                    f"{groups} rank(s) (distributed.plan_class_pieces, what Index.run() does); a rank anchors its pieces of "
                    f"every genome in one co-scheduled launch against a table built from those pieces")
        parallelism = (f"contig-sharded x{groups}, strong scaling: planned positions per rank {pg.plan_loads}, "
                       "no data-path collective")
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
        "timed_region_s": elapsed,  # (the launch gets 3-5 % faster once the clock has settled: DESIGN.md section 4)
        "clock_settle_s": args.settle_s,  # untimed steps before the warm-up steps (see above)
        "higher_is_better": True,
        "scaling": "strong" if strong else "weak",
        "vs_baseline": None,
        "dtype": "u64",
        "data": "synthetic",
        "ranks_observed": ranks_observed,
        "config": {
            "workload": workload,
            "positions_per_step_per_gpu": pos_per_step,
            "table_keys": st["nkeys"], "table_bytes": st["bytes"], "keys_per_128B_line": round(st["nkeys"] / max(1, st["nbuckets"]), 3),
            "table_rehashed": bool(pg_rehashed), "minimizer_length": pg.table.minimizer,
            "table_keys_per_line_at_creation": getattr(pg, "keys_per_line", 0.0) or 3.0,
            "table_built_for_coscheduled_anchors": pg.coscheduled,
            "table_build_s": pg.build_s, "table_spill_fraction": pg.table.measure_spill(), "table_slots_per_line": pg.table.spill()[1],
            "probes_per_position": (G + 63) // 64, "nbytes": (G + 7) // 8,
            "colsums": not args.no_colsums,
            "launches_per_step": len(results),
            "schedule": "one launch per anchor genome" if args.per_genome_launches else
                        "one launch over all anchor genomes, tiles co-scheduled (homologous regions side by side)",
            "parallelism": parallelism,
        },
        "roofline": roofline_block(pos_per_launch, avg_launch_s, avg_epi_s, G, value / world, counters, nruns,
                                   bound="hbm" if args.per_genome_launches else "issue"),
    }

    if world == 1 and not args.per_genome_launches and not args.no_compare:
        # for comparison only (outside the timed region): the same work as one launch per genome — what a single-anchor
        # Index.run(), the run_anchor CLI with one FASTA or unrelated sequences get — against the table the product builds
        # for THAT mode (PanTable(coscheduled=1): how a table will be probed decides its minimizer window)
        from panagram_amd import engine as _engine
        pg1 = pg
        if args.minimizer < 0 and not pg_rehashed:
            t1 = _engine.PanTable(ctx, k, G, expected_keys=int(st["nkeys"] * 1.02) + 1024, coscheduled=1,
                                  keys_per_line=_engine.PanTable.roomy_density(ctx, k, G, int(st["nkeys"] * 1.02) + 1024, 0))
            tb1 = time.perf_counter()
            for g in range(G):
                t1.insert_seqset(g, pg.seqsets[g])
            ctx.synchronize()
            pg1 = type("PerGenomeTable", (), dict(table=t1, seqsets=pg.seqsets, G=G, pieces=None, contig_lens=pg.contig_lens))()
            pg1_build_s = time.perf_counter() - tb1
        alt, _ = make_results(ctx, pg1, not args.no_colsums, True, 0)

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
               "positions_per_launch": pos_per_step / len(alt), "minimizer_length": pg1.table.minimizer,
               "table": ("its own: built for one launch per genome (PanTable(coscheduled=1)) in %.3f s" % pg1_build_s) if pg1 is not pg
                        else "the timed region's"}
        if counters is not None and counters.get("per_genome_launches"):
            c2 = counters["per_genome_launches"]
            s2 = pgl["k_probe_ms_per_launch"] / 1e3
            pgl["hbm_counter_frac"] = c2["hbm_bytes_per_launch"] / s2 / HBM_PEAK
            if c2.get("SQ_INSTS_VALU"):
                pgl["valu_issue_frac_4cycle"] = c2["SQ_INSTS_VALU"] * VALU_CYCLES / (SIMDS * CLOCK_HZ * s2)
                pgl["valu_issue_frac_2cycle"] = c2["SQ_INSTS_VALU"] * 2.0 / (SIMDS * CLOCK_HZ * s2)
            pgl["traffic"] = c2["hbm_bytes_per_launch"]
        out["config"]["per_genome_launches"] = pgl
        out["config"]["per_genome_launches_value"] = pgl["value"]
        for r in alt:
            r.close()
        if pg1 is not pg:
            pg1.table.close()

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
    if e2e_first is not None:
        out["e2e"] = e2e_first
    if world == 1 and default_shape and not args.no_robustness:
        try:
            out["robustness"] = robustness_legs(ctx, dev, args, k)
            out["robustness"]["per_genome_launches_value_on_the_headline_workload"] = out["config"].get("per_genome_launches_value")
        except Exception as e:
            out["robustness"] = {"error": f"{type(e).__name__}: {e}"}
    if world == 1 and default_shape and not args.no_other_shapes:
        try:
            out["config"]["other_shapes"] = [north_star_leg(ctx, dev, args)]
        except Exception as e:
            out["config"]["other_shapes"] = [{"error": f"{type(e).__name__}: {e}"}]
    if world == 1 and default_shape and not args.no_other_shapes:
        try:
            out["config"]["wide_shapes"] = wide_legs(ctx, dev, args)
        except Exception as e:
            out["config"]["wide_shapes"] = [{"error": f"{type(e).__name__}: {e}"}]
    if world == 1 and default_shape and not args.no_other_shapes and not args.no_baseline_configs:
        try:
            out["config"]["baseline_configs"], out["e2e_config4"] = baseline_config_legs(ctx, dev, args)
        except Exception as e:
            out["config"]["baseline_configs"] = [{"error": f"{type(e).__name__}: {e}"}]
    if world == 1 and default_shape and not args.no_config5:
        try:
            out["config"]["config5_leg"] = config5_leg(ctx, dev, args)
        except Exception as e:
            out["config"]["config5_leg"] = {"error": f"{type(e).__name__}: {e}"}
        ctx.trim()
        torch.cuda.empty_cache()
        try:  # what Index.plan_sharding chooses on one GPU since round 6: two genomes per block in a denser table, half the passes
            out["config"]["config5_leg_two_genome_blocks"] = config5_leg(ctx, dev, args, per=2, keys_per_line="auto")
        except Exception as e:
            out["config"]["config5_leg_two_genome_blocks"] = {"error": f"{type(e).__name__}: {e}"}
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
