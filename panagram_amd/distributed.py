"""Multi-GPU modes of the anchor path (one process per GPU, ``torch.distributed``; backend
"nccl" is RCCL over xGMI on MI355X, "gloo" in the CPU tests).

1. **Contig-sharded** (BASELINE configs 2-4): the table is replicated, anchor
   contigs are independent units.  ``plan_shards`` assigns (genome, contig) units to ranks
   longest-first; every rank anchors its units and leaves per-contig part files; the rank
   that owns a genome assembles that genome's files in FASTA order.  No data-path
   collective — one barrier between the two phases.  Outputs do not depend on the GPU count.

2. **Genome-sharded** (config 5: the union of k-mer tables exceeds one GPU's 288 GB;
   ``run_genome_sharded``, reached from ``Index.run()``): the genomes are cut into contiguous
   blocks; a GPU holds the table of ONE block at a time (narrow: only that block's genomes) and
   probes every anchor position against it, chunk of contigs by chunk of contigs.  Per chunk the
   block's COMPACT BIT COLUMNS (one u64 per genome per 64 positions, ``k_cols_extract``) are
   all-gathered over RCCL/xGMI on a side stream while the next chunk is probed; the rank that
   writes the anchor merges the gathered blocks into full rows (``k_cols_merge``), takes the
   statistics from them (``pg_rows_epilogue``) and streams them into the BGZF files.  Per position
   a rank receives (n-1)/n row bytes.  More blocks than GPUs run as passes (one GPU: all of them),
   the blocks' bits OR-ed into the writers' rows pass by pass.
   (``combine_rows_``: a uint8 SUM all-reduce of full-width partial rows — ranks own disjoint bits,
   so SUM == OR; RCCL has no bitwise reductions — moves 2(n-1)/n bytes per row byte; kept as the
   reference formulation the column exchange is tested against.)
"""
from __future__ import annotations

import os
import shutil
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

Unit = Tuple[str, int, int]  # (genome name, contig index, k-mer count)


def plan_shards(units: Sequence[Unit], world: int) -> List[List[Unit]]:
    """Longest-processing-time greedy; deterministic (ties by genome, contig)."""
    order = sorted(units, key=lambda u: (-u[2], u[0], u[1]))
    loads = [0] * world
    shards: List[List[Unit]] = [[] for _ in range(world)]
    for u in order:
        r = min(range(world), key=lambda i: (loads[i], i))
        shards[r].append(u)
        loads[r] += u[2]
    for sh in shards:
        sh.sort(key=lambda u: (u[0], u[1]))
    return shards


def genome_owner(genome_id: int, ngenomes: int, world: int) -> int:
    """Contiguous blocks of ceil(N/world) genomes per rank (genome-sharded mode)."""
    per = (ngenomes + world - 1) // world
    return genome_id // per


# ---------------------------------------------------------------------------
# contig-sharded index build
# ---------------------------------------------------------------------------
def _parts_dir(genome) -> str:
    return os.path.join(genome.prefix, ".parts")


BGZF_EOF = bytes([0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0x00, 0x42, 0x43, 0x02, 0x00, 0x1b, 0x00, 0x03, 0x00,
                  0, 0, 0, 0, 0, 0, 0, 0])


def concat_bgzf(parts: Sequence[Tuple[str, str]], out_gz: str, out_gzi: str) -> None:
    """BGZF files are block-concatenable: the fragments ``(gz, gzi)`` — each a complete BGZF file ending in the EOF
    block — become ONE file (their blocks back to back, one EOF block at the end) with one ``.gzi`` whose entries
    are the fragments' own, re-based, plus one for every fragment's first block.  The payload is the fragments'
    payloads in order; block boundaries fall where the fragments' did (a reader goes by the .gzi, index.py:793-845)."""
    entries: List[Tuple[int, int]] = []
    cpos = upos = 0
    with open(out_gz + ".tmp", "wb") as out:
        for gz, gzi in parts:
            with open(gz, "rb") as f:
                data = f.read()
            if not data.endswith(BGZF_EOF):
                raise ValueError(f"{gz}: not a complete BGZF file")
            data = data[:-len(BGZF_EOF)]
            if not data:
                continue  # an empty fragment (a contig without k-mers)
            raw = np.fromfile(gzi, "<u8")
            n = int(raw[0])
            own = raw[1:1 + 2 * n].reshape(n, 2)
            # uncompressed size of the fragment: start of its last block + that block's ISIZE (last 4 bytes)
            last_u = int(own[-1, 1]) if n else 0
            usize = last_u + int.from_bytes(data[-4:], "little")
            if cpos:
                entries.append((cpos, upos))
            entries += [(cpos + int(c), upos + int(u)) for c, u in own]
            out.write(data)
            cpos += len(data)
            upos += usize
        out.write(BGZF_EOF)
    with open(out_gzi + ".tmp", "wb") as g:
        g.write(np.array([len(entries)] + [x for e in entries for x in e], "<u8").tobytes())
    os.replace(out_gz + ".tmp", out_gz)
    os.replace(out_gzi + ".tmp", out_gzi)


# A piece: k-mer positions [start, start + npos) of contig ``ci`` of genome ``name`` (a whole contig: start 0, npos = its
# k-mer count), in homology class ``cls``, piece number ``j`` of that class
Piece = Tuple[str, int, int, int, int, int]  # (name, ci, start, npos, cls, j)


MIN_PIECE = int(os.environ.get("PG_MIN_PIECE", str(1 << 20)))  # k-mer positions: no contig is cut into shorter pieces


def piece_alignment(lowres_step: int) -> int:
    """Piece starts are multiples of this: whole strides of the low-resolution bitmap (a piece's every lowres_step-th
    row is the contig's) and whole 32-base words of the packed sequence (``pg_seqset_slice``).  Bins need no alignment:
    a piece's statistics are taken over the contig's bin windows clipped to it, and bins that two pieces share are
    added up when the genome is assembled."""
    return int(np.lcm(int(lowres_step), 32))


def plan_class_pieces(contigs: Sequence[Tuple[str, int, int, int]], world: int, lowres_step: int = 100,
                      pieces_per_rank: Optional[int] = None, min_piece: Optional[int] = None) -> List[List[Piece]]:
    """``_plan_class_pieces`` at the granularity (4 to 8 class pieces per rank, unless ``pieces_per_rank`` pins it) whose
    fullest rank is lightest; ties go to the coarser plan."""
    if pieces_per_rank is not None or world <= 1:
        return _plan_class_pieces(contigs, world, lowres_step, pieces_per_rank or 4, min_piece)
    best, best_load = None, None
    for ppr in range(4, 9):
        plan = _plan_class_pieces(contigs, world, lowres_step, ppr, min_piece)
        load = max(sum(p[3] for p in sh) for sh in plan)
        if best is None or load < best_load:
            best, best_load = plan, load
    return best


def _plan_class_pieces(contigs: Sequence[Tuple[str, int, int, int]], world: int, lowres_step: int,
                       pieces_per_rank: int, min_piece: Optional[int]) -> List[List[Piece]]:
    """The contig-sharded partition that keeps the co-scheduling gain: the unit of work is a PIECE OF A
    HOMOLOGY CLASS — the same stretch (by relative position) of the homologous contigs of EVERY anchor genome — so that
    a rank's one co-scheduled launch still finds the lines its genomes share in L2.  ``contigs`` = ``(genome, contig
    index, k-mer count, class)``.  A class is cut into about ``size / (total / (world * pieces_per_rank))`` pieces,
    its contigs at the same relative places rounded to ``piece_alignment``, none shorter than ``min_piece`` positions;
    the class pieces are dealt to the ranks longest-first.  Deterministic: every process computes the same plan."""
    world = max(1, int(world))
    align = piece_alignment(lowres_step)
    min_piece = max(align, MIN_PIECE if min_piece is None else int(min_piece))
    by_class: Dict[int, List[Tuple[str, int, int]]] = {}
    for name, ci, nk, cls in contigs:
        by_class.setdefault(int(cls), []).append((name, int(ci), int(nk)))
    total = sum(nk for _, _, nk, _ in contigs)
    target = max(1, total // (world * max(1, pieces_per_rank)))
    units: List[Tuple[int, int, int, List[Piece]]] = []  # (weight, class, j, pieces)
    # Small classes are dealt in BUNDLES of consecutive classes (the first genome's contig order), a quarter of a rank's
    # target share at most: a draft assembly of 20 000 contigs dealt contig by contig leaves every rank every other contig
    # — 20 000 fragments, markers and bin tables per genome to write and to put together again (two ranks on 4 x 4 000
    # contigs: 35 s against 0.4 s on one).  A rank's contigs of a bundle are neighbours in their genomes, so they leave
    # as ONE fragment (run_index_sharded).
    cap = max(min_piece, target // 4)
    bundle: List[Piece] = []
    bundle_w, bundle_cls = 0, 0

    def close_bundle():
        nonlocal bundle, bundle_w
        if bundle:
            units.append((bundle_w, bundle_cls, 0, bundle))
        bundle, bundle_w = [], 0

    for cls in sorted(by_class):
        members = by_class[cls]
        size = sum(nk for _, _, nk in members)
        npieces = 1
        if world > 1:
            # the cut is decided by the class's LONG members: a short scaffold that an assembly pairs with a 200 Mb
            # chromosome (accession-style names fall back to pairing by position) must not make the class one unit —
            # one rank would anchor that chromosome of every genome, against a table larger than the planner priced.
            # Members too short for that many pieces stay whole, in piece 0.
            npieces = max(1, min(int(round(size / target)), max(nk for _, _, nk in members) // min_piece))
        if world > 1 and npieces == 1 and size < cap:
            if bundle and bundle_w + size > cap:
                close_bundle()
            if not bundle:
                bundle_cls = cls
            bundle += [(name, ci, 0, nk, cls, 0) for name, ci, nk in members if nk > 0]
            bundle_w += size
            continue
        close_bundle()
        for j in range(npieces):
            pieces = []
            for name, ci, nk in members:
                if nk < npieces * min_piece:  # a short member of a class of long ones: whole, with the class's first piece
                    b0, b1 = (0, nk) if j == 0 else (0, 0)
                else:
                    b0 = 0 if j == 0 else align * int(round(j * nk / (npieces * align)))
                    b1 = nk if j == npieces - 1 else align * int(round((j + 1) * nk / (npieces * align)))
                if b1 > b0:
                    pieces.append((name, ci, b0, b1 - b0, cls, j))
            if pieces:
                units.append((sum(p[3] for p in pieces), cls, j, pieces))
    close_bundle()
    order = sorted(units, key=lambda u: (-u[0], u[1], u[2]))
    loads = [0] * world
    shards: List[List[Tuple[int, int, List[Piece]]]] = [[] for _ in range(world)]
    for w, cls, j, pieces in order:
        r = min(range(world), key=lambda i: (loads[i], i))
        shards[r].append((cls, j, pieces))
        loads[r] += w
    out: List[List[Piece]] = []
    for sh in shards:
        sh.sort(key=lambda u: (u[0], u[1]))
        out.append([p for _, _, pieces in sh for p in pieces])
    return out


def _plan_signature(index, world: int, plan_digest: str) -> str:
    """what a fragment left in ``.parts`` must carry to be taken for this run's: the parameters, the inputs (path, size,
    modification time) and the plan.  Outputs are a function of exactly these, so a fragment of an earlier, aborted run
    with the same signature holds the same bytes this run would write."""
    import hashlib
    h = hashlib.sha1()
    h.update(repr((index.k, world, tuple(index.steps), sorted(index.result_geometry.items()), index.ngenomes,
                   index.bgzf_level, plan_digest)).encode())
    for name, g in index.genomes.items():
        if isinstance(g.fasta, str) and os.path.exists(g.fasta):
            st = os.stat(g.fasta)
            h.update(repr((name, g.fasta, st.st_size, st.st_mtime_ns)).encode())
        else:
            h.update(repr((name, None)).encode())
    return h.hexdigest()


def _claim(path: str) -> Optional[int]:
    """Who assembles a genome: an exclusive advisory lock (``flock``) on ``path``, created if need be.  Returns the open
    descriptor — the claim is held for as long as it stays open, and the kernel drops it with the process, so a rank that
    died leaves no claim behind — or None: another rank holds it, or the directory is gone (the genome was assembled
    and cleaned up meanwhile).  (A claim file created with O_EXCL and filled afterwards could be read empty by a second
    rank, taken for a dead process's, removed and claimed again: two ranks assembling one genome at once, seen on the
    GPU box under torchrun.)  A lock on a name that was removed since it was opened excludes nobody: it is refused."""
    import fcntl
    try:
        fd = os.open(path, os.O_CREAT | os.O_RDWR, 0o644)
    except FileNotFoundError:
        return None
    try:
        fcntl.flock(fd, fcntl.LOCK_EX | fcntl.LOCK_NB)
        if os.fstat(fd).st_ino == os.stat(path).st_ino:  # (only a holder of the claim removes the name: it stays ours)
            return fd
    except OSError:
        pass
    os.close(fd)
    return None


def run_index_sharded(index, rank: int, world: int, barrier: Optional[Callable[[], None]] = None,
                      pieces_per_rank: Optional[int] = None) -> None:
    """The contig-sharded multi-GPU mode (what ``Index.run()`` does with more than one rank when the table
    fits a GPU).  Every rank calls this.

    Work is dealt as pieces of homology classes (``plan_class_pieces``): a rank anchors, for its pieces, the
    homologous stretch of EVERY anchor genome in one co-scheduled launch per batch — the same launch shape as the
    single-GPU path, so the genomes still share their table lines in L2.  Its table holds the k-mers of its own pieces
    only (built from them; every sample then only sets its bits, ``pg_table_update_seqset``): 1 / world of the
    pangenome's keys per GPU, the same rows.  Per piece it leaves finished BGZF fragments — compressed on the GPU
    straight out of HBM — plus the piece's bins, column sums and gene histograms, the marker file written last.
    No data-path collective and no rendezvous: whoever finds a genome's pieces complete (the rank that finishes last,
    at the latest) assembles that genome — fragments concatenated in order (``concat_bgzf``), bin rows appended,
    sums added — under an exclusive claim.  ``barrier`` (torch.distributed's, when a process group exists) is only
    used to make the last scan see every rank's markers whatever the file system.  The decompressed outputs do not
    depend on the GPU count (BGZF block boundaries follow the fragments)."""
    import hashlib
    from concurrent.futures import ThreadPoolExecutor
    from . import index as pidx
    engine = pidx.engine
    k = index.k
    ctx = index.context
    geo = index.result_geometry
    anchors = list(index.anchor_genomes)
    # when this rank's anchoring began: what logs/anchor.<name>.benchmark.txt counts from for a genome assembled out of
    # pieces (its log is only opened at assembly time; Genome.write_benchmark)
    import time as _time
    anchoring_t0 = _time.perf_counter()
    gid = {n: i for i, n in enumerate(anchors)}
    seqs = {name: index.seqset_for(name) for name in anchors}
    classes = engine.homology_classes([seqs[n].names for n in anchors])
    contigs, c = [], 0
    for name in anchors:
        for ci, ln in enumerate(seqs[name].lens):
            contigs.append((name, ci, max(0, int(ln) - k + 1), int(classes[c])))
            c += 1
    ppr = (int(os.environ["PG_PIECES_PER_RANK"]) if "PG_PIECES_PER_RANK" in os.environ else None) if pieces_per_rank is None else pieces_per_rank
    plan = plan_class_pieces(contigs, world, geo["lowres_step"], ppr)
    sig = _plan_signature(index, world, hashlib.sha1(repr(plan).encode()).hexdigest())
    nk_of = {(name, ci): nk for name, ci, nk, _ in contigs}
    pieces_of: Dict[str, List[Piece]] = {n: [] for n in anchors}
    for sh in plan:
        for p in sh:
            pieces_of[p[0]].append(p)
    for n in anchors:
        pieces_of[n].sort(key=lambda p: (p[1], p[2]))

    def contig_binlen(nk: int) -> int:  # cpp/anchor.cpp:114-118 / index.py:1169-1172 (result_create applies the same rule)
        bl = geo["max_bin_len"]
        if nk // bl < geo["min_bin_count"]:
            bl = nk // geo["min_bin_count"]
        return max(1, bl)

    def base(p: Piece) -> str:
        return os.path.join(_parts_dir(index.genomes[p[0]]), f"{p[1]}.{p[2]}")

    mine = plan[rank] if rank < len(plan) else []
    if mine:
        for p in mine:  # (what this rank is about to write again must not count as done meanwhile)
            try:
                os.remove(base(p) + ".npz")
            except FileNotFoundError:
                pass
        # this rank's pieces of every genome, cut out of the packed sequences in HBM (k - 1 bases of overlap)
        order = sorted(mine, key=lambda p: (p[4], p[5], gid[p[0]]))
        own = {n: [p for p in order if p[0] == n] for n in anchors}
        cut = {n: seqs[n].slice([(p[1], p[2], p[3] + k - 1) for p in ps]) for n, ps in own.items() if ps}
        table = index.build_table(insert_sets=cut)
        nb = (index.ngenomes + 7) // 8
        genes = {n: index.genomes[n].load_genes() if index.genomes[n].annotated else None for n in cut}

        # batches of class pieces whose rows fit next to the table (the same bound as Index.run()'s batches)
        free = ctx.mem_info()[0]
        batch_bytes = int(min(index.batch_bytes, max(1 << 30, (free - (6 << 30)) / 2.3)))
        batches, cur, cur_bytes = [], [], 0
        import itertools
        for _key, grp_it in itertools.groupby(order, key=lambda p: (p[4], p[5])):  # (order is sorted by (class, piece, genome))
            grp = list(grp_it)
            b = sum(p[3] for p in grp) * nb
            if cur and cur_bytes + b > batch_bytes:
                batches.append(cur)
                cur, cur_bytes = [], 0
            cur += grp
            cur_bytes += b
        if cur:
            batches.append(cur)
        payload = sum(p[3] for p in order) * nb

        def piece_stats(res, i, p):
            """bins (over the CONTIG's bin windows, clipped to the piece), first bin and gene histogram of piece p = contig i of res"""
            nk, bl = nk_of[(p[0], p[1])], contig_binlen(nk_of[(p[0], p[1])])
            bin0 = p[2] // bl
            bins = None
            if p[3] != nk:
                # a piece of a longer contig: the statistics over the CONTIG's bins (cpp/anchor.cpp:114-120,179-189)
                # clipped to the piece — a bin that two pieces share is added up when the genome is assembled
                edges = np.arange(bin0, (p[2] + p[3] - 1) // bl + 2, dtype=np.int64) * bl
                lo, hi = np.maximum(edges[:-1], p[2]) - p[2], np.minimum(edges[1:], p[2] + p[3]) - p[2]
                h, _ = res.window_stats(i, lo, hi, step=1, colsums=False)
                bins = h.astype(np.int64)
            ghist = np.zeros(index.ngenomes + 1, np.int64)
            gt = genes.get(p[0])
            if gt is not None:  # the piece's share of its chromosome's gene occupancy (index.py:1055-1064): genes clipped to it
                sel = gt[gt["chr"] == seqs[p[0]].names[p[1]]]
                st, en = sel["start"].to_numpy(np.int64), sel["end"].to_numpy(np.int64)
                ok = (en > st) & (st >= 0) & (en <= nk)
                a, b = np.maximum(st[ok], p[2]) - p[2], np.minimum(en[ok], p[2] + p[3]) - p[2]
                hit = b > a
                if hit.any():
                    h, _ = res.window_stats(i, a[hit], b[hit], step=1, colsums=False)
                    ghist = h.sum(axis=0).astype(np.int64)
            return bins, bin0, ghist

        def write_unit(res, i0, ps):
            """One fragment for a run of this rank's pieces that follow one another in their genome's files — whole
            neighbouring contigs (a bundle of small classes), or a single piece: the rows' BGZF fragments, then the
            marker with the pieces' bins, sums and gene histograms."""
            g = index.genomes[ps[0][0]]
            os.makedirs(_parts_dir(g), exist_ok=True)
            m = len(ps)
            for s_ in index.steps:
                # (under a temporary name first: a rank that resumes over an aborted run's fragments of the same signature
                # rewrites bytes another rank may be reading — a reader sees the old file or the new one, never half of one)
                gz_, gzi_ = f"{base(ps[0])}.{s_}.gz", f"{base(ps[0])}.{s_}.gzi"
                res.write_bgzf(s_, gz_ + ".tmp", gzi_ + ".tmp", level=index.bgzf_level, threads=2, first_contig=i0, ncontigs=m)
                os.replace(gz_ + ".tmp", gz_)
                os.replace(gzi_ + ".tmp", gzi_)
            small = res.contigs_small(i0, m)
            if [int(x) for x in small.nkmers] != [p[3] for p in ps]:
                raise RuntimeError(f"{ps[0][0]}: the result's contigs {[int(x) for x in small.nkmers]} are not the plan's pieces "
                                   f"{[p[3] for p in ps]} (contig {ps[0][1]} from {ps[0][2]})")
            bins_l, bin0_l, gh_l = [], [], []
            for q, p in enumerate(ps):
                b, b0, gh = piece_stats(res, i0 + q, p) if (m == 1 or genes.get(p[0]) is not None) else (None, 0, np.zeros(index.ngenomes + 1, np.int64))
                bins_l.append(small[q][2] if b is None else b.astype(np.uint32))  # (counts of a contig's bin: 32 bits, as on the device)
                bin0_l.append(b0)
                gh_l.append(gh)
            np.savez(base(ps[0]) + ".tmp.npz", ci=np.array([p[1] for p in ps], np.int64), start=np.array([p[2] for p in ps], np.int64),
                     nkmers=np.asarray(small.nkmers, np.int64), nrows100=np.asarray(small.nrows100, np.int64),
                     nbins=np.array([len(b) for b in bins_l], np.int64), bin0=np.array(bin0_l, np.int64),
                     bins=np.vstack(bins_l).astype(np.uint32) if bins_l else np.zeros((0, index.ngenomes + 1), np.uint32),
                     colsums=res.contig_colsums(i0, m).astype(np.int64).sum(axis=0), gene_hist=np.vstack(gh_l), sig=sig)
            os.replace(base(ps[0]) + ".tmp.npz", base(ps[0]) + ".npz")  # written last: the unit's completion marker

        def units_of(layout):
            """runs of the batch's pieces that one fragment can hold: neighbours in the batch AND in their genome's files
            (whole contigs ci, ci + 1, ...)"""
            out, i = [], 0
            while i < len(layout):
                p, m = layout[i], 1
                whole = lambda x: x[2] == 0 and x[3] == nk_of[(x[0], x[1])]  # noqa: E731
                while (i + m < len(layout) and whole(layout[i + m - 1]) and whole(layout[i + m]) and layout[i + m][0] == p[0]
                       and layout[i + m][1] == layout[i + m - 1][1] + 1):
                    m += 1
                out.append((i, layout[i:i + m]))
                i += m
            return out

        with ThreadPoolExecutor(max_workers=index.writer_jobs(payload)) as pool:
            previous = None
            for batch in batches:
                # the batch's pieces side by side, genome after genome; co-scheduled by (class, piece)
                per_genome = [[p for p in batch if p[0] == n] for n in anchors]
                parts, layout = [], []
                for n, ps in zip(anchors, per_genome):
                    if ps:
                        first = own[n].index(ps[0])
                        if own[n][first:first + len(ps)] != ps:
                            raise RuntimeError(f"{n}: a batch's pieces are not a run of this rank's slices (piece {ps[0][1:4]})")
                        parts.append((cut[n], first, len(ps)))
                        layout += ps
                merged = engine.SeqSet.concat_ranges(ctx, parts)
                res = engine.AnchorResult(table, merged, colsums=True, **geo)
                if len({p[0] for p in layout}) > 1:
                    cls_ids = {key: i for i, key in enumerate(sorted({(p[4], p[5]) for p in layout}))}
                    res.coschedule(np.array([gid[p[0]] for p in layout], np.uint32),
                                   contig_class=np.array([cls_ids[(p[4], p[5])] for p in layout], np.uint32))
                res.run()
                futs = [pool.submit(write_unit, res, i0, ps) for i0, ps in units_of(layout)]
                if previous is not None:  # at most two batches of rows resident
                    _finish(*previous)
                previous = (res, merged, futs)
            if previous is not None:
                _finish(*previous)
        for ss in cut.values():
            ss.close()

    def try_assemble(name: str) -> bool:
        g = index.genomes[name]
        ps = pieces_of[name]
        pdir = _parts_dir(g)
        if not ps:
            # a genome without a single k-mer position (an empty FASTA, contigs all shorter than k) has no piece and so
            # no marker anybody could find complete: ONE rank — by the genome's number, no claim needed — writes its
            # (empty) bitmaps and tables, as the one-rank run does
            if gid[name] % world == rank:
                os.makedirs(pdir, exist_ok=True)
                return assemble_under_claim(name, g, ps, pdir, None, lambda full=True: [])
            return os.path.exists(g.chrs_fname)
        if not os.path.isdir(pdir):
            return os.path.exists(g.chrs_fname)  # assembled (and cleaned up) by another rank
        def load_markers(full=True):
            """the markers in the genome's .parts directory that carry this run's signature, if together they cover every
            piece of the genome exactly once: [(first piece's base path, marker fields)] in file order — else None.
            ``full`` off: only the fields that say what a marker covers (the bins of a fragmented genome are megabytes)"""
            want = {(p[1], p[2]): p for p in ps}
            found, seen = [], set()
            try:
                files = sorted(f for f in os.listdir(pdir) if f.endswith(".npz") and not f.endswith(".tmp.npz"))
            except OSError:
                return None
            for f in files:
                try:
                    with np.load(os.path.join(pdir, f)) as z:
                        if str(z["sig"]) != sig:
                            continue
                        d = {x: z[x] for x in (z.files if full else ("ci", "start", "nkmers"))}
                except (FileNotFoundError, OSError, KeyError, ValueError):
                    return None
                keys = list(zip(d["ci"].tolist(), d["start"].tolist()))
                if any(k_ not in want or k_ in seen for k_ in keys) or any(int(n_) != want[k_][3] for k_, n_ in zip(keys, d["nkmers"])):
                    return None
                seen.update(keys)
                found.append((os.path.join(pdir, f[:-4]), keys, d))
            if len(seen) != len(want):
                return None
            found.sort(key=lambda u: u[1][0])
            return found

        if load_markers(full=False) is None:
            return False
        os.makedirs(pdir, exist_ok=True)
        lock = os.path.join(pdir, "assemble.lock")
        claim = _claim(lock)
        if claim is None:
            return False
        try:
            return assemble_under_claim(name, g, ps, pdir, lock, load_markers)
        finally:
            os.close(claim)

    def assemble_under_claim(name, g, ps, pdir, lock, load_markers) -> bool:
        # (whoever assembled meanwhile removed the markers BEFORE giving up its claim: seeing them all under our own
        # claim means the genome is ours to assemble)
        metas = load_markers()
        if metas is None:
            _remove_quietly(lock)
            try:
                os.rmdir(pdir)
            except OSError:
                pass
            return os.path.exists(g.chrs_fname)
        g.ensure_log(started=anchoring_t0)
        os.makedirs(g.prefix, exist_ok=True)
        names = list(seqs[name].names)
        for s_ in index.steps:
            concat_bgzf([(f"{b_}.{s_}.gz", f"{b_}.{s_}.gzi") for b_, _, _ in metas], g.bitmap_gz_fname(s_), g.bitmap_gzi_fname(s_))
        # the genome's bins in one array (contig after contig), every unit's rows added at its place
        N1 = index.ngenomes + 1
        nks = np.array([nk_of[(name, ci)] for ci in range(len(names))], np.int64)
        bls = np.array([contig_binlen(int(nk)) for nk in nks], np.int64)
        nbins = (nks + bls - 1) // bls
        bin_off = np.concatenate([[0], np.cumsum(nbins)])
        bins_all = np.zeros((int(bin_off[-1]), N1), np.int64)
        nrows100 = np.zeros(len(names), np.int64)
        covered = np.zeros(len(names), np.int64)
        gene_sum = {}
        cs = np.zeros(index.ngenomes, np.int64)
        for _, keys, z in metas:
            cs += z["colsums"]
            off = np.concatenate([[0], np.cumsum(z["nbins"])])
            whole_run = len(keys) > 1  # (whole neighbouring contigs: the unit's rows are the genome's, one block)
            if whole_run:
                a = int(bin_off[keys[0][0]])
                if int(off[-1]) != int(bin_off[keys[-1][0] + 1]) - a:
                    raise RuntimeError(f"{name}: the fragment of contigs {keys[0][0]}..{keys[-1][0]} holds {int(off[-1])} bins, "
                                       f"the genome's geometry {int(bin_off[keys[-1][0] + 1]) - a} (a marker of another geometry?)")
                bins_all[a:a + int(off[-1])] += z["bins"]
            for q, (ci, _st) in enumerate(keys):
                if not whole_run:
                    a = int(bin_off[ci]) + int(z["bin0"][q])
                    bins_all[a:a + int(z["nbins"][q])] += z["bins"][int(off[q]):int(off[q + 1])]
                nrows100[ci] += int(z["nrows100"][q])
                covered[ci] += int(z["nkmers"][q])
                if g.annotated:
                    gene_sum[names[ci]] = gene_sum.get(names[ci], 0) + z["gene_hist"][q]
        if not np.array_equal(covered, nks):
            bad = int(np.flatnonzero(covered != nks)[0])
            raise RuntimeError(f"{name}: the fragments cover {int(covered[bad])} of contig {bad}'s {int(nks[bad])} k-mer positions")
        bins_infos = pidx.engine.SmallOutputs(nks.astype(np.uint64), nrows100.astype(np.uint64), nbins.astype(np.uint32), bls.astype(np.uint32),
                                              bins_all.astype(np.uint32))
        gene_hists = None
        if g.annotated:
            gene_hists = {}
            for chrom, grp in g.load_genes().groupby("chr", sort=True):
                if chrom in names:
                    size = nk_of[(name, names.index(chrom))]
                    st, en = grp["start"].to_numpy(np.int64), grp["end"].to_numpy(np.int64)
                    for s_, e_ in zip(st[~((en > st) & (st >= 0) & (en <= size))], en[~((en > st) & (st >= 0) & (en <= size))]):
                        g.log.warning(f"Skipping gene at {chrom}:{s_}-{e_}, coordinates out-of-bounds")
                gene_hists[chrom] = (len(grp), np.asarray(gene_sum.get(chrom, np.zeros(index.ngenomes + 1, np.int64)), np.int64))
        g._write_tables(names, bins_infos, cs, gene_hists)
        g.close_log()
        # markers first: from here on nobody takes the genome for complete-and-unassembled.  (A marker may be gone already:
        # its rank resumed over an aborted run's fragments and removed it to write that unit again — same bytes.)
        for b_, _, _ in metas:
            _remove_quietly(b_ + ".npz")
        # then what this assembly consumed — and nothing else: a rank still rewriting a stale unit keeps its temporary
        # files and its directory, and clears them itself once it is through (leftovers(), below)
        for b_, _, _ in metas:
            for s_ in index.steps:
                _remove_quietly(f"{b_}.{s_}.gz")
                _remove_quietly(f"{b_}.{s_}.gzi")
        # and what an aborted run of ANOTHER signature (other world size, plan or parameters: other base names) left here —
        # nobody will ever read it, and it is GBs.  Under the claim, behind a complete assembly, every finished file of the
        # directory is garbage: this run's units were all consumed above, and a rank that is rewriting a unit it took for
        # stale works under *.tmp names until its rename (those, and only those, are left alone; it clears up behind itself).
        _clear_stale_fragments(pdir)
        if lock is not None:
            _remove_quietly(lock)  # (the claim goes last; a rank that finds the directory again finds no marker)
        try:
            os.rmdir(pdir)
        except OSError:
            try:
                left = sorted(os.listdir(pdir))
            except OSError:
                left = []
            if left:
                logger_info("%s: %d file(s) left in %s (another rank's temporary files: %s ...)", name, len(left), pdir, left[:3])
        return True

    def leftovers(name: str) -> None:
        """this rank's units of a genome that was assembled over an aborted run's fragments while this rank was still
        writing them again: nobody will read them"""
        pdir = _parts_dir(index.genomes[name])
        if not os.path.isdir(pdir):
            return
        for p in mine:
            if p[0] == name:
                for suffix in [".npz", ".tmp.npz"] + [f".{s_}.{e}" for s_ in index.steps for e in ("gz", "gzi", "gz.tmp", "gzi.tmp")]:
                    _remove_quietly(base(p) + suffix)
        try:
            os.rmdir(pdir)
        except OSError:
            pass

    todo = [n for n in anchors]
    for phase in range(2):
        todo = [n for n in todo if not try_assemble(n)]
        if barrier is None or phase == 1:
            break
        barrier()  # every rank's markers are out: whatever is still unassembled is complete now
    if barrier is not None:
        # no rank leaves (and tears the process group down, or reads the outputs) while another is still concatenating a
        # genome it holds the claim of; after this barrier every genome must be there
        barrier()
        missing = [n for n in anchors if not os.path.exists(index.genomes[n].chrs_fname)]
        if missing:
            raise RuntimeError(f"rank {rank}: no rank assembled {missing} (fragments incomplete or of another run: see "
                               f"{[_parts_dir(index.genomes[n]) for n in missing]})")
        for n in anchors:
            leftovers(n)
    # (without a rendezvous — ranks started on their own, possibly one after the other — a rank cannot tell "another rank
    # will assemble this" from "nobody will": the rank that finishes last finds every genome complete and assembles it)


def _clear_stale_fragments(pdir: str) -> int:
    """Remove every finished fragment, index and marker of a genome's .parts directory — called under the assembly claim
    once the genome's files are complete, when nothing in there can be needed again (see the call site).  Temporary files
    (``*.tmp``, ``*.tmp.npz``: a live rank's unit in the making) and the claim file are not touched.  Returns the number
    of files removed."""
    n = 0
    try:
        files = os.listdir(pdir)
    except OSError:
        return 0
    for f in files:
        if f == "assemble.lock" or f.endswith(".tmp") or f.endswith(".tmp.npz"):
            continue
        if f.endswith((".npz", ".gz", ".gzi")):
            _remove_quietly(os.path.join(pdir, f))
            n += 1
    return n


def _remove_quietly(path: str) -> None:
    try:
        os.remove(path)
    except FileNotFoundError:
        pass


def _finish(res, merged, futs):
    try:
        for f in futs:
            f.result()
    finally:
        res.close()
        merged.close()


# ---------------------------------------------------------------------------
# genome-sharded combine step
# ---------------------------------------------------------------------------
def combine_rows_(rows, group=None) -> None:
    """In-place combine of per-rank partial rows (uint8 tensor, any device).  Ranks own
    disjoint genome bits, so SUM == OR; one all-reduce over RCCL/xGMI (or gloo on CPU)."""
    import torch.distributed as dist
    dist.all_reduce(rows, op=dist.ReduceOp.SUM, group=group)


def combine_rows_allgather(rows, group=None):
    """Reference formulation: all-gather every rank's partial rows, OR them locally."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    parts = [torch.empty_like(rows) for _ in range(world)]
    dist.all_gather(parts, rows, group=group)
    out = parts[0].clone()
    for p in parts[1:]:
        out |= p
    return out


def genomes_per_rank(ngenomes: int, world: int) -> int:
    return (ngenomes + world - 1) // world


def gather_columns(mine, group=None):
    """All-gather of the ranks' compact bit-column blocks (equal-sized uint8 tensors, any device):
    the one collective of the genome-sharded mode — RCCL over xGMI on GPUs, gloo on CPU.  Returns the
    concatenation, block i from rank i."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    out = torch.empty(world * mine.numel(), dtype=mine.dtype, device=mine.device)
    dist.all_gather_into_tensor(out, mine.contiguous(), group=group)
    return out


def exchange_columns_(res, ngenomes: int, rank: int, world: int, group=None) -> None:
    """Complete this rank's partial rows (only its own genomes' bits are set) in place: extract the
    compact columns of its genomes, all-gather every rank's block, merge them into the rows.
    Moves (world-1)/world row bytes per position per rank — half of what the SUM all-reduce of
    ``combine_rows_`` moves."""
    import torch
    per = genomes_per_rank(ngenomes, world)
    dev = torch.device("cuda", res.table.ctx.device)
    mine = torch.empty(res.columns_bytes(per), dtype=torch.uint8, device=dev)
    res.extract_columns(rank * per, per, mine.data_ptr())
    res.table.ctx.synchronize()
    allb = gather_columns(mine, group)
    res.merge_columns(allb.data_ptr(), world, per)
    res.table.ctx.synchronize()  # allb may be freed by torch once we return


def build_partial_table(ctx, k: int, ngenomes: int, rank: int, world: int, genome_seqs):
    """Table of the genomes this rank owns (full-width rows).  ``genome_seqs(g)`` -> list of
    contig byte strings of genome g, or None to skip."""
    from . import engine
    tbl = engine.PanTable(ctx, k, ngenomes)
    for g in range(ngenomes):
        if genome_owner(g, ngenomes, world) != rank:
            continue
        ss = engine.SeqSet.from_host(ctx, genome_seqs(g))
        tbl.insert_seqset(g, ss)
        ss.close()
    return tbl


def anchor_genome_sharded(table, seqs: Sequence[bytes], group=None, rank: Optional[int] = None,
                          world: Optional[int] = None, exchange: str = "columns"):
    """Anchor contigs against this rank's partial table, complete the rows across ranks, derive
    bitmap.100 / bins / column sums from the completed rows.  Returns like Genome.anchor_contigs.
    ``exchange``: "columns" = all-gather of compact bit columns (default; ranks must own the
    contiguous genome blocks of ``genome_owner``), "allreduce" = SUM all-reduce of the rows."""
    from . import engine
    ss = engine.SeqSet.from_host(table.ctx, seqs)
    res = engine.AnchorResult(table, ss, colsums=True, rows_only=True)
    res.run()
    table.ctx.synchronize()
    if exchange == "columns":
        import torch.distributed as dist
        rank = dist.get_rank(group) if rank is None else rank
        world = dist.get_world_size(group) if world is None else world
        exchange_columns_(res, table.ngenomes, rank, world, group)
    else:
        combine_rows_(res.rows_tensor(), group)
    res.rows_epilogue()
    out = [res.download(ci) for ci in range(len(seqs))]
    cs = res.colsums().astype(np.int64)
    res.close()
    ss.close()
    return out, cs


# ---------------------------------------------------------------------------
# genome-sharded index build: the product path for pangenomes whose table exceeds one GPU
# ---------------------------------------------------------------------------
CHUNK_POSITIONS = int(os.environ.get("PG_SHARD_CHUNK", str(1 << 27)))  # positions per exchanged chunk (about)


class _Pipe:
    """Stream plumbing of the chunk pipeline.  With a collective on a GPU: the library's kernels run on ``main`` (the
    context is pointed at it), the collectives on ``comm``; events order them, the host never waits inside the loop.
    Without a collective (one process) everything runs in order on the context's own stream and torch is not even
    imported; on CPU tensors (the gloo tests) everything is synchronous."""

    def __init__(self, ctx, dev):
        self.ctx = ctx
        self.gpu = dev is not None and dev.type == "cuda"
        if self.gpu:
            import torch
            self.torch = torch
            self.main = torch.cuda.Stream(dev)
            self.comm = torch.cuda.Stream(dev)
            ctx.set_stream(self.main.cuda_stream)

    def mark_main(self):
        if not self.gpu:
            return None
        ev = self.torch.cuda.Event()
        ev.record(self.main)
        return ev

    def main_waits(self, ev):
        if self.gpu and ev is not None:
            self.main.wait_event(ev)

    def on_comm(self, after, fn):
        """run ``fn`` (a collective) on the side stream once ``after`` has happened; returns its completion event"""
        if not self.gpu:
            fn()
            return None
        with self.torch.cuda.stream(self.comm):
            self.comm.wait_event(after)
            fn()
            ev = self.torch.cuda.Event()
            ev.record(self.comm)
        return ev

    def close(self):
        if self.gpu:
            self.ctx.synchronize()
            self.torch.cuda.synchronize()
            self.ctx.set_stream(None)


class _TorchBuffer:
    """exchange buffer owned by torch (the collective takes tensors); same face as engine.DeviceBuffer"""

    def __init__(self, torch, nbytes, dev, stream=None):
        self.t = torch.zeros(max(int(nbytes), 8), dtype=torch.uint8, device=dev)
        self._torch, self._stream = torch, stream

    def data_ptr(self):
        return self.t.data_ptr()

    def zero(self, nbytes=None):
        v = self.t if nbytes is None else self.t[:nbytes]
        if self._stream is not None:
            with self._torch.cuda.stream(self._stream):
                v.zero_()
        else:
            v.zero_()

    def close(self):
        self.t = None


def contig_chunks(lens: Sequence[int], k: int, limit: Optional[int] = None) -> List[Tuple[int, int]]:
    """Contiguous contig ranges ``(first, count)`` of about ``limit`` (default CHUNK_POSITIONS) k-mer positions each
    (a contig is never cut)."""
    limit = CHUNK_POSITIONS if limit is None else limit
    out, first, acc = [], 0, 0
    for ci, ln in enumerate(lens):
        nk = max(0, int(ln) - k + 1)
        if ci > first and acc + nk > limit:
            out.append((first, ci - first))
            first, acc = ci, 0
        acc += nk
    if len(lens) > first:
        out.append((first, len(lens) - first))
    return out


class ShardedAnchoring:
    """The chunk pipeline of the genome-sharded mode for ONE process (rank): persistent exchange buffers, one
    narrow result (rows of the block's genomes only) over ALL anchors' sequences, the writers' full-row containers.

        pipe = ShardedAnchoring(engine, ctx, k, N, per, rank, world, seqs, writer, geometry, group)
        pipe.run_pass(table_of_my_block, part0, nparts, accumulate, on_anchor_complete)

    ``seqs``: anchor name -> SeqSet (every rank holds every anchor's sequence); ``writer``: anchor name -> rank
    that assembles its rows.  Every anchor's contigs are cut into chunks of about CHUNK_POSITIONS positions; chunk
    GROUP i = the i-th chunk of every anchor — homologous stretches of the pangenome, probed in ONE co-scheduled
    launch so that the anchors share their table lines in L2 exactly as in the replicated mode (the anchors'
    sequences are laid out group by group in one merged seqset for this).  ``run_pass`` probes every group against
    ``table`` (None: this rank has no block in the pass and contributes zeros):

        probe group i  ->  extract its bit columns  ->  exchange on the side stream  ->  merge on the writers
                           probe group i+1 ...

    and calls ``on_anchor_complete(name, rows_container)`` on the writer once an anchor's last chunk is merged.

    The exchange (round 6): only an anchor's WRITER merges its columns, so a rank sends each anchor's columns to that
    writer alone — an all-to-all with split sizes (RCCL ``all_to_all_single``; batched ``isend`` / ``irecv`` where the
    backend has no all-to-all: gloo) — instead of all-gathering every block's columns to every rank: a rank receives
    (world - 1) / world of ITS OWN anchors' columns, 1 / world of what the all-gather delivered (config 5 on 8 GPUs: 2.6 GB
    per GPU instead of 21 GB).  A group's members are laid out writer by writer for this, so that what goes to one rank is
    one contiguous piece of the send buffer.  ``PG_SHARD_EXCHANGE=allgather`` keeps the all-gather (the north star's
    wording; the same bytes on disk), ``host`` / ``host-allgather`` take either through pinned host memory and gloo."""

    # widest block whose columns the probe assembles itself (a ballot + two LDS words per genome and batch): one genome
    # per block — config 5's layout — where it also spares the narrow row buffer (one byte per anchor position: 24 GB
    # at config 5); wider blocks are faster through rows + k_cols_extract (measured per rank on the configs[1]
    # pangenome, G k-mers/s: 1 genome per block 123.5 direct vs 121.6; 2: 114.9 vs 118.9; 8: 82 vs 109)
    DIRECT_MAX_WIDTH = int(os.environ.get("PG_DIRECT_MAX_WIDTH", "1"))

    def __init__(self, engine, ctx, k: int, ngenomes: int, per: int, rank: int, world: int, seqs: Dict[str, object],
                 writer: Dict[str, int], geometry: Optional[dict] = None, group=None, always_gather: bool = False,
                 direct_columns: Optional[bool] = None):
        """``always_gather``: issue the collective even with one rank (a process group of size 1) — the side-stream
        and RCCL code path on a single GPU.  ``direct_columns``: whether the probe emits the block's bit columns itself where
        it can (blocks of up to DIRECT_MAX_WIDTH genomes; None: engine.COLUMNS_DIRECT, i.e. PG_COLUMNS_DIRECT) — no narrow row
        buffer then, one byte per anchor position saved; the planner asks for it when a block table only fits dense."""
        self.engine, self.ctx, self.k, self.N, self.per = engine, ctx, k, ngenomes, per
        self.rank, self.world, self.seqs, self.writer, self.group = rank, max(1, world), seqs, writer, group
        self.geometry = geometry or {}
        self.direct_columns = direct_columns
        self.dist = None
        self.collective = self.world > 1 or always_gather
        # how the blocks' bit columns travel: "rccl" — all_gather_into_tensor of device buffers on the default process
        # group (RCCL over xGMI) — or "host" (SURVEY §8e's fallback without a GPU collective): every rank copies its
        # columns to pinned host memory, the hosts all-gather them over a gloo group, the gathered blocks go back to
        # the GPU and are merged as usual.  1 bit per genome and position: the detour costs PCIe time, not correctness.
        # (round 6) "rccl" and "host" send an anchor's columns to its writer only; "allgather" / "host-allgather": to every rank
        mode = os.environ.get("PG_SHARD_EXCHANGE", "rccl").lower()
        if mode not in ("rccl", "host", "allgather", "host-allgather"):
            raise ValueError(f"PG_SHARD_EXCHANGE must be 'rccl', 'host', 'allgather' or 'host-allgather', got {mode!r}")
        self.exchange = "host" if mode.startswith("host") else "rccl"
        self.to_writers = not mode.endswith("allgather")
        self.host_group = None
        if self.collective:
            import torch.distributed as dist
            if not dist.is_initialized():
                raise RuntimeError("genome-sharded mode on several ranks needs torch.distributed initialised, e.g.\n"
                                   "  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 "
                                   "--master-port 29500 -m panagram_amd index samples.tsv -o out")
            self.dist = dist
            if dist.get_world_size(group) != self.world or dist.get_rank(group) != self.rank:
                raise RuntimeError(f"process group says rank {dist.get_rank(group)} of {dist.get_world_size(group)}, the index "
                                   f"was opened as rank {self.rank} of {self.world} (RANK / WORLD_SIZE)")
            if self.exchange == "host" and dist.get_backend(group) != "gloo":
                self.host_group = dist.new_group(backend="gloo")  # (collective call: every rank is here)
        tile = engine.tile_positions()
        self.col_bytes = tile // 8  # bytes of one genome's bit column per tile (a u64 per 64 positions)
        names = list(seqs)
        chunks = {a: contig_chunks(seqs[a].lens, k) for a in names}
        # groups[i] = [(anchor, first contig, count, tile offset inside the group)], the same on every rank
        # (members writer by writer, a writer's anchors in their own order: what goes to one rank is ONE contiguous piece of
        # the group's column buffer — dest_off / dest_tiles [i][w], in tiles)
        self.groups, parts, self.group_first, self.group_tiles = [], [], [], []
        self.dest_off, self.dest_tiles = [], []
        contig_anchor = []
        by_writer = sorted(range(len(names)), key=lambda ai: (int(writer.get(names[ai], 0)), ai))
        for i in range(max([len(c) for c in chunks.values()] or [0])):
            members, toff = [], 0
            self.group_first.append(len(contig_anchor))
            doff, dtiles = [0] * self.world, [0] * self.world
            for ai in by_writer:
                a = names[ai]
                if i < len(chunks[a]):
                    c0, nc = chunks[a][i]
                    w = int(writer.get(a, 0))
                    if not 0 <= w < self.world:
                        raise ValueError(f"anchor {a!r} has writer rank {w}, the run has {self.world} rank(s)")
                    if dtiles[w] == 0:
                        doff[w] = toff
                    members.append((a, c0, nc, toff))
                    nt = sum((max(0, int(ln) - k + 1) + tile - 1) // tile for ln in seqs[a].lens[c0:c0 + nc])
                    toff += nt
                    dtiles[w] += nt
                    parts.append((seqs[a], c0, nc))
                    contig_anchor += [ai] * nc
            # (a writer without an anchor in this group — its anchors are shorter — gets an empty piece at its place in the order:
            # all_to_all_single's input splits are consecutive)
            doff = [sum(dtiles[:w]) for w in range(self.world)]
            self.groups.append(members)
            self.group_tiles.append(toff)
            self.dest_off.append(doff)
            self.dest_tiles.append(dtiles)
        self.last_group = {a: max(i for i, m in enumerate(self.groups) if any(x[0] == a for x in m))
                           for a in names if chunks[a]}
        self.merged = engine.SeqSet.concat_ranges(ctx, parts) if parts else None
        self._contig_anchor = np.asarray(contig_anchor, np.uint32)
        biggest = max(self.group_tiles or [0]) * self.col_bytes * per
        # what this rank receives per group: `world` blocks of its OWN anchors' columns (to_writers), or of everybody's
        own_biggest = max([d[self.rank] for d in self.dest_tiles] or [0]) * self.col_bytes * per if self.to_writers else biggest
        if self.collective:  # torch owns the buffers (the collective takes tensors) and the two streams
            import torch
            dev = ctx.torch_device()
            self.send = [_TorchBuffer(torch, biggest, dev) for _ in range(2)]
            self.recv = [_TorchBuffer(torch, own_biggest * self.world, dev) for _ in range(2)]
            if dev.type == "cuda":
                torch.cuda.synchronize(dev)  # the buffers were zeroed on torch's stream; the kernels run on the pipe's
            self.pipe = _Pipe(ctx, dev)
            for b_ in self.send:
                b_._stream = getattr(self.pipe, "main", None)
            self._host = None
            if self.exchange == "host" and dev.type == "cuda":
                self._host = (torch.empty(max(biggest, 8), dtype=torch.uint8).pin_memory(),
                              torch.empty(max(own_biggest, 8) * self.world, dtype=torch.uint8).pin_memory())
            self._first_contact(torch, dev)
        else:  # one process: its own block is all there is — merged straight out of the send buffer; no torch
            self.send = [engine.DeviceBuffer(ctx, biggest) for _ in range(2)]
            self.recv = self.send
            self.pipe = _Pipe(ctx, None)
        self.full: Dict[str, object] = {}  # writer side: anchor -> rows container (kept across passes)
        self._part, self._part_table = None, None  # the narrow result, kept while the table stays the same
        self.bytes_received = 0

    def _first_contact(self, torch, dev) -> None:
        """One tiny all-gather before any work is queued, with a deadline: a collective that cannot complete (ranks on
        one GPU, a dead xGMI link, a rank that never arrived) is reported here, with what to try next, instead of as a
        hang in the middle of the first pass.  Also checks that every rank answers with its own number."""
        import threading
        timeout = float(os.environ.get("PG_COLLECTIVE_TIMEOUT_S", "180"))
        if dev.type == "cuda" and self.exchange == "rccl" and self.world > 1 and self.dist.get_backend(self.group) == "nccl":
            seen = [None] * self.world
            self.dist.all_gather_object(seen, (os.uname().nodename, str(getattr(torch.cuda.get_device_properties(dev), "uuid", dev.index))),
                                        group=self.group)
            if len(set(seen)) != self.world:
                raise RuntimeError(f"{self.world} ranks on {len(set(seen))} distinct GPU(s): RCCL needs one process per GPU "
                                   "(LOCAL_RANK picks the device); PG_SHARD_EXCHANGE=host runs the exchange through the hosts instead")
            n = torch.cuda.device_count()
            blocked = [(dev.index, j) for j in range(n) if j != dev.index and not torch.cuda.can_device_access_peer(dev.index, j)]
            if blocked:
                logger_info("no peer access from GPU %d to %s: RCCL will not use xGMI there", dev.index, [j for _, j in blocked])
        box = {}

        def probe():
            try:
                got = self._gather(torch.full((8,), self.rank, dtype=torch.uint8, device=dev), dev)
                if dev.type == "cuda":
                    torch.cuda.synchronize(dev)
                box["got"] = got.cpu().view(self.world, 8)[:, 0].tolist()
            except Exception as e:  # noqa: BLE001 — reported below
                box["err"] = e
        t = threading.Thread(target=probe, daemon=True)
        t.start()
        t.join(timeout)
        how = "RCCL collective (all_gather_into_tensor)" if self.exchange == "rccl" else "host all-gather (gloo)"
        if t.is_alive():
            raise RuntimeError(f"the first {how} of {self.world} ranks did not complete within {timeout:.0f} s "
                               "(PG_COLLECTIVE_TIMEOUT_S).  Check that every rank was started (WORLD_SIZE), one per GPU; "
                               + ("PG_SHARD_EXCHANGE=host takes the exchange off RCCL/xGMI." if self.exchange == "rccl" else ""))
        if "err" in box:
            raise RuntimeError(f"the first {how} failed: {box['err']}"
                               + ("; PG_SHARD_EXCHANGE=host takes the exchange off RCCL/xGMI" if self.exchange == "rccl" else "")) from box["err"]
        if box["got"] != list(range(self.world)):
            raise RuntimeError(f"the first {how} returned blocks from ranks {box['got']}, expected 0..{self.world - 1}")

    def _gather(self, mine, dev):
        """all-gather of equal-sized uint8 blocks ``mine`` -> the concatenation, block i from rank i (the one exchange
        step of the mode), by the configured route"""
        import torch
        out = torch.empty(mine.numel() * self.world, dtype=torch.uint8, device=mine.device)
        self._gather_into(out, mine)
        return out

    def _gather_into(self, out_t, in_t) -> None:
        if self.exchange == "rccl" or in_t.device.type != "cuda":
            self.dist.all_gather_into_tensor(out_t, in_t, group=self.group if self.host_group is None else self.host_group)
            return
        import torch
        n = in_t.numel()
        if self._host is None or self._host[0].numel() < n:
            self._host = (torch.empty(n, dtype=torch.uint8).pin_memory(), torch.empty(n * self.world, dtype=torch.uint8).pin_memory())
        hs, hr = self._host[0][:n], self._host[1][:n * self.world]
        hs.copy_(in_t, non_blocking=True)
        torch.cuda.current_stream(in_t.device).synchronize()
        self.dist.all_gather_into_tensor(hr, hs, group=self.host_group if self.host_group is not None else self.group)
        out_t.copy_(hr, non_blocking=True)

    def _use_all_to_all(self, in_t, backend: str) -> bool:
        """one all_to_all_single (RCCL) instead of batched isend / irecv: device tensors on an "nccl" group"""
        return self.exchange == "rccl" and in_t.device.type == "cuda" and backend == "nccl"

    def _to_writers_into(self, out_t, in_t, send_off, send_len, own: int) -> None:
        """The writer-only exchange of one chunk group: piece [send_off[w], send_off[w] + send_len[w]) of ``in_t`` — the columns
        of the anchors rank w writes — goes to rank w; ``out_t`` receives ``world`` blocks of ``own`` bytes, block j = what rank
        j (genome block part0 + j) computed for THIS rank's anchors.  RCCL: one all_to_all_single with split sizes; a backend
        without all-to-all (gloo) or the host route: batched isend / irecv, the rank's own piece copied in place."""
        dist, world, rank = self.dist, self.world, self.rank
        group = self.group if self.host_group is None else self.host_group
        on_gpu = in_t.device.type == "cuda"
        backend = dist.get_backend(self.group)
        if self._use_all_to_all(in_t, backend):
            # (the pieces are laid out writer by writer: input split w is exactly what rank w gets)
            assert all(send_off[w] == sum(send_len[:w]) for w in range(world))
            dist.all_to_all_single(out_t[:own * world], in_t[:sum(send_len)], output_split_sizes=[own] * world,
                                   input_split_sizes=list(send_len), group=self.group)
            return
        import torch
        src, dst = in_t, out_t
        if on_gpu:  # through pinned host memory (the "host" route, or CUDA tensors over a gloo group)
            n_in, n_out = int(in_t.numel()), own * world
            if self._host is None or self._host[0].numel() < n_in or self._host[1].numel() < n_out:
                self._host = (torch.empty(max(n_in, 8), dtype=torch.uint8).pin_memory(), torch.empty(max(n_out, 8), dtype=torch.uint8).pin_memory())
            src, dst = self._host[0][:n_in], self._host[1][:max(n_out, 1)]
            src.copy_(in_t, non_blocking=True)
            torch.cuda.current_stream(in_t.device).synchronize()
        ops = []
        for j in range(world):
            if j == rank:
                continue
            if send_len[j]:
                ops.append(dist.P2POp(dist.isend, src[send_off[j]:send_off[j] + send_len[j]], j, group))
            if own:
                ops.append(dist.P2POp(dist.irecv, dst[j * own:(j + 1) * own], j, group))
        if own:
            dst[rank * own:(rank + 1) * own].copy_(src[send_off[rank]:send_off[rank] + own])
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        if on_gpu and own:
            out_t[:own * world].copy_(dst[:own * world], non_blocking=True)

    def release(self, a: str) -> None:
        """the writer is done with anchor ``a`` (files written): its full-width rows go back to the context"""
        r = self.full.pop(a, None)
        if r is not None:
            r.close()

    def container(self, a: str):
        if a not in self.full:
            self.full[a] = self.engine.AnchorResult.rows_container(self.ctx, self.k, self.N, self.seqs[a], colsums=True,
                                                                   **self.geometry)
        return self.full[a]

    def _narrow(self, table):
        if self._part is None or self._part_table is not table:
            if self._part is not None:
                self._part.close()
            # a block of up to 8 genomes (config 5: ONE genome per GPU): the probe emits the bit columns itself, the
            # narrow result needs no row buffer
            want_direct = getattr(self.engine, "COLUMNS_DIRECT", False) if self.direct_columns is None else bool(self.direct_columns)
            self._direct = (want_direct and table.ngenomes <= 8 and self.per <= max(self.DIRECT_MAX_WIDTH, 2 if self.direct_columns else 0)
                            and table.spill()[1] == 8)  # (8-slot lines: what pg_result_columns_direct asks for)
            self._part = self.engine.AnchorResult(table, self.merged, colsums=False, rows_only=True,
                                                  **({"columns_only": True} if self._direct else {}))
            if self._direct and not self._part.columns_direct(self.per):
                raise RuntimeError("internal: the block table does not qualify for direct columns")
            if len(self.seqs) > 1:
                self._part.coschedule_ranges(self._contig_anchor, self.group_first)
            self._part_table = table
        return self._part

    def run_pass(self, table, part0: int, nparts: int, accumulate: bool, on_anchor_complete=None, phase_s: Optional[dict] = None) -> None:
        """``phase_s`` (measurement only: bench.py's config5_leg): a dict that receives the seconds this pass spends in
        "probe", "extract" and "merge" — the context is synchronised after each stage, so the pass no longer overlaps
        anything; never set by the product."""
        import time
        pipe, per = self.pipe, self.per
        part = self._narrow(table) if (table is not None and self.merged is not None) else None
        ncontigs = len(self._contig_anchor)
        pending = None  # (group, slot, gather-done event) of the group whose gather is in flight

        def now():
            if phase_s is None:
                return 0.0
            self.ctx.synchronize()
            return time.perf_counter()

        def lap(name, t0):
            if phase_s is None:
                return t0
            t1 = now()
            phase_s[name] = phase_s.get(name, 0.0) + (t1 - t0)
            return t1

        to_writers = self.collective and self.to_writers

        def settle(pend):
            i, slot, ev = pend
            pipe.main_waits(ev)  # (also what frees send[slot] for the next extract into it)
            # recv holds `world` blocks, block j = genome block part0 + j: of the whole group's columns (all-gather / one
            # process), or of this writer's anchors only, which start at tile dest_off[i][rank] of the group
            stride = (self.dest_tiles[i][self.rank] if to_writers else self.group_tiles[i]) * self.col_bytes * per
            t_first = self.dest_off[i][self.rank] if to_writers else 0
            t0 = now()
            for a, c0, nc, toff in self.groups[i]:
                if self.writer[a] != self.rank:
                    continue
                self.container(a).merge_columns_range(self.recv[slot].data_ptr() + (toff - t_first) * self.col_bytes * per, part0, nparts, per,
                                                      c0, nc, accumulate=accumulate, part_stride_bytes=stride)
                t0 = lap("merge", t0)
                if on_anchor_complete is not None and self.last_group[a] == i:
                    on_anchor_complete(a, self.full[a])
                    t0 = lap("statistics", t0)

        for i in range(len(self.groups)):
            slot, nbytes = i & 1, self.group_tiles[i] * self.col_bytes * per
            m0 = self.group_first[i]
            cnt = (self.group_first[i + 1] if i + 1 < len(self.groups) else ncontigs) - m0
            t0 = now()
            if part is not None and self._direct:
                part.run_columns_range(m0, cnt, per, self.send[slot].data_ptr())
                lap("probe", t0)
            elif part is not None:
                part.run_range(m0, cnt)
                t0 = lap("probe", t0)
                part.extract_columns_range(0, per, m0, cnt, self.send[slot].data_ptr())
                lap("extract", t0)
            else:
                self.send[slot].zero(nbytes)  # a rank without a block in this pass contributes zeros
            ready = pipe.mark_main()
            if to_writers:
                unit = self.col_bytes * per
                own = self.dest_tiles[i][self.rank] * unit
                soff, slen = [t * unit for t in self.dest_off[i]], [t * unit for t in self.dest_tiles[i]]
                ev = pipe.on_comm(ready, lambda o=self.recv[slot].t, t=self.send[slot].t[:nbytes], so=soff, sl=slen, ow=own:
                                  self._to_writers_into(o, t, so, sl, ow))
                self.bytes_received += own * (self.world - 1)
            elif self.collective:
                out_t, in_t = self.recv[slot].t[:nbytes * self.world], self.send[slot].t[:nbytes]
                ev = pipe.on_comm(ready, lambda o=out_t, t=in_t: self._gather_into(o, t))
                self.bytes_received += nbytes * (self.world - 1)
            else:
                ev = ready
            if pending is not None:
                settle(pending)
            pending = (i, slot, ev)
        if pending is not None:
            settle(pending)

    def drop_table(self):
        """the block table is about to be closed: its narrow result goes first"""
        if self._part is not None:
            self._part.close()
        self._part, self._part_table = None, None

    def close(self):
        self.drop_table()
        for r in self.full.values():
            r.close()
        self.full = {}
        self.pipe.close()
        for b_ in {id(x): x for x in self.send + self.recv}.values():
            b_.close()
        if self.merged is not None:
            self.merged.close()
            self.merged = None


def run_genome_sharded(index, nblocks: int, group=None, exchange_stats: Optional[dict] = None) -> None:
    """Every rank calls this (``Index.run()`` does, when ``plan_sharding`` says so).

    Genome block b = genomes [b*per, (b+1)*per), per = ceil(N / nblocks).  Pass p: rank r builds the table of block
    p*world + r — a table of that block's genomes only — and probes EVERY anchor genome against it through the
    chunk pipeline of ``ShardedAnchoring``.  The anchor genomes are dealt to the ranks as in the replicated mode
    (``Index.writer_of_anchor``); a writer keeps the full rows of its anchors across the passes and finishes each —
    statistics, BGZF files, tables — as soon as its last chunk of the last pass has been merged.  The files are the
    ones a single table of all genomes gives (``tests/test_gpu_genome_shard.py``, ``tests/test_distributed_cpu.py``)."""
    from concurrent.futures import ThreadPoolExecutor
    from . import index as pidx
    engine = pidx.engine
    rank, world = index.rank, max(1, index.world)
    N, k = index.ngenomes, index.k
    per = (N + nblocks - 1) // nblocks
    nblocks = (N + per - 1) // per
    passes = (nblocks + world - 1) // world
    ctx = index.context
    inputs = {i[1].id: i for i in index.load_inputs()}
    anchors = list(index.anchor_genomes)
    writer = index.writer_of_anchor() if world > 1 else {a: 0 for a in anchors}
    seqs = {a: index.seqset_for(a) for a in anchors}
    for a in anchors:
        if writer[a] != rank:
            continue
        index.genomes[a].ensure_log()
        for nm, ln in zip(seqs[a].names, seqs[a].lens):
            if int(ln) < k:
                index.genomes[a].log.warning(f"Contig {nm} is shorter than k={k}: 0 k-mers (the reference underflows here)")
        index.genomes[a].log.info("Anchoring Started")
    sh = ShardedAnchoring(engine, ctx, k, N, per, rank, world, seqs, writer, index.result_geometry, group,
                          direct_columns=True if getattr(index, "_block_direct", False) else None)
    payload = sum(int(seqs[a].lens.sum()) for a in anchors if writer[a] == rank) * ((N + 7) // 8)
    pool = ThreadPoolExecutor(max_workers=index.writer_jobs(payload))
    joins = []
    # ONE table allocation for all of this rank's blocks, sized for the largest and emptied between the passes
    my_blocks = [b for b in range(rank, nblocks, world)]
    blocks = {b: [inputs[g] for g in range(b * per, min(N, (b + 1) * per)) if g in inputs] for b in my_blocks}
    tbl_mem = None
    if my_blocks:
        # (denser than the library's 3 keys per line when the planner chose fewer, wider blocks: Index.plan_sharding)
        kpl = float(getattr(index, "_block_keys_per_line", 0.0) or 0.0)
        tbl_mem = engine.PanTable(ctx, k, per, expected_keys=max(index._expected_keys(blk) for blk in blocks.values()),
                                  coscheduled=max(1, len(anchors)),  # (a chunk group co-schedules every anchor's chunk)
                                  **({"keys_per_line": kpl} if kpl else {}))
    in_flight = 2 * index.writer_jobs(payload)  # anchors whose full-width rows wait for their writer: bounded

    def finished(a, res):
        joins.append((a, _finish_anchor(index, a, res, pool)))
        while len(joins) > in_flight:  # the oldest writer first: its rows go back to the context before more pile up
            a0, j0 = joins.pop(0)
            j0()
            sh.release(a0)

    try:
        for p in range(passes):
            b = p * world + rank
            g_lo, g_hi = b * per, min(N, (b + 1) * per)
            tbl = None
            if b < nblocks:
                tbl = tbl_mem
                if p:
                    tbl.clear()
                for name, g, ss, min_count, _ in blocks[b]:
                    tbl.insert_seqset(g.id - g_lo, ss, min_count=min_count)
                logger_info("pass %d: table of genomes %d..%d: %s", p, g_lo, g_hi - 1, tbl.stats())
            done = finished if p == passes - 1 else None
            sh.run_pass(tbl, p * world, min(world, nblocks - p * world), passes > 1, done)
            if tbl is not None:
                ctx.synchronize()  # (the pass's probes are done before the table is emptied for the next block)
        for a in anchors:  # an anchor FASTA without a record: nothing was exchanged, its (empty) files are still due
            if a not in sh.last_group and writer[a] == rank:
                joins.append((a, _finish_anchor(index, a, sh.container(a), pool)))
        for a, j in joins:
            j()
            sh.release(a)  # (its full-width rows: next to the block table the largest allocation of the mode)
    finally:
        pool.shutdown(wait=True)
        sh.close()
        if tbl_mem is not None:
            tbl_mem.close()
    if exchange_stats is not None:
        unit = sh.col_bytes * per  # bytes of one block's columns per tile
        exchange_stats.update(bytes_received=sh.bytes_received, passes=passes, nblocks=nblocks, per=per, chunks=len(sh.groups),
                              to_writers=bool(sh.to_writers), exchange=sh.exchange,
                              own_column_bytes_per_pass=sum(d[rank] for d in sh.dest_tiles) * unit,
                              all_column_bytes_per_pass=sum(sh.group_tiles) * unit)
    if sh.dist is not None:
        sh.dist.barrier(group=group)


def logger_info(fmt, *args):
    import logging
    logging.getLogger("panagram_amd.index").info(fmt, *args)


def _finish_anchor(index, name: str, res, pool):
    """statistics from the completed rows, then the genome's files on a writer thread; returns the join"""
    g = index.genomes[name]
    res.rows_epilogue()
    job = dict(res=res, merged=None, genomes=[g.tabulate(res, 0, len(res.seqs.names), list(res.seqs.names))])
    fut = pool.submit(g.write_from_result, job, 0)
    return fut.result
