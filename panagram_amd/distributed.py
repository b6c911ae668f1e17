"""Multi-GPU modes of the anchor path (one process per GPU, ``torch.distributed``; backend
"nccl" is RCCL over xGMI on MI355X, "gloo" in the CPU tests).

1. **Contig-sharded** (BASELINE configs 2-4): the table is replicated, anchor
   contigs are independent units.  ``plan_shards`` assigns (genome, contig) units to ranks
   longest-first; every rank anchors its units and leaves per-contig part files; the rank
   that owns a genome assembles that genome's files in FASTA order.  No data-path
   collective — one barrier between the two phases.  Outputs do not depend on the GPU count.

2. **Genome-sharded** (config 5: the union of k-mer tables exceeds one GPU's 288 GB): rank r
   holds the table of ITS genomes only (full-width rows, the other genomes' bits zero);
   every rank anchors every position; the partial rows are combined over xGMI and the
   row statistics are taken from the combined rows (``pg_rows_epilogue``).  Because ranks own
   disjoint bits, a uint8 SUM all-reduce equals the bitwise OR (no carries) — RCCL has no
   bitwise reductions — and moves 2(n-1)/n bytes per row byte, less than all-gathering n
   partial copies.  ``all_gather`` + local OR is kept as the reference formulation.
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


def run_index_sharded(index, rank: int, world: int, barrier: Callable[[], None],
                      anchor_fn: Optional[Callable] = None) -> None:
    """Every rank calls this.  ``anchor_fn(genome, seqs) -> ([(rows, rows100, bins, info)], colsums)``
    defaults to the GPU path (``Genome.anchor_contigs`` against this rank's replica of the table)."""
    from .index import read_fasta
    k = index.k
    recs: Dict[str, List[Tuple[str, bytes]]] = {}
    units: List[Unit] = []
    for name in index.anchor_genomes:
        recs[name] = list(read_fasta(index.genomes[name].fasta))
        units += [(name, ci, max(0, len(s) - k + 1)) for ci, (_, s) in enumerate(recs[name])]
    mine = plan_shards(units, world)[rank]
    if anchor_fn is None:
        table = index.build_table()  # replicated: every rank builds (or loads) the whole table
        anchor_fn = lambda genome, seqs: genome.anchor_contigs(table, seqs)  # noqa: E731
    by_genome: Dict[str, List[int]] = {}
    for name, ci, _ in mine:
        by_genome.setdefault(name, []).append(ci)
    for name, cis in by_genome.items():
        g = index.genomes[name]
        os.makedirs(_parts_dir(g), exist_ok=True)
        results, cs = anchor_fn(g, [recs[name][ci][1] for ci in cis])
        for ci, (rows, rows100, bins, info) in zip(cis, results):
            np.savez(os.path.join(_parts_dir(g), f"{ci}.tmp.npz"), rows=rows, rows100=rows100, bins=bins,
                     nkmers=info["nkmers"], nbins=info["nbins"], binlen=info["binlen"], nrows100=info["nrows100"])
            os.replace(os.path.join(_parts_dir(g), f"{ci}.tmp.npz"), os.path.join(_parts_dir(g), f"{ci}.npz"))
        np.save(os.path.join(_parts_dir(g), f"colsums.{rank}.npy"), np.asarray(cs, dtype=np.int64))
    barrier()
    # phase 2: the owner of a genome assembles its files in FASTA order
    for gi, name in enumerate(index.anchor_genomes):
        if gi % world != rank:
            continue
        g = index.genomes[name]
        results = []
        for ci in range(len(recs[name])):
            z = np.load(os.path.join(_parts_dir(g), f"{ci}.npz"))
            info = dict(nkmers=int(z["nkmers"]), nbins=int(z["nbins"]), binlen=int(z["binlen"]),
                        nrows100=int(z["nrows100"]))
            results.append((z["rows"], z["rows100"], z["bins"], info))
        cs = np.zeros(index.ngenomes, np.int64)
        for r in range(world):
            p = os.path.join(_parts_dir(g), f"colsums.{r}.npy")
            if os.path.exists(p):
                cs += np.load(p)
        g.write_outputs([nm for nm, _ in recs[name]], results, cs)
        shutil.rmtree(_parts_dir(g))
    barrier()


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
