"""GPU: genome-sharded mode on one device — per-"rank" partial tables (disjoint genome bits),
rows-only anchoring, SUM-combine of the partial rows, statistics from the combined rows.
Must equal the fused single-table result and the reference's golden outputs.
("n8_k21", 8) is BASELINE config 5's layout: 8 genomes, k=21, ONE genome per rank — columns of width 1, per = 1.)
The product path (Index.run() -> plan_sharding -> run_genome_sharded: narrow block tables, chunk pipeline, rows
containers, passes) runs on the one GPU with as many genome blocks as the case asks for."""
import numpy as np
import pytest
import torch

from oracle import pyoracle as po
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", ["columns", "sum"])
@pytest.mark.parametrize("name,world", [("n9_k21", 2), ("n65_k21", 4), ("n40_k31", 8), ("n8_k21", 8)])
def test_partial_tables_combine_equals_reference(ctx, name, world, mode):
    from panagram_amd import engine
    from panagram_amd.distributed import genome_owner, genomes_per_rank
    fx = H.load_case(name)
    n, k = int(fx["ngenomes"]), int(fx["k"])
    dbs = H.case_dbs(fx)
    g = int(fx["anchors"][0])
    recs = po.parse_fasta_cpp(fx[f"fasta_{g}"].tobytes())
    seqs = [s for _, s in recs]
    tables = []
    for r in range(world):
        t = engine.PanTable(ctx, k, n)
        for d, (keys, masks) in enumerate(dbs):
            own = np.uint32(0)
            for gg in range(32 * d, min(32 * d + 32, n)):
                if genome_owner(gg, n, world) == r:
                    own |= np.uint32(1 << (gg % 32))
            m = masks & own
            t.insert_keys(d, keys[m != 0], m[m != 0])
        tables.append(t)
    ss = [engine.SeqSet.from_host(ctx, seqs) for _ in range(world)]
    res = [engine.AnchorResult(tables[r], ss[r], colsums=True, rows_only=True) for r in range(world)]
    for r_ in res:
        r_.run()
    ctx.synchronize()
    if mode == "sum":
        total = res[0].rows_tensor()
        for r_ in res[1:]:
            total += r_.rows_tensor()  # what the SUM all-reduce does across GPUs
    else:
        # what exchange_columns_ does across GPUs: every rank extracts the compact bit columns of
        # its own genomes into its slice of the gathered buffer; rank 0 merges all slices
        per = genomes_per_rank(n, world)
        nb_cols = res[0].columns_bytes(per)
        allb = torch.zeros(world * nb_cols, dtype=torch.uint8, device="cuda")
        for r, r_ in enumerate(res):
            r_.extract_columns(r * per, per, allb.data_ptr() + r * nb_cols)
        ctx.synchronize()
        # the layout is the one the oracle restates (include/panagram_hip.h)
        nks = [len(s_) - k + 1 for s_ in seqs]
        part0, off = [], 0
        for ci in range(len(seqs)):
            part0.append(res[0].download(ci)[0])
        assert np.array_equal(allb[:nb_cols].cpu().numpy(), po.extract_columns(part0, n, 0, per, tile=engine.tile_positions()))
        res[0].merge_columns(allb.data_ptr(), world, per)
        ctx.synchronize()
    torch.cuda.synchronize()
    res[0].rows_epilogue()
    b1, b100, bins, binlens = [], [], [], []
    for ci in range(len(seqs)):
        rows, rows100, bn, info = res[0].download(ci)
        b1.append(rows.tobytes()); b100.append(rows100.tobytes()); bins.append(bn); binlens.append(info["binlen"])
    assert b"".join(b1) == fx[f"a{g}_bitmap1"].tobytes()
    assert b"".join(b100) == fx[f"a{g}_bitmap100"].tobytes()
    assert H.bins_text(n, bins, binlens).encode() == fx[f"a{g}_bitsum.bins.tsv"].tobytes()
    ora = po.anchor_fasta(dbs, fx[f"fasta_{g}"].tobytes(), k, n)
    assert np.array_equal(res[0].colsums().astype(np.int64), ora["colsums"])
    for x in res + ss + tables:
        x.close()


def test_rccl_code_path_world1(ctx):
    """The real torch.distributed path (backend "nccl" = RCCL) with world_size 1 on the one GPU
    of the test box: zero-copy view of the library's row buffer -> all_reduce -> pg_rows_epilogue."""
    import os
    import torch.distributed as dist
    from panagram_amd import engine
    from panagram_amd.distributed import anchor_genome_sharded, build_partial_table
    fx = H.load_case("n9_k21")
    n, k = int(fx["ngenomes"]), int(fx["k"])
    genomes = [[s for _, s in po.parse_fasta_cpp(fx[f"fasta_{g}"].tobytes())] for g in range(n)]
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 300))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        tbl = build_partial_table(ctx, k, n, 0, 1, lambda g: genomes[g])
        g = int(fx["anchors"][0])
        out, cs = anchor_genome_sharded(tbl, genomes[g])
        payload = b"".join(r[0].tobytes() for r in out)
        assert payload == fx[f"a{g}_bitmap1"].tobytes()
        assert b"".join(r[1].tobytes() for r in out) == fx[f"a{g}_bitmap100"].tobytes()
        ora = po.anchor_fasta(H.case_dbs(fx), fx[f"fasta_{g}"].tobytes(), k, n)
        assert np.array_equal(cs, ora["colsums"])
        tbl.close()
    finally:
        dist.destroy_process_group()


def _write_case(tmp_path, fx):
    n = int(fx["ngenomes"])
    rows = ["name\tfasta"]
    for g in range(n):
        fa = tmp_path / f"g{g}.fa"
        fa.write_bytes(fx[f"fasta_{g}"].tobytes())
        rows.append(f"g{g}\t{fa}")
    s = tmp_path / "samples.tsv"
    s.write_text("\n".join(rows) + "\n")
    return s


def _check_tree(out, fx, gzi_like_reference=True):
    import gzip
    import pandas as pd
    n, k = int(fx["ngenomes"]), int(fx["k"])
    dbs = H.case_dbs(fx)
    for g in fx["anchors"]:
        adir = out / "anchor" / f"g{g}"
        for step in (1, 100):
            assert gzip.open(adir / f"bitmap.{step}.gz", "rb").read() == fx[f"a{g}_bitmap{step}"].tobytes()
            gzi = np.fromfile(adir / f"bitmap.{step}.gzi", "<u8")
            if gzi_like_reference:
                assert gzi[0] == np.frombuffer(fx[f"a{g}_gzi{step}"].tobytes(), "<u8")[0]
            else:  # fragments of several ranks back to back: the blocks are cut where the fragments were
                assert len(gzi) == 1 + 2 * int(gzi[0])
        assert (adir / "bitsum.bins.tsv").read_bytes() == fx[f"a{g}_bitsum.bins.tsv"].tobytes()
        assert (adir / "chrs.tsv").read_bytes() == fx[f"a{g}_chrs.tsv"].tobytes()
        ora = po.anchor_fasta(dbs, fx[f"fasta_{g}"].tobytes(), k, n)
        tp = pd.read_csv(adir / "total_paircounts.csv", index_col="name")
        assert np.array_equal(tp["count"].to_numpy(), ora["colsums"])


@pytest.mark.parametrize("direct", [False, True])
@pytest.mark.parametrize("name,nblocks,chunk", [("n8_k21", 8, 1500), ("n8_k21", 1, 1 << 27), ("n9_k21", 4, 700),
                                                ("n65_k21", 4, 1 << 27), ("n40_k31", 3, 900), ("n2_k21", 2, 512)])
def test_index_run_genome_sharded_on_one_gpu(name, nblocks, chunk, direct, tmp_path, monkeypatch):
    """Index.run() in the genome-sharded mode: one GPU works the genome blocks off as passes.  ("n8_k21", 8): one
    genome per block — config 5's layout; chunk: positions per exchanged chunk, small ones make every anchor
    several chunks (double-buffered pipeline) whose last ones are short."""
    from panagram_amd import distributed as pdist
    from panagram_amd import index as pidx
    fx = H.load_case(name)
    s = _write_case(tmp_path, fx)
    out = tmp_path / "idx"
    if direct and (name, nblocks) not in (("n8_k21", 8), ("n2_k21", 2)):
        pytest.skip("columns straight from the probe only exist for one-genome blocks")
    from panagram_amd import engine
    monkeypatch.setattr(engine, "COLUMNS_DIRECT", direct)  # (PG_COLUMNS_DIRECT: off by default)
    monkeypatch.setattr(pdist, "CHUNK_POSITIONS", chunk)
    idx = pidx.Index(str(s), prefix=str(out), k=int(fx["k"]), anchor_genomes=[f"g{g}" for g in fx["anchors"]],
                     shard="genome", genome_blocks=nblocks)
    assert idx.plan_sharding() == ("genome", nblocks)
    idx.run()
    _check_tree(out, fx)


def test_planner_switches_to_genome_blocks_when_the_table_does_not_fit(tmp_path, monkeypatch):
    """the decision itself: with the HBM budget shrunk below the pangenome's table, plan_sharding answers
    ("genome", blocks) with blocks whose tables fit, and the run writes the reference's tree"""
    from panagram_amd import engine
    from panagram_amd import index as pidx
    fx = H.load_case("n9_k21")
    s = _write_case(tmp_path, fx)
    out = tmp_path / "idx"
    idx = pidx.Index(str(s), prefix=str(out), k=int(fx["k"]), anchor_genomes=[f"g{g}" for g in fx["anchors"]])
    assert idx.plan_sharding() == ("replicated", 1)
    whole = engine.PanTable.bytes_for(idx.k, idx.ngenomes, idx._expected_keys(idx.load_inputs()))
    free = idx.context.mem_info()[0]
    # (small fixtures sit at pg_table_create's floor of 2^18 keys: make the reserve eat everything but half a table)
    monkeypatch.setattr(pidx.Index, "batch_bytes", 0)
    monkeypatch.setattr(pidx.Index, "HBM_RESERVE", free - whole // 2)
    mode, nblocks = idx.plan_sharding()
    assert mode == "genome" and nblocks >= 1
    idx.shard, idx.genome_blocks = "genome", 3
    idx.run()
    _check_tree(out, fx)


def test_range_probe_extract_merge_equals_whole(ctx):
    """contig-range calls (the pipeline's units) against the whole-result calls: same rows, same columns"""
    from panagram_amd import engine
    fx = H.load_case("n9_k21")
    n, k = int(fx["ngenomes"]), int(fx["k"])
    dbs = H.case_dbs(fx)
    g = int(fx["anchors"][1])
    seqs = [s_ for _, s_ in po.parse_fasta_cpp(fx[f"fasta_{g}"].tobytes())]
    seqs = seqs + [seqs[0][:777], seqs[-1][5:1900]]  # four contigs, ragged tile counts
    t = engine.PanTable(ctx, k, n)
    for d, (keys, masks) in enumerate(dbs):
        t.insert_keys(d, keys, masks)
    ss = engine.SeqSet.from_host(ctx, seqs)
    whole = engine.AnchorResult(t, ss, colsums=True, rows_only=True)
    whole.run()
    piece = engine.AnchorResult(t, ss, colsums=True, rows_only=True)
    for c0, nc in [(2, 2), (0, 1), (1, 1)]:
        piece.run_range(c0, nc)
    ctx.synchronize()
    for ci in range(len(seqs)):
        assert np.array_equal(piece.download(ci, want_bitmap100=False)[0], whole.download(ci, want_bitmap100=False)[0])
    per = 4
    allc = torch.zeros(whole.columns_bytes(per), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    whole.extract_columns(4, per, allc.data_ptr())
    full = engine.AnchorResult.rows_container(ctx, k, n, ss, colsums=True)
    off = 0
    for c0, nc in [(0, 1), (1, 2), (3, 1)]:
        nb_ = piece.columns_bytes_range(per, c0, nc)
        buf = torch.zeros(nb_, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()  # (torch fills it on its own stream)
        piece.extract_columns_range(4, per, c0, nc, buf.data_ptr())
        ctx.synchronize()
        assert torch.equal(buf, allc[off:off + nb_])
        off += nb_
        # genome block 1 (genomes 4..7) OR-ed into a zeroed container
        full.merge_columns_range(buf.data_ptr(), 1, 1, per, c0, nc, accumulate=True)
        ctx.synchronize()
    assert off == allc.numel()
    keep = np.zeros(((n + 7) // 8) * 8, np.uint8)
    keep[4:8] = 1
    mask = np.packbits(keep.reshape(-1, 8), axis=1, bitorder="little").reshape(-1)
    for ci in range(len(seqs)):
        assert np.array_equal(full.download(ci, want_bitmap100=False)[0], whole.download(ci, want_bitmap100=False)[0] & mask)
    for x in (full, piece, whole, ss, t):
        x.close()


@pytest.mark.parametrize("exchange", ["rccl", "host"])
def test_sharded_pipeline_through_rccl_on_one_rank(ctx, exchange, monkeypatch):
    """The pipeline's collective path on the one GPU of the test box: a process group of size 1 over "nccl"
    (= RCCL), all_gather_into_tensor issued on the side stream for every chunk group, merged rows == golden.
    exchange = "host" (PG_SHARD_EXCHANGE): the fallback without a GPU collective — columns to pinned host memory, an
    all-gather over a gloo group made next to the nccl one, back to the GPU, merged as usual."""
    import os
    monkeypatch.setenv("PG_SHARD_EXCHANGE", exchange)
    import torch.distributed as dist
    from panagram_amd import distributed as pdist
    from panagram_amd import engine
    fx = H.load_case("n8_k21")
    n, k = int(fx["ngenomes"]), int(fx["k"])
    anchors = [int(g) for g in fx["anchors"]]
    seqs = {f"g{g}": engine.SeqSet.from_fasta(ctx, fx[f"fasta_{g}"].tobytes()) for g in anchors}
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 300))
    torch.cuda.set_device(0)
    started = not dist.is_initialized()
    if started:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    old = pdist.CHUNK_POSITIONS
    try:
        pdist.CHUNK_POSITIONS = 1000  # several chunk groups: double buffering across the two streams
        tbl = engine.PanTable(ctx, k, n)
        for g in range(n):
            ss = engine.SeqSet.from_fasta(ctx, fx[f"fasta_{g}"].tobytes())
            tbl.insert_seqset(g, ss)
            ss.close()
        sh = pdist.ShardedAnchoring(engine, ctx, k, n, n, 0, 1, seqs, {a: 0 for a in seqs}, None, None, always_gather=True)
        assert len(sh.groups) > 1
        done = []
        for _ in range(2):  # the second pass overwrites the first: same rows
            sh.run_pass(tbl, 0, 1, False, lambda a, res: (res.rows_epilogue(), done.append(a)))
        ctx.synchronize()
        torch.cuda.synchronize()
        assert sh.bytes_received == 0 and sorted(set(done)) == sorted(seqs)
        for g in anchors:
            res = sh.full[f"g{g}"]
            b1 = b"".join(res.download(ci)[0].tobytes() for ci in range(len(res.seqs.names)))
            b100 = b"".join(res.download(ci)[1].tobytes() for ci in range(len(res.seqs.names)))
            assert b1 == fx[f"a{g}_bitmap1"].tobytes() and b100 == fx[f"a{g}_bitmap100"].tobytes()
        sh.close()
        tbl.close()
    finally:
        pdist.CHUNK_POSITIONS = old
        for s_ in seqs.values():
            s_.close()
        if started:
            dist.destroy_process_group()


@pytest.mark.parametrize("block", [(0, 8), (3, 4), (5, 6)])
def test_probe_emits_columns_directly(ctx, block):
    """pg_anchor_run_columns_range (the probe assembling the bit columns of a narrow block table itself) against
    rows + k_cols_extract, over co-scheduled contig ranges of several anchors: the same bytes.  Blocks: all 8
    genomes (width 8), genome 3 alone (config 5's one genome per GPU, width 1), genome 5 in a wider slot (width 3)."""
    from panagram_amd import engine
    g_lo, g_hi = block
    width = {(0, 8): 8, (3, 4): 1, (5, 6): 3}[block]
    fx = H.load_case("n8_k21")
    k = int(fx["k"])
    tbl = engine.PanTable(ctx, k, g_hi - g_lo)
    for g in range(g_lo, g_hi):
        ss = engine.SeqSet.from_fasta(ctx, fx[f"fasta_{g}"].tobytes())
        tbl.insert_seqset(g - g_lo, ss)
        ss.close()
    sets = [engine.SeqSet.from_fasta(ctx, fx[f"fasta_{g}"].tobytes()) for g in (0, 3, 7)]
    # chunk groups as the pipeline lays them out: contig 0 of every anchor, then contig 1 of every anchor
    merged = engine.SeqSet.concat_ranges(ctx, [(s_, 0, 1) for s_ in sets] + [(s_, 1, 1) for s_ in sets])
    rows = engine.AnchorResult(tbl, merged, colsums=False, rows_only=True)
    cols = engine.AnchorResult(tbl, merged, colsums=False, columns_only=True)
    assert cols.columns_direct(width)
    for r_ in (rows, cols):
        r_.coschedule_ranges([0, 1, 2, 0, 1, 2], [0, 3], 2)
    for c0, nc in [(0, 3), (3, 3)]:
        nb_ = rows.columns_bytes_range(width, c0, nc)
        a = torch.zeros(nb_, dtype=torch.uint8, device="cuda")
        b = torch.full((nb_,), 0xAB, dtype=torch.uint8, device="cuda")
        rows.run_range(c0, nc)
        rows.extract_columns_range(0, width, c0, nc, a.data_ptr())
        cols.run_columns_range(c0, nc, width, b.data_ptr())
        ctx.synchronize()
        assert a.any() and torch.equal(a, b)
    with pytest.raises(engine.PanagramHipError):
        cols.run_range(0, 1)
    for x in (rows, cols, merged, *sets, tbl):
        x.close()


# ---------------------------------------------------------------------------
# TWO processes on the one GPU: the product's own multi-rank path — Index.run() under RANK / WORLD_SIZE, the real
# engine and kernels in both processes, the chunk pipeline with its side stream, a real collective between them.
# RCCL refuses two ranks on one device, so the process group is gloo (it stages the CUDA tensors through the host);
# everything else is what runs under torchrun on a multi-GPU node.
# ---------------------------------------------------------------------------
def _two_rank_worker(rank, world, port, samples, out, k, anchors, shard, nblocks, chunk):
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from panagram_amd import distributed as pdist
        from panagram_amd import index as pidx
        pdist.CHUNK_POSITIONS = chunk
        idx = pidx.Index(samples, prefix=out, k=k, anchor_genomes=anchors, shard=shard, genome_blocks=nblocks)
        assert (idx.rank, idx.world) == (rank, world)
        idx.run()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("exchange", ["rccl", "host"])
@pytest.mark.parametrize("name,shard,nblocks,chunk", [("n8_k21", "genome", 2, 1500), ("n8_k21", "genome", 8, 1 << 27),
                                                      ("n9_k21", "genome", 4, 700), ("n9_k21", "replicated", 0, 1 << 27)])
def test_two_processes_on_one_gpu(name, shard, nblocks, chunk, exchange, tmp_path, monkeypatch):
    """("n8_k21", genome, 2): one pass, each rank one block of 4 genomes; (…, 8): four passes of one-genome blocks
    (config 5's layout; columns straight from the probe); ("n9_k21", 4): blocks of 3, 3, 3 on two ranks — the second
    pass has an idle rank; replicated: pieces of homology classes dealt to the two ranks (both anchor every genome's
    share, co-scheduled), fragments assembled by whoever finds a genome complete, no collective."""
    import torch.multiprocessing as mp
    import os
    if exchange == "host" and shard == "replicated":
        pytest.skip("no exchange step in the contig-sharded mode")
    # ("host": the two ranks' columns go GPU -> pinned host -> gloo all-gather -> GPU, PG_SHARD_EXCHANGE's fallback route;
    # "rccl" here hands the CUDA tensors to the gloo group of this one-GPU set-up, see above)
    monkeypatch.setenv("PG_SHARD_EXCHANGE", exchange)
    fx = H.load_case(name)
    s = _write_case(tmp_path, fx)
    out = tmp_path / "idx"
    port = 29600 + (os.getpid() % 2000)
    mp.spawn(_two_rank_worker, args=(2, port, str(s), str(out), int(fx["k"]), [f"g{g}" for g in fx["anchors"]], shard, nblocks, chunk),
             nprocs=2, join=True)
    _check_tree(out, fx, gzi_like_reference=shard != "replicated")


@pytest.mark.parametrize("shard,nblocks,chunk", [("genome", 8, 1500), ("replicated", 0, 1 << 27)])
def test_eight_processes_on_one_gpu(shard, nblocks, chunk, tmp_path, monkeypatch):
    """An 8-rank dress rehearsal on the one device — the first contact with an 8-GPU node should not be the first time
    anything runs at world = 8.  genome / 8 blocks on "n8_k21" is BASELINE.json configs[4]'s layout exactly: EIGHT real-engine
    processes, one genome's table each, every rank probing every anchor position (several chunk groups per anchor), the 8
    bit columns exchanged per group and merged on the anchors' writers, ONE pass.  RCCL refuses several ranks on one device,
    so the exchange takes the product's host route (PG_SHARD_EXCHANGE=host: columns to pinned memory, all-gather over gloo,
    back to the GPU) — what that route is for.  replicated: pieces of homology classes dealt to 8 ranks, fragments assembled
    without a rendezvous.  Both trees equal the reference binary's golden outputs (cpp/anchor.cpp:139-164, 217-223)."""
    import torch.multiprocessing as mp
    import os
    monkeypatch.setenv("PG_SHARD_EXCHANGE", "host")
    monkeypatch.setenv("PG_MIN_PIECE", "500")  # (the fixture's contigs are a few kb: cut them all the same)
    fx = H.load_case("n8_k21")
    s = _write_case(tmp_path, fx)
    out = tmp_path / "idx"
    port = 29600 + (os.getpid() % 2000)
    mp.spawn(_two_rank_worker, args=(8, port, str(s), str(out), int(fx["k"]), [f"g{g}" for g in fx["anchors"]], shard, nblocks, chunk),
             nprocs=8, join=True)
    _check_tree(out, fx, gzi_like_reference=shard != "replicated")


def _real_gpu_worker(rank, world, port, samples, out, k, anchors, shard, nblocks, chunk):
    """rank r on GPU r, torch.distributed over RCCL ("nccl"): the product's multi-GPU set-up as a launcher gives it"""
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    os.environ.pop("PG_DIST_BACKEND", None)
    os.environ.pop("PG_SHARD_EXCHANGE", None)
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    try:
        from panagram_amd import distributed as pdist
        from panagram_amd import index as pidx
        pdist.CHUNK_POSITIONS = chunk
        idx = pidx.Index(samples, prefix=out, k=k, anchor_genomes=anchors, shard=shard, genome_blocks=nblocks, device=rank)
        assert (idx.rank, idx.world) == (rank, world)
        idx.run()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name,shard,nblocks,chunk", [("n9_k21", "replicated", 0, 1 << 27), ("n8_k21", "genome", 0, 1500),
                                                      ("n8_k21", "genome", 8, 1 << 27)])
def test_real_gpus_over_rccl_give_the_one_rank_tree(name, shard, nblocks, chunk, tmp_path, monkeypatch):
    """SURVEY section 4's "1 vs N GPUs produce identical payload bytes", on REAL GPUs: Index.run() on min(8, device_count)
    ranks, one process per GPU, torch.distributed backend "nccl" (= RCCL over xGMI) — the contig-sharded mode ("replicated":
    pieces of homology classes dealt to the ranks, no collective; cpp/anchor.cpp:217-223 is the axis) and the genome-sharded
    mode (one genome block per GPU, or 8 one-genome blocks as passes: the bit columns cross xGMI) — against the reference
    binary's golden outputs AND against the tree one rank writes on its own.  Skipped on a one-GPU box (there the same
    code runs with several processes on the one device over gloo: test_two/eight_processes_on_one_gpu)."""
    import os
    import torch.multiprocessing as mp
    ngpu = torch.cuda.device_count()
    if ngpu < 2:
        pytest.skip("needs at least two GPUs (RCCL refuses several ranks on one device)")
    world = min(8, ngpu)
    monkeypatch.setenv("PG_MIN_PIECE", "500")  # (the fixture's contigs are a few kb: cut them all the same)
    fx = H.load_case(name)
    s = _write_case(tmp_path, fx)
    anchors = [f"g{g}" for g in fx["anchors"]]
    nb = nblocks if nblocks else world
    out = tmp_path / "idx"
    port = 29800 + (os.getpid() % 2000)
    mp.spawn(_real_gpu_worker, args=(world, port, str(s), str(out), int(fx["k"]), anchors, shard, nb if shard == "genome" else 0, chunk),
             nprocs=world, join=True)
    _check_tree(out, fx, gzi_like_reference=shard != "replicated")
    # ... and the decompressed payloads, TSVs and column sums equal what ONE rank writes
    from panagram_amd import index as pidx
    one = tmp_path / "idx1"
    idx = pidx.Index(str(s), prefix=str(one), k=int(fx["k"]), anchor_genomes=anchors)
    idx.run()
    import gzip
    for a in anchors:
        for fn in ("bitmap.1.gz", "bitmap.100.gz"):
            with gzip.open(out / "anchor" / a / fn) as f1, gzip.open(one / "anchor" / a / fn) as f2:
                assert f1.read() == f2.read(), (a, fn)
        for fn in ("bitsum.bins.tsv", "chrs.tsv", "total_paircounts.csv"):
            assert (out / "anchor" / a / fn).read_bytes() == (one / "anchor" / a / fn).read_bytes(), (a, fn)


@pytest.mark.parametrize("shard,nblocks", [(None, 0), ("genome", 2)])
def test_cli_under_torchrun_joins_the_group_itself(shard, nblocks, tmp_path):
    """`python -m torch.distributed.run --nproc-per-node 2 -m panagram_amd index …`: nobody initialises torch.distributed
    for the CLI, so Index.run() joins the launcher's rendezvous itself (index._ensure_process_group) — without that the
    genome-sharded mode has no group to exchange columns over and the ranks cannot check that they agree on the plan.
    (gloo instead of RCCL and --device 0 for both ranks: this box has one GPU.)"""
    import os
    import subprocess
    import sys
    fx = H.load_case("n8_k21")
    s = _write_case(tmp_path, fx)
    out = tmp_path / "idx"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PG_DIST_BACKEND="gloo", PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    for v in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "PG_SHARD", "PG_GENOME_BLOCKS"):
        env.pop(v, None)
    if shard:
        env.update(PG_SHARD=shard, PG_GENOME_BLOCKS=str(nblocks))
    port = 29700 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "-m", "panagram_amd", "index", str(s), "-o", str(out), "-k", str(int(fx["k"])),
           "--device", "0", "--anchor_genomes", *[f"g{g}" for g in fx["anchors"]]]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert p.returncode == 0, "\n".join(ln for ln in p.stderr.splitlines() if "Traceback" in ln or "Error" in ln or ln.startswith("  File"))[-4000:]
    _check_tree(out, fx, gzi_like_reference=shard == "genome")


@pytest.mark.parametrize("n,per", [(40, 20), (64, 8), (64, 32), (70, 24), (100, 70), (130, 33)])
def test_columns_of_wide_blocks_round_trip(ctx, n, per):
    """The exchange format at more genomes than config 5's eight: bit columns of every genome block extracted from the
    rows (blocks that straddle the rows' 32-bit words, blocks wider than 64 genomes) and merged back — written whole
    (accumulate off: the words no block covers must come out zero; a loop bound that wrapped there hung the kernel) and
    OR-ed in block by block — give the rows again, block by block and all together."""
    from panagram_amd import engine
    k = 21
    gen = po.synth_genomes(n, [3000, 700, 1300], 0.05, 4200 + n)
    genomes = [[po.codes_to_ascii(c) for c in g] for g in gen]
    tbl = engine.PanTable(ctx, k, n)
    for g in range(n):
        ss_ = engine.SeqSet.from_host(ctx, genomes[g])
        tbl.insert_seqset(g, ss_)
        ss_.close()
    ss = engine.SeqSet.from_host(ctx, genomes[n // 2])
    whole = engine.AnchorResult(tbl, ss, colsums=True, rows_only=True)
    whole.run()
    ctx.synchronize()
    rows = [whole.download(ci, want_bitmap100=False)[0] for ci in range(3)]
    assert all(r.any() for r in rows)
    nblocks = (n + per - 1) // per
    nbytes = (n + 7) // 8
    bufs = [torch.zeros(whole.columns_bytes(per), dtype=torch.uint8, device="cuda") for _ in range(nblocks)]
    torch.cuda.synchronize()  # (torch fills them on ITS stream; the library works on the context's)
    for b in range(nblocks):
        whole.extract_columns(b * per, per, bufs[b].data_ptr())
    ctx.synchronize()

    def block_mask(b):
        keep = np.zeros(nbytes * 8, np.uint8)
        keep[b * per:min(n, (b + 1) * per)] = 1
        return np.packbits(keep.reshape(-1, 8), axis=1, bitorder="little").reshape(-1)

    one = engine.AnchorResult.rows_container(ctx, k, n, ss, colsums=True)
    for b in range(nblocks):  # one block written whole: everything else zero
        one.merge_columns_range(bufs[b].data_ptr(), b, 1, per, 0, 3, accumulate=False)
        ctx.synchronize()
        for ci in range(3):
            assert np.array_equal(one.download(ci, want_bitmap100=False)[0], rows[ci] & block_mask(b)), (b, ci)
    acc = engine.AnchorResult.rows_container(ctx, k, n, ss, colsums=True)
    for b in reversed(range(nblocks)):  # all blocks OR-ed in, last first
        acc.merge_columns_range(bufs[b].data_ptr(), b, 1, per, 0, 3, accumulate=True)
    ctx.synchronize()
    for ci in range(3):
        assert np.array_equal(acc.download(ci, want_bitmap100=False)[0], rows[ci])
    for x in (one, acc, whole, ss, tbl):
        x.close()


def test_config5_as_specified_at_reduced_length_through_the_pass_mode(ctx):
    """BASELINE.json configs[4]'s parameters — 8 genomes in 24 contigs each, k = 21, d = 0.05, genome_blocks = 8 (one genome
    per block) — at a tenth of the length (8 x 300 Mb: 2.4 x 10^9 positions, 3 x 10^8 keys per block table), on one GPU
    through the product's pass mode (distributed.ShardedAnchoring: the block tables built one after another in ONE
    allocation, every anchor position probed against each, the block's bit column OR-ed into the full rows) — the
    same function ``bench.py`` runs at full size as ``config5_leg``.  The rows' head (anchor 0) and tail (the end of
    anchor 5's last contig) equal the CPU oracle's, whose k-mer DB is built by brute force with torch; every anchor
    holds all of its own k-mers.  Byte layout: cpp/anchor.cpp:139-164."""
    import types
    import bench
    dev = torch.device("cuda", 0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        out = bench.config5_leg(ctx, dev, types.SimpleNamespace(seed=1234), genome_mb=300, contigs=24, sample_n=200_000)
    finally:
        ctx.set_stream(None)
    assert out["rows_equal_gpu"] is True, out["rows_check"]
    assert out["anchors_hold_all_own_kmers"] and out["anchors_completed"] == 8
    assert out["genome_blocks"] == 8 and len(out["passes"]) == 8 and out["chunk_groups_per_pass"] >= 2
    assert out["positions"] == 8 * 24 * (12_500_000 - 20)
    # d = 0.05: a derived genome shares 0.95^21 = 34 % of its k-mers with the base genome — the blocks' tables are as large
    # as each other, and their union (what ONE table would have to hold) is several times one block's
    keys = [p["table_keys"] for p in out["passes"]]
    assert min(keys) > 0.95 * max(keys) and 2.9e8 < max(keys) < 3.0e8


def test_config5_in_four_passes_of_two_genome_blocks(ctx):
    """Round 6: the same job with TWO genomes per block in a table created denser than the library's 3 keys per line
    (pg_table_create_dense: the union of two genomes' k-mers then fits one GPU at BASELINE.json configs[4]'s full size) — four
    passes instead of eight, two bit columns per pass through the one-byte rows of the narrow result; at a tenth of the
    length here.  Heads and tails against the CPU oracle again, every anchor holds all of its own k-mers."""
    import types
    import bench
    dev = torch.device("cuda", 0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        out = bench.config5_leg(ctx, dev, types.SimpleNamespace(seed=1234), genome_mb=300, contigs=24, sample_n=200_000, per=2, keys_per_line=4.5)
    finally:
        ctx.set_stream(None)
    assert out["rows_equal_gpu"] is True, out["rows_check"]
    assert out["anchors_hold_all_own_kmers"] and out["anchors_completed"] == 8
    assert out["genome_blocks"] == 4 and out["genomes_per_block"] == 2 and len(out["passes"]) == 4
    # two genomes at d = 0.05 from one ancestor share a ninth of their k-mers (0.95^42): the union is nearly twice one genome's
    keys = [p["table_keys"] for p in out["passes"]]
    assert 4.6e8 < min(keys) and max(keys) < 6.0e8
    # created at 4.5 keys per line and not grown back while it filled: the table is 128 B x keys / 4.5, within rounding
    for p in out["passes"]:
        assert 0.9 < p["table_bytes"] / (128.0 * max(keys) * 1.04 / 4.5) < 1.1, (p["table_bytes"], max(keys))


def test_config5_pass_mode_with_many_chunk_groups(ctx):
    """the same at 8 x 24 Mb with chunks of 2^20 positions: every anchor's contigs spread over 24 chunk groups (the
    double-buffered extract / merge pipeline turns over many times), heads and tails again against the oracle"""
    import types
    import bench
    dev = torch.device("cuda", 0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        out = bench.config5_leg(ctx, dev, types.SimpleNamespace(seed=77), genome_mb=24, contigs=24, sample_n=100_000,
                                chunk_positions=1 << 20)
    finally:
        ctx.set_stream(None)
    assert out["rows_equal_gpu"] is True and out["anchors_hold_all_own_kmers"] and out["chunk_groups_per_pass"] == 24
