"""CPU, world_size 2 over gloo: the multi-GPU host logic (sharding plan, BGZF fragment assembly, row combine,
the genome-sharded pipeline).  The engine layer is swapped for an oracle-backed stand-in (tests/fake_engine.py) so
the N>1 control paths run here without a device."""
import gzip
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import pyoracle as po
from tests import helpers as H


def test_plan_shards_is_balanced_and_deterministic():
    from panagram_amd.distributed import plan_shards
    units = [(f"g{g}", c, n) for g in range(5) for c, n in enumerate([3000, 2000, 2300, 1900, 2700, 10])]
    for world in (1, 2, 3, 8):
        sh = plan_shards(units, world)
        assert sorted(u for s in sh for u in s) == sorted(units)
        loads = [sum(u[2] for u in s) for s in sh]
        assert max(loads) - min(loads) <= max(u[2] for u in units)
        assert sh == plan_shards(list(reversed(units)), world)


def _worker(rank, world, port, root, name):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from panagram_amd import index as pidx
        from panagram_amd.distributed import combine_rows_, combine_rows_allgather, genome_owner, run_index_sharded
        fx = H.load_case(name)
        n, k = int(fx["ngenomes"]), int(fx["k"])
        dbs = H.case_dbs(fx)
        from tests import fake_engine
        pidx.engine = fake_engine  # the oracle stands in for the GPU below the engine layer; everything above is the product's
        idx = pidx.Index(os.path.join(root, "idx"), mode="w")
        run_index_sharded(idx, rank, world, dist.barrier)
        # genome-sharded combine: each rank holds rows with only its genomes' bits
        g = int(fx["anchors"][0])
        full = np.frombuffer(fx[f"a{g}_bitmap1"].tobytes(), np.uint8).copy()
        nb = (n + 7) // 8
        keep = np.zeros(nb * 8, np.uint8)
        for gg in range(n):
            keep[gg] = genome_owner(gg, n, world) == rank
        mask = np.packbits(keep.reshape(nb, 8), axis=1, bitorder="little").reshape(-1)
        part = (full.reshape(-1, nb) & mask).reshape(-1)
        t = torch.from_numpy(part.copy())
        ag = combine_rows_allgather(t)
        combine_rows_(t)
        assert np.array_equal(t.numpy(), full) and np.array_equal(ag.numpy(), full)
        # the default exchange: compact bit columns of this rank's genomes, all-gathered, merged
        from panagram_amd.distributed import gather_columns, genomes_per_rank
        recs = po.parse_fasta_cpp(fx[f"fasta_{g}"].tobytes())
        nks = [len(sq) - k + 1 for _, sq in recs]
        per = genomes_per_rank(n, world)
        rows_part, off = [], 0
        for nk in nks:
            rows_part.append(part.reshape(-1, nb)[off:off + nk])
            off += nk
        mine = po.extract_columns(rows_part, n, rank * per, per, tile=1024)
        allb = gather_columns(torch.from_numpy(mine)).numpy()
        blocks = np.split(allb, world)
        assert np.array_equal(blocks[rank], mine)
        merged = po.merge_columns(blocks, nks, n, per, tile=1024)
        assert np.array_equal(np.concatenate(merged).reshape(-1), full)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name", ["n9_k21", "n65_k21"])
def test_world2_sharded_index_equals_reference(name, tmp_path):
    from panagram_amd import index as pidx
    fx = H.load_case(name)
    n = int(fx["ngenomes"])
    rows = ["name\tfasta"]
    for g in range(n):
        fa = tmp_path / f"g{g}.fa"
        fa.write_bytes(fx[f"fasta_{g}"].tobytes())
        rows.append(f"g{g}\t{fa}")
    s = tmp_path / "samples.tsv"
    s.write_text("\n".join(rows) + "\n")
    pidx.Index(str(s), prefix=str(tmp_path / "idx"), k=int(fx["k"]), prepare=True,
               anchor_genomes=[f"g{g}" for g in fx["anchors"]])
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path), name), nprocs=2, join=True)
    for g in fx["anchors"]:
        adir = tmp_path / "idx" / "anchor" / f"g{g}"
        assert gzip.open(adir / "bitmap.1.gz", "rb").read() == fx[f"a{g}_bitmap1"].tobytes()
        assert gzip.open(adir / "bitmap.100.gz", "rb").read() == fx[f"a{g}_bitmap100"].tobytes()
        assert (adir / "bitsum.bins.tsv").read_bytes() == fx[f"a{g}_bitsum.bins.tsv"].tobytes()
        assert (adir / "chrs.tsv").read_bytes() == fx[f"a{g}_chrs.tsv"].tobytes()
        assert not (adir / ".parts").exists()


def test_index_deals_anchor_genomes_to_ranks(tmp_path, monkeypatch):
    """Index.run under torchrun (RANK / WORLD_SIZE): every anchor genome goes to exactly one rank,
    larger FASTAs first, the same answer in every process."""
    from panagram_amd import index as pidx
    rows = ["name\tfasta"]
    sizes = [900, 100, 500, 400, 300, 50, 800]
    for g, sz in enumerate(sizes):
        fa = tmp_path / f"g{g}.fa"
        fa.write_bytes(b">c\n" + b"A" * sz + b"\n")
        rows.append(f"g{g}\t{fa}")
    s = tmp_path / "samples.tsv"
    s.write_text("\n".join(rows) + "\n")
    for world in (1, 2, 3, 8):
        monkeypatch.setenv("WORLD_SIZE", str(world))
        dealt = []
        for rank in range(world):
            monkeypatch.setenv("RANK", str(rank))
            idx = pidx.Index(str(s), prefix=str(tmp_path / f"idx{world}"), k=21)
            assert (idx.rank, idx.world) == (rank, world)
            mine = idx.my_anchor_genomes()
            assert mine == [n for n in idx.anchor_genomes if n in mine]  # sample order is kept
            dealt.append(mine)
        flat = [n for m in dealt for n in m]
        assert sorted(flat) == sorted(f"g{g}" for g in range(len(sizes)))
        loads = [sum(sizes[int(n[1:])] for n in m) for m in dealt]
        if world == 2:
            assert abs(loads[0] - loads[1]) <= 150
        if world == 8:
            assert sum(1 for m in dealt if not m) == 1  # 7 genomes on 8 ranks


# ---------------------------------------------------------------------------
# Index.run() itself on 2 ranks over gloo: the engine is swapped for the oracle-backed stand-in
# (tests/fake_engine.py), everything above it is the product's own code — mode choice, genome blocks,
# chunk pipeline, all-gather of bit columns, merge on the writer rank, files.
# ---------------------------------------------------------------------------
def _prepare_index(tmp_path, fx, **kw):
    from panagram_amd import index as pidx
    n = int(fx["ngenomes"])
    rows = ["name\tfasta"]
    for g in range(n):
        fa = tmp_path / f"g{g}.fa"
        fa.write_bytes(fx[f"fasta_{g}"].tobytes())
        rows.append(f"g{g}\t{fa}")
    s = tmp_path / "samples.tsv"
    s.write_text("\n".join(rows) + "\n")
    pidx.Index(str(s), prefix=str(tmp_path / "idx"), k=int(fx["k"]), prepare=True,
               anchor_genomes=[f"g{g}" for g in fx["anchors"]], **kw)
    return str(tmp_path / "idx")


def _check_tree(idx_dir, fx):
    import pandas as pd
    n, k = int(fx["ngenomes"]), int(fx["k"])
    for g in fx["anchors"]:
        adir = os.path.join(idx_dir, "anchor", f"g{g}")
        assert gzip.open(os.path.join(adir, "bitmap.1.gz"), "rb").read() == fx[f"a{g}_bitmap1"].tobytes()
        assert gzip.open(os.path.join(adir, "bitmap.100.gz"), "rb").read() == fx[f"a{g}_bitmap100"].tobytes()
        assert open(os.path.join(adir, "bitsum.bins.tsv"), "rb").read() == fx[f"a{g}_bitsum.bins.tsv"].tobytes()
        assert open(os.path.join(adir, "chrs.tsv"), "rb").read() == fx[f"a{g}_chrs.tsv"].tobytes()
        ora = po.anchor_fasta(H.case_dbs(fx), fx[f"fasta_{g}"].tobytes(), k, n)
        tp = pd.read_csv(os.path.join(adir, "total_paircounts.csv"), index_col="name")
        assert np.array_equal(tp["count"].to_numpy(), ora["colsums"])
        assert os.path.exists(os.path.join(idx_dir, "logs", f"anchor.g{g}.log.txt"))
        # the reference's own timing artefact for the step (Snakemake's benchmark: file, workflow/Snakefile:43-44)
        bench = open(os.path.join(idx_dir, "logs", f"anchor.g{g}.benchmark.txt")).read().split("\n")
        assert bench[0].split("\t") == ["s", "h:m:s", "max_rss", "max_vms", "max_uss", "max_pss", "io_in", "io_out", "mean_load", "cpu_time"]
        row = bench[1].split("\t")
        assert len(row) == 10 and float(row[0]) >= 0 and row[1].count(":") == 2 and float(row[2]) > 0 and bench[2] == ""


def _index_run_worker(rank, world, port, idx_dir, shard, nblocks, chunk):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from panagram_amd import distributed as pdist
        from panagram_amd import index as pidx
        from tests import fake_engine
        pidx.engine = fake_engine
        pdist.CHUNK_POSITIONS = chunk  # several chunks per anchor: the pipeline's double buffering is exercised
        idx = pidx.Index(idx_dir, mode="w", shard=shard, genome_blocks=nblocks)
        assert (idx.rank, idx.world) == (rank, world)
        idx.run()
        if getattr(idx, "exchange_stats", None):  # what the rank's exchange moved, for the tests that count bytes
            import json
            with open(os.path.join(idx_dir, f"exchange.rank{rank}.json"), "w") as f:
                json.dump(idx.exchange_stats, f)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name,nblocks", [("n8_k21", 2), ("n8_k21", 8), ("n9_k21", 4), ("n65_k21", 2)])
def test_world2_index_run_genome_sharded(name, nblocks, tmp_path):
    """genome blocks = ranks (one pass), more blocks than ranks (passes, accumulate), one genome per block
    (config 5's layout), a last block that is short (9 genomes in blocks of 3 -> 3 blocks on 2 ranks: the second
    pass has an idle rank sending zeros), rows wider than a word"""
    fx = H.load_case(name)
    idx_dir = _prepare_index(tmp_path, fx)
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_index_run_worker, args=(2, port, idx_dir, "genome", nblocks, 1500), nprocs=2, join=True)
    _check_tree(idx_dir, fx)


@pytest.mark.parametrize("shard,nblocks,chunk", [("genome", 8, 1500), ("replicated", 0, 1 << 27)])
def test_world8_index_run_dress_rehearsal(shard, nblocks, chunk, tmp_path):
    """EIGHT ranks (the node size BASELINE.json's multi-GPU configs name) through Index.run() itself over gloo, the
    stand-in engine below it.  genome / 8 blocks on "n8_k21" is configs[4]'s layout exactly — ONE genome per rank, one
    pass, every rank probing every anchor position against its one-genome table, 8 bit columns all-gathered per chunk
    group (several groups per anchor) and merged on the anchors' writers; replicated = pieces of homology classes dealt to
    8 ranks, fragments assembled without a rendezvous.  The tree equals the reference binary's golden outputs.  The
    reference's only parallel axis is one OpenMP thread per anchor FASTA (cpp/anchor.cpp:217-223)."""
    fx = H.load_case("n8_k21")
    idx_dir = _prepare_index(tmp_path, fx)
    port = 29500 + (os.getpid() % 2000)
    env_before = os.environ.get("PG_MIN_PIECE")
    os.environ["PG_MIN_PIECE"] = "500"  # (the fixture's contigs are a few kb: cut them all the same)
    try:
        mp.spawn(_index_run_worker, args=(8, port, idx_dir, shard, nblocks, chunk), nprocs=8, join=True)
    finally:
        if env_before is None:
            os.environ.pop("PG_MIN_PIECE", None)
        else:
            os.environ["PG_MIN_PIECE"] = env_before
    _check_tree(idx_dir, fx)


@pytest.mark.parametrize("world,nblocks,chunk", [(2, 2, 1500), (8, 8, 1500), (3, 5, 900)])
def test_exchange_sends_columns_to_their_writer_only(world, nblocks, chunk, tmp_path, monkeypatch):
    """The genome-sharded exchange (round 6): a rank's columns of an anchor go to that anchor's WRITER only (batched isend /
    irecv here: gloo has no all-to-all; all_to_all_single on RCCL) — every rank receives (world - 1) x its OWN anchors'
    columns per pass — where PG_SHARD_EXCHANGE=allgather delivers every anchor's columns to every rank: world x the bytes
    over the run.  Both trees equal the reference binary's golden outputs; (3 ranks, 5 blocks): two passes, the second with an
    idle rank, 8 anchors dealt 3 / 3 / 2."""
    import json
    fx = H.load_case("n8_k21")
    monkeypatch.setenv("PG_MIN_PIECE", "500")
    got = {}
    for mode in ("rccl", "allgather"):
        monkeypatch.setenv("PG_SHARD_EXCHANGE", mode)
        sub = tmp_path / mode
        sub.mkdir()
        idx_dir = _prepare_index(sub, fx)
        port = 29500 + (os.getpid() % 2000)
        mp.spawn(_index_run_worker, args=(world, port, idx_dir, "genome", nblocks, chunk), nprocs=world, join=True)
        _check_tree(idx_dir, fx)
        got[mode] = [json.load(open(os.path.join(idx_dir, f"exchange.rank{r}.json"))) for r in range(world)]
    w, g = got["rccl"], got["allgather"]
    assert all(x["to_writers"] for x in w) and not any(x["to_writers"] for x in g)
    passes = w[0]["passes"]
    assert passes == (nblocks + world - 1) // world
    total = w[0]["all_column_bytes_per_pass"]
    assert total > 0 and all(x["all_column_bytes_per_pass"] == total for x in w + g)
    assert sum(x["own_column_bytes_per_pass"] for x in w) == total  # every anchor has exactly one writer
    for x in w:   # (world - 1) x the rank's own anchors' columns, every pass
        assert x["bytes_received"] == (world - 1) * x["own_column_bytes_per_pass"] * passes
    for x in g:   # the all-gather: (world - 1) x everybody's
        assert x["bytes_received"] == (world - 1) * total * passes
    assert sum(x["bytes_received"] for x in g) == world * sum(x["bytes_received"] for x in w)


def _a2a_worker(rank, world, port, idx_dir, nblocks, chunk):
    """genome-sharded Index.run() whose exchange takes the all_to_all_single BRANCH (what RCCL gets) on a gloo group: the
    collective itself is stood in for by isend / irecv with the same split semantics, so the split arithmetic — input pieces
    consecutive by destination, equal output pieces, empty pieces for writers without an anchor in a chunk group — is what is
    exercised; every call's splits are checked against the collective's contract"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from panagram_amd import distributed as pdist
        from panagram_amd import index as pidx
        from tests import fake_engine
        pidx.engine = fake_engine
        pdist.CHUNK_POSITIONS = chunk
        calls = []

        def all_to_all_single(out, inp, output_split_sizes=None, input_split_sizes=None, group=None):
            assert sum(input_split_sizes) == inp.numel() and sum(output_split_sizes) == out.numel()
            assert len(input_split_sizes) == len(output_split_sizes) == world and len(set(output_split_sizes)) == 1
            calls.append((list(input_split_sizes), output_split_sizes[0]))
            io, oo, ops = 0, 0, []
            for j in range(world):
                a, b = inp[io:io + input_split_sizes[j]], out[oo:oo + output_split_sizes[j]]
                if j == rank:
                    b.copy_(a)
                else:
                    if a.numel():
                        ops.append(dist.P2POp(dist.isend, a, j, group))
                    if b.numel():
                        ops.append(dist.P2POp(dist.irecv, b, j, group))
                io, oo = io + input_split_sizes[j], oo + output_split_sizes[j]
            for req in (dist.batch_isend_irecv(ops) if ops else []):
                req.wait()
        dist.all_to_all_single = all_to_all_single
        pdist.ShardedAnchoring._use_all_to_all = lambda self, in_t, backend: True
        idx = pidx.Index(idx_dir, mode="w", shard="genome", genome_blocks=nblocks)
        idx.run()
        assert calls and any(0 in c[0] for c in calls) == (os.environ.get("PG_TEST_EXPECT_EMPTY") == "1"), calls[:4]
    finally:
        dist.destroy_process_group()


def test_all_to_all_splits_with_writers_that_have_no_anchor_in_a_group(tmp_path, monkeypatch):
    """The RCCL branch of the writer-only exchange (all_to_all_single with split sizes), never run on hardware by the builder,
    exercised through a stand-in with the collective's contract: anchors of very different lengths on 3 ranks and small chunks,
    so that the late chunk groups hold anchors of ONE writer only — the others' input pieces are empty and must still sit at
    their place in the order of destinations.  The tree equals the one-rank run's."""
    from panagram_amd import index as pidx
    from tests import fake_engine
    s = _small_pangenome(tmp_path, [[30000, 9000], [5000], [12000, 700], [2500]])
    geo = dict(k=21, lowres_step=50, max_bin_kbp=3, min_bin_count=5)
    monkeypatch.setattr(pidx, "engine", fake_engine)
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setenv("RANK", "0")
    pidx.Index(str(s), prefix=str(tmp_path / "one"), **geo).run()
    pidx.Index(str(s), prefix=str(tmp_path / "many"), prepare=True, **geo)
    monkeypatch.setenv("PG_TEST_EXPECT_EMPTY", "1")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_a2a_worker, args=(3, port, str(tmp_path / "many"), 4, 2000), nprocs=3, join=True)
    _trees_equal(tmp_path / "one", tmp_path / "many", [f"g{g}" for g in range(4)])


def test_world2_index_run_replicated(tmp_path):
    fx = H.load_case("n9_k21")
    idx_dir = _prepare_index(tmp_path, fx)
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_index_run_worker, args=(2, port, idx_dir, "replicated", 0, 1 << 27), nprocs=2, join=True)
    _check_tree(idx_dir, fx)


def test_single_process_passes_and_mode_choice(tmp_path, monkeypatch):
    """One process: the planner picks the genome-sharded mode when the table would not fit, with as many genome
    blocks (passes) as it takes; the files are the same."""
    from panagram_amd import index as pidx
    from tests import fake_engine
    fx = H.load_case("n8_k21")
    idx_dir = _prepare_index(tmp_path, fx)
    monkeypatch.setattr(pidx, "engine", fake_engine)
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setenv("RANK", "0")
    idx = pidx.Index(idx_dir, mode="w")
    assert idx.plan_sharding() == ("replicated", 1)
    idx._inputs = None
    # shrink the "HBM": the whole table (about 3e4 keys x 43 B) no longer fits next to the reserve, a block of 2 does
    monkeypatch.setattr(pidx.Index, "HBM_RESERVE", 0)
    monkeypatch.setattr(pidx.Index, "batch_bytes", 0)
    keys_all = pidx.Index._expected_keys(idx.load_inputs())
    monkeypatch.setattr(fake_engine, "HBM_FREE", int(keys_all * fake_engine.BYTES_PER_KEY * 0.45) + 2 * 4200)
    mode, nblocks = idx.plan_sharding()
    assert mode == "genome" and 2 <= nblocks <= 8
    idx.run()
    _check_tree(idx_dir, fx)


def test_planner_prefers_fewer_denser_genome_blocks(tmp_path, monkeypatch):
    """Round 6: when the tables do not fit at the library's 3 keys per line, the planner weighs FEWER, denser blocks
    (pg_table_create_dense) against more passes with the measured rate model (Index._block_rate): HBM that holds one genome's
    table at 3 keys per line but two genomes' union only at ~4 gives blocks of two at that density — half the passes — and
    the files are the same."""
    from panagram_amd import index as pidx
    from tests import fake_engine
    fx = H.load_case("n8_k21")
    idx_dir = _prepare_index(tmp_path, fx)
    monkeypatch.setattr(pidx, "engine", fake_engine)
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setattr(pidx.Index, "HBM_RESERVE", 0)
    monkeypatch.setattr(pidx.Index, "batch_bytes", 0)
    idx = pidx.Index(idx_dir, mode="w")
    inputs = idx.load_inputs()
    by_id = {i[1].id: i for i in inputs}
    keys1 = max(pidx.Index._expected_keys([by_id[g]]) for g in range(8))
    keys2 = max(pidx.Index._expected_keys([by_id[g], by_id[g + 1]]) for g in range(0, 8, 2))
    positions = sum(int(i[2].lens.sum()) for i in inputs if i[0] in idx.anchor_genomes)
    longest = max(int(i[2].lens.sum()) for i in inputs)
    rows = max(2 * longest, len(idx.anchor_genomes) * longest)  # (what plan_sharding keeps for full-width rows, one byte per position here)
    # room for a two-genome table at 4 keys per line (+ its narrow rows), not at 3; one genome at 3 fits easily
    room = int(keys2 * 128 * 1.02 / 4.0)
    assert fake_engine.PanTable.bytes_for(21, 1, keys1) < room < fake_engine.PanTable.bytes_for(21, 2, keys2) * 128 // (fake_engine.BYTES_PER_KEY * 1)
    monkeypatch.setattr(fake_engine, "HBM_FREE", room + positions + rows)
    monkeypatch.setattr(fake_engine, "BYTES_PER_KEY", 128 / 3.0)  # (bytes_for at the library's density = 128-byte lines of 3 keys, as the real engine)
    del fake_engine.CREATED_DENSITIES[:]
    mode, nblocks = idx.plan_sharding()
    # (the fixture's genomes are near copies of each other: four of them have few more distinct k-mers than two, so blocks of
    # four at a higher density may beat blocks of two by the model — either way fewer than eight blocks, denser than 3)
    kpl = idx._block_keys_per_line
    assert mode == "genome" and nblocks in (2, 4), (mode, nblocks)
    assert 3.0 < kpl <= pidx.Index.BLOCK_KPL_MAX + 0.1, kpl
    per = 8 // nblocks
    assert nblocks / pidx.Index._block_rate(per, kpl) < 8 / pidx.Index._block_rate(1, 3.0)  # cheaper than one genome per block by the model
    # the rate model: denser and wider is slower per pass, but not by the factor of two the halved passes win
    r = pidx.Index._block_rate
    assert r(1, 3.0) == 1.0 and r(2, 3.6) > 0.75 and r(2, 5.5) < r(2, 4.0) < r(2, 3.0) < r(1, 3.0) and 4 / r(2, 4.0) < 8 / r(1, 3.0)
    idx.run()
    assert fake_engine.CREATED_DENSITIES == [kpl], fake_engine.CREATED_DENSITIES  # ONE block table (re-used pass after pass), at the planned density
    _check_tree(idx_dir, fx)


# ---------------------------------------------------------------------------
# the contig-sharded multi-GPU partition: pieces of homology classes
# ---------------------------------------------------------------------------
def test_plan_class_pieces_covers_balances_and_aligns():
    from panagram_amd.distributed import piece_alignment, plan_class_pieces
    # 8 genomes x 5 chromosomes of 20 Mb (config 2's shape), class = chromosome
    contigs = [(f"g{g}", c, 20_000_000 - 20 - 1000 * g, c) for g in range(8) for c in range(5)]
    align = piece_alignment(100)
    assert align == 800
    for world in (1, 2, 3, 8):
        plan = plan_class_pieces(contigs, world)
        assert len(plan) == world and plan == plan_class_pieces(list(contigs), world)
        pieces = [p for sh in plan for p in sh]
        for name, ci, nk, _ in contigs:  # every contig is covered exactly once, in aligned pieces
            mine = sorted(p for p in pieces if p[0] == name and p[1] == ci)
            assert mine[0][2] == 0 and sum(p[3] for p in mine) == nk
            for a, b in zip(mine, mine[1:]):
                assert a[2] + a[3] == b[2] and b[2] % align == 0
        loads = [sum(p[3] for p in sh) for sh in plan]
        assert max(loads) <= 1.05 * sum(loads) / world  # 5 chromosomes on 8 ranks still balance
        for sh in plan:  # a rank holds the SAME pieces of every genome: its launch co-schedules all of them
            keys = {(p[4], p[5]) for p in sh}
            for key in keys:
                assert {p[0] for p in sh if (p[4], p[5]) == key} == {f"g{g}" for g in range(8)}
    # no CUT piece below the minimum length; a short member of a class of long ones (a scaffold that position pairing put
    # beside a chromosome) does not hold the class back: the long member is cut, the short one stays whole in piece 0
    mixed = [("a", 0, 30_000_000, 0), ("b", 0, 1_500_000, 0), ("a", 1, 40_000_000, 1), ("b", 1, 41_000_000, 1)]
    plan = plan_class_pieces(mixed, 4)
    c0 = sorted(p for sh in plan for p in sh if p[4] == 0)
    assert [p for p in c0 if p[0] == "b"] == [("b", 0, 0, 1_500_000, 0, 0)]
    a0 = [p for p in c0 if p[0] == "a"]
    assert len(a0) > 2 and a0[0][2] == 0 and sum(p[3] for p in a0) == 30_000_000
    assert all(x[2] + x[3] == y[2] and y[2] % align == 0 for x, y in zip(a0, a0[1:]))
    assert sum(1 for sh in plan for p in sh if p[4] == 1) > 2
    assert min(p[3] for sh in plan for p in sh) >= 1 << 20
    loads = [sum(p[3] for p in sh) for sh in plan]
    assert max(loads) <= 1.2 * sum(loads) / 4  # (one unit of 31.5 M on one rank before: 1.12 of ALL the work / 4 ... 1.0 of it)
    # a class whose members are ALL short stays one unit
    assert len([p for sh in plan_class_pieces([("a", 0, 900_000, 0), ("b", 0, 800_000, 0), ("a", 1, 9_000_000, 1),
                                                ("b", 1, 9_000_000, 1)], 2) for p in sh if p[4] == 0]) == 2


def _piece_pangenome(tmp_path):
    """3 genomes x (40 kb, 24 kb, 700 b, 15 b) with SNPs, an N run and lower case; the third genome names its records
    differently (paired by position); a GFF with genes that straddle piece boundaries on the first genome"""
    rng = np.random.default_rng(77)
    gen = po.synth_genomes(3, [40000, 24000, 700, 15], 0.02, 5)
    rows = ["name\tfasta\tgff"]
    for g, contigs in enumerate(gen):
        seqs = [po.codes_to_ascii(c) for c in contigs]
        if g == 1:
            seqs[0] = seqs[0][:15990] + b"N" * 30 + seqs[0][16020:20000] + seqs[0][20000:20400].lower() + seqs[0][20400:]
        names = [f"chr{ci + 1}" if g < 2 else f"CM00{ci}.1" for ci in range(len(seqs))]
        fa = tmp_path / f"g{g}.fa"
        fa.write_bytes(po.fasta_text(names, seqs, (80, 70, 61)[g]))
        gff = ""
        if g == 0:
            gff = str(tmp_path / "g0.gff")
            with open(gff, "w") as f:
                for st, en in ((100, 900), (15500, 16500), (19000, 21000), (31990, 32010), (39000, 39980), (39000, 50000)):
                    f.write(f"chr1\tx\tgene\t{st}\t{en}\t.\t+\t.\tID=g{st}\n")
                f.write("chr2\tx\tgene\t50\t23000\t.\t+\t.\tID=h\n")
        rows.append(f"g{g}\t{fa}\t{gff}")
    s = tmp_path / "samples.tsv"
    s.write_text("\n".join(rows) + "\n")
    return s


def _pieces_worker(rank, world, port, idx_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from panagram_amd import index as pidx
        from tests import fake_engine
        pidx.engine = fake_engine
        idx = pidx.Index(idx_dir, mode="w")
        assert (idx.rank, idx.world) == (rank, world) and idx.plan_sharding() == ("replicated", 1)
        idx.run()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_index_run_deals_pieces_of_homology_classes(world, tmp_path, monkeypatch):
    """Index.run() with several ranks and a table that fits: chromosomes are cut into aligned pieces, every rank
    anchors ITS pieces of EVERY genome (co-scheduled), fragments are assembled by whoever finds a genome complete.
    Decompressed bitmaps, bins, chrs, paircounts and gene histograms equal the one-rank run's."""
    import pandas as pd
    from panagram_amd import distributed as pdist
    from panagram_amd import index as pidx
    from tests import fake_engine
    s = _piece_pangenome(tmp_path)
    geo = dict(k=21, lowres_step=50, max_bin_kbp=3, min_bin_count=5)
    monkeypatch.setattr(pidx, "engine", fake_engine)
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setenv("RANK", "0")
    pidx.Index(str(s), prefix=str(tmp_path / "one"), **geo).run()
    pidx.Index(str(s), prefix=str(tmp_path / "many"), prepare=True, **geo)
    # the plan really cuts the long chromosomes and gives every rank pieces of every genome
    idx = pidx.Index(str(tmp_path / "many"), mode="w")
    seqs = {n: idx.seqset_for(n) for n in idx.anchor_genomes}
    cls = fake_engine.homology_classes([seqs[n].names for n in idx.anchor_genomes])
    contigs, c = [], 0
    for n in idx.anchor_genomes:
        for ci, ln in enumerate(seqs[n].lens):
            contigs.append((n, ci, max(0, int(ln) - 20), int(cls[c])))
            c += 1
    monkeypatch.setenv("PG_MIN_PIECE", "3000")  # (read by the workers at import: pieces of a few thousand positions)
    plan = pdist.plan_class_pieces(contigs, world, 50, min_piece=3000)
    assert all({p[0] for p in sh} == set(idx.anchor_genomes) for sh in plan)
    assert max(p[5] for sh in plan for p in sh) >= 1 and all(p[2] % 800 == 0 for sh in plan for p in sh)
    assert any(p[2] % 3000 for sh in plan for p in sh)  # pieces that do NOT start on a bin boundary
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_pieces_worker, args=(world, port, str(tmp_path / "many")), nprocs=world, join=True)
    for g in range(3):
        a, b = tmp_path / "one" / "anchor" / f"g{g}", tmp_path / "many" / "anchor" / f"g{g}"
        for step in (1, 50):
            assert gzip.open(a / f"bitmap.{step}.gz", "rb").read() == gzip.open(b / f"bitmap.{step}.gz", "rb").read()
            blocks = pidx.load_bgz_blocks(str(b / f"bitmap.{step}.gzi"))
            raw = gzip.open(b / f"bitmap.{step}.gz", "rb").read()
            for start in (0, 3999, 4000, len(raw) - 7):  # the .gzi of the concatenated fragments addresses the payload
                assert pidx.bgzf_read(str(b / f"bitmap.{step}.gz"), blocks, start, 7) == raw[start:start + 7]
        for t in ("bitsum.bins.tsv", "chrs.tsv", "total_paircounts.csv"):
            assert (a / t).read_bytes() == (b / t).read_bytes(), (g, t)
        assert not (b / ".parts").exists()
    assert (tmp_path / "one" / "anchor" / "g0" / "bitsum.genes.tsv").read_bytes() == \
        (tmp_path / "many" / "anchor" / "g0" / "bitsum.genes.tsv").read_bytes()
    genes = pd.read_table(tmp_path / "many" / "anchor" / "g0" / "bitsum.genes.tsv", index_col="chr")
    assert genes.loc["chr1"].sum() > 0


def _small_pangenome(tmp_path, lens_by_genome, seed=5):
    ncontigs = max(len(x) for x in lens_by_genome)
    longest = [max(x[ci] for x in lens_by_genome if ci < len(x)) for ci in range(ncontigs)]
    gen = po.synth_genomes(len(lens_by_genome), longest, 0.02, seed)
    rows = ["name\tfasta"]
    for g, lens in enumerate(lens_by_genome):
        seqs = [po.codes_to_ascii(c)[:ln] for c, ln in zip(gen[g], lens)]
        fa = tmp_path / f"g{g}.fa"
        fa.write_bytes(po.fasta_text([f"chr{ci + 1}" for ci in range(len(seqs))], seqs, 70))
        rows.append(f"g{g}\t{fa}")
    s = tmp_path / "samples.tsv"
    s.write_text("\n".join(rows) + "\n")
    return s


def _trees_equal(a_root, b_root, genomes, steps=(1, 50)):
    for g in genomes:
        a, b = a_root / "anchor" / g, b_root / "anchor" / g
        for step in steps:
            assert gzip.open(a / f"bitmap.{step}.gz", "rb").read() == gzip.open(b / f"bitmap.{step}.gz", "rb").read(), (g, step)
            assert (b / f"bitmap.{step}.gzi").exists()
        for t in ("bitsum.bins.tsv", "chrs.tsv", "total_paircounts.csv"):
            assert (a / t).read_bytes() == (b / t).read_bytes(), (g, t)
        assert not (b / ".parts").exists(), g


@pytest.mark.parametrize("world", [2, 3])
def test_an_anchor_genome_without_kmer_positions_still_gets_its_files(world, tmp_path, monkeypatch):
    """An anchor genome whose contigs are all shorter than k (or an empty FASTA) has no piece, so no rank's marker ever
    completes it: one rank writes its (empty) bitmaps and tables, as the one-rank run does — it used to be left without
    an output directory, silently."""
    from panagram_amd import index as pidx
    from tests import fake_engine
    s = _small_pangenome(tmp_path, [[30000, 9000], [15], [30000, 9000, 12], [18, 7]])
    geo = dict(k=21, lowres_step=50, max_bin_kbp=3, min_bin_count=5)
    monkeypatch.setattr(pidx, "engine", fake_engine)
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setenv("RANK", "0")
    pidx.Index(str(s), prefix=str(tmp_path / "one"), **geo).run()
    pidx.Index(str(s), prefix=str(tmp_path / "many"), prepare=True, **geo)
    monkeypatch.setenv("PG_MIN_PIECE", "3000")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_pieces_worker, args=(world, port, str(tmp_path / "many")), nprocs=world, join=True)
    _trees_equal(tmp_path / "one", tmp_path / "many", [f"g{g}" for g in range(4)])
    assert gzip.open(tmp_path / "many" / "anchor" / "g1" / "bitmap.1.gz", "rb").read() == b""
    # ... and with the ranks run by hand one after the other (no process group, no barrier)
    pidx.Index(str(s), prefix=str(tmp_path / "byhand"), prepare=True, **geo)
    from panagram_amd import distributed as pdist
    monkeypatch.setattr(pdist, "MIN_PIECE", 3000)
    monkeypatch.setenv("WORLD_SIZE", str(world))
    for rank in range(world):
        monkeypatch.setenv("RANK", str(rank))
        pidx.Index(str(tmp_path / "byhand"), mode="w").run()
    _trees_equal(tmp_path / "one", tmp_path / "byhand", [f"g{g}" for g in range(4)])


def test_a_run_over_an_aborted_runs_fragments_writes_the_same_tree(tmp_path, monkeypatch):
    """Rank 0 of a two-rank run dies after its pieces (fragments and markers of this run's signature stay behind); the
    run is started again: the ranks write their units anew (under temporary names, renamed when complete), the genome is
    assembled once, nothing is left in .parts — also when a marker vanishes under the assembler (its rank resuming)."""
    from panagram_amd import distributed as pdist
    from panagram_amd import index as pidx
    from tests import fake_engine
    s = _small_pangenome(tmp_path, [[30000, 9000], [30000, 9000], [29000, 9000]])
    geo = dict(k=21, lowres_step=50, max_bin_kbp=3, min_bin_count=5)
    monkeypatch.setattr(pidx, "engine", fake_engine)
    monkeypatch.setattr(pdist, "MIN_PIECE", 3000)
    monkeypatch.setenv("PG_MIN_PIECE", "3000")
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setenv("RANK", "0")
    pidx.Index(str(s), prefix=str(tmp_path / "one"), **geo).run()
    pidx.Index(str(s), prefix=str(tmp_path / "many"), prepare=True, **geo)
    monkeypatch.setenv("WORLD_SIZE", "2")
    pidx.Index(str(tmp_path / "many"), mode="w").run()  # rank 0 alone: its units, nothing assembled
    parts = tmp_path / "many" / "anchor" / "g0" / ".parts"
    stale = sorted(f for f in os.listdir(parts) if f.endswith(".npz"))
    assert stale and not (tmp_path / "many" / "anchor" / "g0" / "chrs.tsv").exists()
    assert not [f for f in os.listdir(parts) if f.endswith(".tmp")]
    # ... and an aborted run of ANOTHER signature (three ranks: other pieces, other base names) left its units there too:
    # nobody will read them; the assembly clears them away instead of leaving the directory behind with GBs in it
    with np.load(parts / stale[0]) as z:
        foreign = {x: z[x] for x in z.files}
    foreign["sig"] = np.array("a run with three ranks")
    np.savez(parts / "0.777.npz", **foreign)
    for sfx in ("1.gz", "1.gzi", "50.gz", "50.gzi"):
        (parts / f"0.777.{sfx}").write_bytes(b"left behind")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_pieces_worker, args=(2, port, str(tmp_path / "many")), nprocs=2, join=True)
    _trees_equal(tmp_path / "one", tmp_path / "many", ["g0", "g1", "g2"])
    # a marker that is gone when the assembler comes to remove it does not stop the assembly
    pidx.Index(str(s), prefix=str(tmp_path / "again"), prepare=True, **geo)
    for rank in (0, 1):
        monkeypatch.setenv("RANK", str(rank))
        if rank == 1:
            real = pdist._remove_quietly
            gone = []

            def remove_twice(path, real=real, gone=gone):
                if path.endswith(".npz") and not gone:
                    os.remove(path)  # (what a resuming rank does to its stale marker under the assembler's feet)
                    gone.append(path)
                real(path)
            monkeypatch.setattr(pdist, "_remove_quietly", remove_twice)
        pidx.Index(str(tmp_path / "again"), mode="w").run()
    assert gone
    _trees_equal(tmp_path / "one", tmp_path / "again", ["g0", "g1", "g2"])


def _claim_racer(path, rounds, start, done, q):
    from panagram_amd.distributed import _claim
    wins = []
    for r in range(rounds):
        start.wait()
        fd = _claim(f"{path}.{r}")
        wins.append(fd is not None)
        done.wait()  # the winner holds its claim until every racer has had its go
        if fd is not None:
            os.close(fd)
    q.put(wins)


@pytest.mark.parametrize("stale", [False, True])
def test_the_assembly_claim_has_one_winner(stale, tmp_path):
    """Eight processes go for the same claim at the same moment, 400 times over: exactly one gets each.  (A claim created
    empty and filled afterwards let a second rank read it in between and take it for a dead process's: two ranks
    assembled one genome at once — seen on the GPU box under torchrun.)  ``stale``: every claim file is already there,
    left by a process that no longer exists; still one winner."""
    import multiprocessing as mp
    from panagram_amd.distributed import _claim
    ctx = mp.get_context("fork")
    n, rounds = 8, 400
    path = str(tmp_path / "assemble.lock")
    if stale:
        for r in range(rounds):
            with open(f"{path}.{r}", "w") as f:
                f.write("left behind")
    start, done, q = ctx.Barrier(n), ctx.Barrier(n), ctx.Queue()
    procs = [ctx.Process(target=_claim_racer, args=(path, rounds, start, done, q)) for _ in range(n)]
    for p in procs:
        p.start()
    wins = np.array([q.get(timeout=120) for _ in procs])
    for p in procs:
        p.join(30)
    assert (wins.sum(axis=0) == 1).all(), np.flatnonzero(wins.sum(axis=0) != 1)
    fd = _claim(path + ".0")  # released with the descriptor
    assert fd is not None and _claim(path + ".0") is None
    os.remove(path + ".0")
    os.close(fd)
    assert _claim(str(tmp_path / "gone" / "assemble.lock")) is None  # (the genome's .parts directory cleaned up)


def _fragmented_pangenome(tmp_path):
    """3 draft assemblies: one 30-kb scaffold and 60 contigs of 300..900 b each, SNPs between them; the third genome lists
    its contigs in another order (paired by record id); genes on a long and on two short contigs of the first genome"""
    rng = np.random.default_rng(11)
    lens = [30000] + [int(x) for x in rng.integers(300, 900, 60)]
    gen = po.synth_genomes(3, lens, 0.02, 9)
    rows = ["name\tfasta\tgff"]
    for g, contigs in enumerate(gen):
        seqs = [po.codes_to_ascii(c) for c in contigs]
        names = [f"ctg{ci:03d}" for ci in range(len(seqs))]
        order = list(range(len(seqs)))
        if g == 2:
            order = order[:5] + order[40:] + order[5:40]
        fa = tmp_path / f"g{g}.fa"
        fa.write_bytes(po.fasta_text([names[i] for i in order], [seqs[i] for i in order], (80, 70, 61)[g]))
        gff = ""
        if g == 0:
            gff = str(tmp_path / "g0.gff")
            with open(gff, "w") as f:
                for chrom, st, en in (("ctg000", 100, 900), ("ctg000", 11900, 12100), ("ctg007", 10, 200), ("ctg033", 50, 260)):
                    f.write(f"{chrom}\tx\tgene\t{st}\t{en}\t.\t+\t.\tID=g{st}\n")
        rows.append(f"g{g}\t{fa}\t{gff}")
    s = tmp_path / "samples.tsv"
    s.write_text("\n".join(rows) + "\n")
    return s


@pytest.mark.parametrize("world", [2, 3])
def test_fragmented_assemblies_are_dealt_in_bundles_and_written_in_runs(world, tmp_path, monkeypatch):
    """Small homology classes go to the ranks in bundles of consecutive classes, and a rank's neighbouring whole contigs
    leave as one fragment with one marker (contig by contig, two ranks took 35 s for 4 x 4 000 contigs on the GPU box
    against 0.4 s on one).  The long scaffold is still cut into pieces; a genome that orders its contigs differently gets
    shorter runs.  The outputs equal the one-rank run's."""
    from panagram_amd import distributed as pdist
    from panagram_amd import index as pidx
    from tests import fake_engine
    s = _fragmented_pangenome(tmp_path)
    geo = dict(k=21, lowres_step=50, max_bin_kbp=3, min_bin_count=5)
    monkeypatch.setattr(pidx, "engine", fake_engine)
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setenv("RANK", "0")
    pidx.Index(str(s), prefix=str(tmp_path / "one"), **geo).run()
    pidx.Index(str(s), prefix=str(tmp_path / "many"), prepare=True, **geo)
    idx = pidx.Index(str(tmp_path / "many"), mode="w")
    seqs = {n: idx.seqset_for(n) for n in idx.anchor_genomes}
    cls = fake_engine.homology_classes([seqs[n].names for n in idx.anchor_genomes])
    contigs, c = [], 0
    for n in idx.anchor_genomes:
        for ci, ln in enumerate(seqs[n].lens):
            contigs.append((n, ci, max(0, int(ln) - 20), int(cls[c])))
            c += 1
    monkeypatch.setenv("PG_MIN_PIECE", "3000")
    plan = pdist.plan_class_pieces(contigs, world, 50, min_piece=3000)
    # every position once; the scaffold cut; the small contigs of the first genome in a few runs per rank
    cover = {}
    for sh in plan:
        for p in sh:
            cover[(p[0], p[1])] = cover.get((p[0], p[1]), 0) + p[3]
    assert cover == {(n, ci): nk for n, ci, nk, _ in contigs if nk > 0}
    assert max(p[5] for sh in plan for p in sh if p[1] == 0) >= 1
    for sh in plan:
        mine = sorted(p[1] for p in sh if p[0] == "g0" and p[1] > 0)
        runs = 1 + sum(1 for a, b in zip(mine, mine[1:]) if b != a + 1) if mine else 0
        assert not mine or runs < len(mine), (mine, runs)  # (neighbours together; at this size a bundle holds 2-3 contigs — see the plan test below)
    loads = [sum(p[3] for p in sh) for sh in plan]
    assert max(loads) <= 1.5 * sum(loads) / world
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_pieces_worker, args=(world, port, str(tmp_path / "many")), nprocs=world, join=True)
    for g in range(3):
        a, b = tmp_path / "one" / "anchor" / f"g{g}", tmp_path / "many" / "anchor" / f"g{g}"
        for step in (1, 50):
            assert gzip.open(a / f"bitmap.{step}.gz", "rb").read() == gzip.open(b / f"bitmap.{step}.gz", "rb").read()
        for t in ("bitsum.bins.tsv", "chrs.tsv", "total_paircounts.csv") + (("bitsum.genes.tsv",) if g == 0 else ()):
            assert (a / t).read_bytes() == (b / t).read_bytes(), (g, t)
        assert not (b / ".parts").exists()


@pytest.mark.parametrize("world", [2, 8])
def test_plan_bundles_small_classes_at_scale(world):
    """4 assemblies of 4 000 contigs of 10 kb (+ one 30-Mb chromosome): the chromosome is cut, the contigs are dealt in
    bundles of a quarter of a rank's target share at most — a rank's contigs of a genome are a few dozen runs of
    neighbours, not every other contig — and the ranks' loads stay within 10 % of each other."""
    from panagram_amd import distributed as pdist
    G, C = 4, 4000
    contigs = [(f"g{g}", ci, 30_000_000 if ci == 0 else 9_980, ci) for g in range(G) for ci in range(C + 1)]
    plan = pdist.plan_class_pieces(contigs, world, 100)
    cover = {}
    for sh in plan:
        for p in sh:
            cover[(p[0], p[1])] = cover.get((p[0], p[1]), 0) + p[3]
    assert cover == {(n, ci): nk for n, ci, nk, _ in contigs}
    loads = [sum(p[3] for p in sh) for sh in plan]
    assert max(loads) <= 1.1 * sum(loads) / world
    assert len({(p[1], p[2]) for sh in plan for p in sh if p[0] == "g0" and p[1] == 0}) >= world  # the chromosome in pieces
    for sh in plan:
        for g in range(G):
            mine = sorted(p[1] for p in sh if p[0] == f"g{g}" and p[1] > 0)
            runs = 1 + sum(1 for a, b in zip(mine, mine[1:]) if b != a + 1) if mine else 0
            assert runs <= 40, (world, g, len(mine), runs)
