"""GPU: genome-sharded mode on one device — per-"rank" partial tables (disjoint genome bits),
rows-only anchoring, SUM-combine of the partial rows, statistics from the combined rows.
Must equal the fused single-table result and the reference's golden outputs."""
import numpy as np
import pytest
import torch

from oracle import pyoracle as po
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", ["columns", "sum"])
@pytest.mark.parametrize("name,world", [("n9_k21", 2), ("n65_k21", 4), ("n40_k31", 8)])
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
        assert np.array_equal(allb[:nb_cols].cpu().numpy(), po.extract_columns(part0, n, 0, per))
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
