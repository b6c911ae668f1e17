"""GPU: size-independent properties at BASELINE.json's configs[1] size (8 x 100 Mb, k=21)
and at the multi-word shape of configs[3] (64 genomes, k=31, scaled down in length)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

COMP = bytes.maketrans(b"ACGT", b"TGCA")


def _build(ctx, G, contig_lens, k, d, seed):
    import bench
    from panagram_amd import engine
    dev = torch.device("cuda", 0)
    genomes = bench.synth_genomes_device(G, contig_lens, d, seed, dev)
    torch.cuda.synchronize()
    seqsets = []
    for g in range(G):
        ss = engine.SeqSet(ctx, contig_lens)
        for c, t in enumerate(genomes[g]):
            ss.load_dev(c, t.data_ptr(), t.numel())
        seqsets.append(ss)
    ctx.synchronize()
    tbl = engine.PanTable(ctx, k, G)
    for g in range(G):
        tbl.insert_seqset(g, seqsets[g])
    return genomes, seqsets, tbl


def _check_anchor_properties(tbl, ss, g, G, k, contig_lens):
    from panagram_amd import engine
    res = engine.AnchorResult(tbl, ss, colsums=True)
    res.run()
    cs = res.colsums().astype(np.int64)
    total = sum(L - k + 1 for L in contig_lens)
    # the anchor genome contains every one of its own k-mers
    assert cs[g] == total
    assert (cs <= total).all() and (cs > 0).all()
    popc_weighted = 0
    for c, L in enumerate(contig_lens):
        rows, rows100, bins, info = res.download(c)
        nk = L - k + 1
        assert info["nkmers"] == nk and rows.shape == (nk, (G + 7) // 8)
        # bitmap.100 is every 100th row of bitmap.1
        assert np.array_equal(rows100, rows[::100])
        # histogram: each bin sums to its length, all bins to nkmers
        assert bins.sum() == nk
        lens = np.minimum(info["binlen"], nk - np.arange(info["nbins"]) * info["binlen"])
        assert np.array_equal(bins.sum(axis=1), lens)
        # checksum of checksums: sum_p p*hist[p] == sum of all set bits == sum of column sums
        popc_weighted += int((bins.astype(np.int64) * np.arange(G + 1)).sum())
        if c == 0:
            bits = np.unpackbits(rows[:200000], axis=1, bitorder="little")[:, :G]
            assert bits[:, g].all()
            if info["binlen"] == 200000:
                assert np.array_equal(np.bincount(bits.sum(axis=1), minlength=G + 1), bins[0])
    assert popc_weighted == int(cs.sum())
    # idempotence: a second run gives identical bytes
    first = res.download(0)[0].copy()
    res.run()
    assert np.array_equal(res.download(0)[0], first)
    res.close()


def test_config2_fullsize_properties(ctx):
    G, k = 8, 21
    contig_lens = [20_000_000] * 5
    genomes, seqsets, tbl = _build(ctx, G, contig_lens, k, 0.01, 1234)
    st = tbl.stats()
    assert 2.0e8 < st["nkeys"] < 2.6e8  # ~2.3e8 union keys
    for g in (0, 5):
        _check_anchor_properties(tbl, seqsets[g], g, G, k, contig_lens)
    # strand symmetry: canonical k-mers => anchoring the reverse complement reverses the rows
    from panagram_amd import engine
    seq = bytes(genomes[3][1][:3_000_000].cpu().numpy())
    rc = seq.translate(COMP)[::-1]
    a = tbl.anchor_contig(seq, colsums=False)[0]
    b = tbl.anchor_contig(rc, colsums=False)[0]
    assert np.array_equal(a, b[::-1])
    # a denser table gives the same bytes
    before = tbl.anchor_contig(seq, colsums=False)[0]
    tbl.rehash(6.0)
    assert np.array_equal(tbl.anchor_contig(seq, colsums=False)[0], before)
    for s in seqsets:
        s.close()
    tbl.close()


@pytest.mark.parametrize("G,k", [(64, 31), (65, 31), (27, 21)])
def test_multiword_shapes_properties(ctx, G, k):
    """configs[3] shape (N=64 -> 8-byte rows, one 2-word sub-table; N=65 -> second sub-table)
    and configs[2] shape (27 genomes), shortened genomes."""
    contig_lens = [3_000_000, 1_000_000]
    genomes, seqsets, tbl = _build(ctx, G, contig_lens, k, 0.005, 77)
    for g in (0, G - 1):
        _check_anchor_properties(tbl, seqsets[g], g, G, k, contig_lens)
    for s in seqsets:
        s.close()
    tbl.close()


def test_more_than_2_32_positions_in_one_result(ctx):
    """Human-pangenome scale: one result over > 2^32 positions (byte offsets, tile counts and the
    co-scheduled order are 64-bit clean).  46 resident copies of a 2 x 100 Mb anchor are concatenated on
    the device; every copy must reproduce the single-copy rows, bins and column sums."""
    from panagram_amd import engine
    G, k, copies = 2, 21, 46
    contig_lens = [20_000_000] * 5
    genomes, seqsets, tbl = _build(ctx, G, contig_lens, k, 0.01, 77)
    tbl.rehash(2.0)
    nc = len(contig_lens)
    one = engine.AnchorResult(tbl, seqsets[1], colsums=True)
    one.run()
    ref_cs = one.contig_colsums()
    ref = {c: one.download(c) for c in (0, nc - 1)}
    big = engine.SeqSet.concat(ctx, [seqsets[1]] * copies)
    total = copies * sum(L - k + 1 for L in contig_lens)
    assert total > 2 ** 32
    for grouped in (False, True):
        res = engine.AnchorResult(tbl, big, colsums=True)
        if grouped:
            res.coschedule(np.repeat(np.arange(copies), nc), 64)
        res.run()
        cs = res.contig_colsums()
        assert np.array_equal(cs, np.tile(ref_cs, (copies, 1)))
        assert int(res.colsums()[1]) == total
        for copy in (0, copies // 2, copies - 1):
            for c in (0, nc - 1):
                rows, rows100, bins, info = res.download(copy * nc + c)
                assert np.array_equal(rows, ref[c][0]) and np.array_equal(rows100, ref[c][1])
                assert np.array_equal(bins, ref[c][2])
        res.close()  # (its 4.6 GB row buffer goes to the context's cache and serves the second pass)
    big.close()
    one.close()
    ctx.trim()  # give the cached row buffers back
