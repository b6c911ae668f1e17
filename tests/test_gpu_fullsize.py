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
    del genomes
    torch.cuda.empty_cache()
    ctx.trim()
    # the headline workload exactly as bench.py runs it (all 8 genomes in ONE co-scheduled result) against the CPU oracle:
    # head of the first contig and tail of the last one of two genomes (cpp/anchor.cpp:112-195 restated in oracle/)
    st2 = _fullsize_pangenome_properties(ctx, G, contig_lens, k, 0.01, 1234, picks=(0, 5))
    assert 2.0e8 < st2["nkeys"] < 2.6e8


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


def _fullsize_pangenome_properties(ctx, G, contig_lens, k, d, seed, picks, sample_n=2_000_000):
    """A BASELINE config at FULL size, every genome anchored in one co-scheduled result (the bench's mode):
    size-independent properties over all of it + the CPU oracle on the first ``sample_n`` positions of the first contig
    and the last ``sample_n / 2`` of the last contig of the ``picks`` genomes, its k-mer DB built by brute force with torch (bench.sample_db_by_brute_force: no HIP kernel involved)."""
    import bench
    from oracle import coracle
    from panagram_amd import engine
    dev = torch.device("cuda", 0)
    pg = bench.Pangenome(ctx, dev, G, contig_lens, d, seed, k, keep_ascii=True)
    C = len(contig_lens)
    # two stretches per picked genome: the head of its first contig and the TAIL of its last one (the last tiles of a
    # late contig: the end of the co-schedule, a partial last tile, the last bin)
    tail_n = sample_n // 2
    where = [(g, 0, 0) for g in picks] + [(g, C - 1, contig_lens[C - 1] - tail_n) for g in picks]
    samples = [pg.ascii[g][ci][s0:s0 + (sample_n if ci == 0 else tail_n)] for g, ci, s0 in where]
    dbs = bench.sample_db_by_brute_force(pg.ascii, samples, k, G)
    samples_host = [s.cpu().numpy() for s in samples]
    pg.ascii = None
    torch.cuda.empty_cache()
    merged = engine.SeqSet.concat(ctx, pg.seqsets)
    res = engine.AnchorResult(pg.table, merged, colsums=True)
    res.coschedule(np.repeat(np.arange(G), C))
    res.run()
    ccs = res.contig_colsums().astype(np.int64)          # [G*C, G]
    nk = np.array([L - k + 1 for L in contig_lens], np.int64)
    nb = (G + 7) // 8
    total_bits = 0
    for g in range(G):
        own = ccs[g * C:(g + 1) * C]
        assert np.array_equal(own[:, g], nk), f"anchor genome {g} must hold every one of its own k-mers"
        assert (own <= nk[:, None]).all() and (own.sum(axis=0) > 0).all()
    # bins: every contig's histogram sums to its bin lengths; sum_p p * hist[p] == sum of the column sums
    for ci in range(G * C):
        _, _, bins, info = res.download(ci, want_bitmap1=False, want_bitmap100=False)
        n = int(nk[ci % C])
        assert info["nkmers"] == n and info["binlen"] == (200000 if n // 200000 >= 100 else n // 100)
        lens = np.minimum(info["binlen"], n - np.arange(info["nbins"], dtype=np.int64) * info["binlen"])
        assert np.array_equal(bins.sum(axis=1), lens)
        total_bits += int((bins.astype(np.int64) * np.arange(G + 1)).sum())
    assert total_bits == int(ccs.sum())
    # rows of the sampled genomes: bitmap.100 == bitmap.1[::100]; first sample_n rows == the CPU oracle's
    odbs = [coracle.OracleDB.from_arrays(kk, mm, k) for kk, mm in dbs]
    for (g, ci, s0), s in zip(where, samples_host):
        rows, rows100, _, _ = res.download(g * C + ci)
        assert rows.shape == (nk[ci], nb) and np.array_equal(rows100, rows[::100])
        want = coracle.write_bits(odbs, G, s, k)[0]
        assert s0 + len(want) == nk[ci] or ci == 0
        assert np.array_equal(rows[s0:s0 + len(want)], want), f"genome {g}, contig {ci} from {s0}: GPU rows differ from the CPU oracle's"
        del rows, rows100
    for o in odbs:
        o.close()
    # idempotence of the whole launch: same column sums again
    res.run()
    assert np.array_equal(res.contig_colsums().astype(np.int64), ccs)
    st = pg.stats
    res.close()
    merged.close()
    pg.close()
    ctx.trim()
    return st


def test_config3_fullsize_properties(ctx):
    """BASELINE.json configs[2]: 27 Arabidopsis-scale (~135 Mb) genomes, k=21, all 27 anchored (3.6e9 positions, 4-byte
    rows) on one GPU"""
    st = _fullsize_pangenome_properties(ctx, 27, [34_000_000, 23_000_000, 26_000_000, 21_000_000, 31_000_000], 21, 0.01,
                                        2718, picks=(0, 19))
    assert 6.5e8 < st["nkeys"] < 9.5e8


def test_config4_fullsize_properties(ctx):
    """BASELINE.json configs[3]: 64 synthetic 200 Mb genomes, k=31, all 64 anchored (1.28e10 positions, 8-byte rows:
    about 90 GB of table + 102 GB of rows in one GPU's HBM)"""
    st = _fullsize_pangenome_properties(ctx, 64, [20_000_000] * 10, 31, 0.005, 3141, picks=(0, 41), sample_n=1_000_000)
    assert 1.5e9 < st["nkeys"] < 2.4e9
