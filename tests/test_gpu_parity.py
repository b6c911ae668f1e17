"""GPU parity tests: the HIP path (through the C-ABI) vs the golden vectors the reference
binary produced and vs the oracle on seeded inputs.  Bit-exact everywhere (integer work)."""
import os

import numpy as np
import pytest

from oracle import pyoracle as po
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _table_from_dbs(ctx, fx, via_kmc1_files, tmp_path):
    from panagram_amd import engine
    n, k = int(fx["ngenomes"]), int(fx["k"])
    tbl = engine.PanTable(ctx, k, n)
    for i, (keys, masks) in enumerate(H.case_dbs(fx)):
        if via_kmc1_files:
            p = str(tmp_path / f"bitvec{i}")
            po.write_kmc1(p, keys, masks, k, min_count=int(fx["min_count"]), max_count=int(fx["max_count"]))
            tbl.load_kmc1(i, open(p + ".kmc_pre", "rb").read(), open(p + ".kmc_suf", "rb").read())
        else:
            sel = (masks >= int(fx["min_count"])) & (masks <= int(fx["max_count"]))
            tbl.insert_keys(i, keys[sel], masks[sel])
    return tbl


def _anchor_fasta_gpu(ctx, tbl, fasta_bytes, colsums=True):
    """Anchor all contigs of a FASTA in one launch; returns payloads + texts like run_anchor."""
    from panagram_amd import engine
    recs = po.parse_fasta_cpp(fasta_bytes)  # test-side parse; the product parser is tested elsewhere
    ss = engine.SeqSet.from_host(ctx, [s for _, s in recs])
    res = engine.AnchorResult(tbl, ss, colsums=colsums)
    res.run()
    b1, b100, bins, binlens, sizes = [], [], [], [], []
    for i in range(len(recs)):
        rows, rows100, bn, info = res.download(i)
        b1.append(rows.tobytes())
        b100.append(rows100.tobytes())
        bins.append(bn)
        binlens.append(info["binlen"])
        sizes.append(info["nkmers"])
    cs = res.colsums() if colsums else None
    chrs = "name\tid\tsize\tgene_count\n" + "".join(
        f"{nm}\t{i}\t{sz}\t0\n" for i, ((nm, _), sz) in enumerate(zip(recs, sizes)))
    out = dict(bitmap1=b"".join(b1), bitmap100=b"".join(b100),
               bins_tsv=H.bins_text(tbl.ngenomes, bins, binlens), chrs_tsv=chrs, colsums=cs)
    res.close()
    ss.close()
    return out


@pytest.mark.parametrize("via_files", [False, True])
@pytest.mark.parametrize("name", H.payload_cases())
def test_anchor_matches_reference_golden(ctx, name, via_files, tmp_path):
    fx = H.load_case(name)
    n, k = int(fx["ngenomes"]), int(fx["k"])
    tbl = _table_from_dbs(ctx, fx, via_files, tmp_path)
    dbs = H.case_dbs(fx)
    for g in fx["anchors"]:
        fa = fx[f"fasta_{g}"].tobytes()
        got = _anchor_fasta_gpu(ctx, tbl, fa)
        assert got["bitmap1"] == fx[f"a{g}_bitmap1"].tobytes(), "bitmap.1 payload differs from the reference"
        assert got["bitmap100"] == fx[f"a{g}_bitmap100"].tobytes(), "bitmap.100 payload differs"
        assert got["bins_tsv"].encode() == fx[f"a{g}_bitsum.bins.tsv"].tobytes()
        assert got["chrs_tsv"].encode() == fx[f"a{g}_chrs.tsv"].tobytes()
        ora = po.anchor_fasta(dbs, fa, k, n, int(fx["min_count"]), int(fx["max_count"]))
        assert np.array_equal(got["colsums"].astype(np.int64), ora["colsums"])
    tbl.close()


@pytest.mark.parametrize("name", H.payload_cases())
def test_gpu_set_construction_equals_kmc_semantics(ctx, name):
    """insert_seqset (replaces kmc + kmc_tools) -> export == the fixture's DB contents."""
    from panagram_amd import engine
    fx = H.load_case(name)
    if int(fx["min_count"]) != 1:
        pytest.skip("header filter case")
    n, k = int(fx["ngenomes"]), int(fx["k"])
    tbl = engine.PanTable(ctx, k, n)
    for g in range(n):
        recs = po.parse_fasta_cpp(fx[f"fasta_{g}"].tobytes())
        ss = engine.SeqSet.from_host(ctx, [s for _, s in recs])
        tbl.insert_seqset(g, ss)
        ss.close()
    total = 0
    for i, (fk, fm) in enumerate(H.case_dbs(fx)):
        keys, vals = tbl.export(i)
        o = np.argsort(keys)
        assert np.array_equal(keys[o], fk) and np.array_equal(vals[o], fm)
        total += len(fk)
    st = tbl.stats()
    assert st["nkeys"] <= total and st["nkeys"] >= max(len(k_) for k_, _ in H.case_dbs(fx))
    # after a re-hash to a tighter table the contents are unchanged
    tbl.rehash(5.0)
    for i, (fk, fm) in enumerate(H.case_dbs(fx)):
        keys, vals = tbl.export(i)
        o = np.argsort(keys)
        assert np.array_equal(keys[o], fk) and np.array_equal(vals[o], fm)
    tbl.close()


@pytest.mark.parametrize("name", ["n2_k21", "n40_k31", "n33_k16", "n3_k32", "n65_k21"])
def test_counters_for_read_equals_oracle(ctx, name, tmp_path):
    """the literal GetCountersForRead equivalent, per 32-genome group"""
    fx = H.load_case(name)
    k = int(fx["k"])
    tbl = _table_from_dbs(ctx, fx, False, tmp_path)
    dbs = H.case_dbs(fx)
    g = int(fx["anchors"][0])
    for _, seq in po.parse_fasta_cpp(fx[f"fasta_{g}"].tobytes()):
        for d in range(len(dbs)):
            want = po.counters_for_read(dbs[d], seq, k)
            got = tbl.counters_for_read(d, seq)
            assert np.array_equal(got, want)
    # shorter than k: no k-mers, no error
    assert len(tbl.counters_for_read(0, b"ACG")) == 0
    tbl.close()


@pytest.mark.parametrize("name", H.seed_cases())
def test_seeded_large_cases_sha256(ctx, name):
    """config-1 shape (2 x 1 Mb) and the 20 Mb / 102-bin case: tables built ON the GPU from the
    sequences, payload sha256 + TSV texts must equal what the reference binary produced."""
    from panagram_amd import engine
    fx = H.load_case(name)
    n, k = int(fx["ngenomes"]), int(fx["k"])
    genomes, fastas = H.regen_seed_case(fx)
    tbl = engine.PanTable(ctx, k, n)
    for g in range(n):
        ss = engine.SeqSet.from_host(ctx, genomes[g])
        tbl.insert_seqset(g, ss)
        ss.close()
    for g in fx["anchors"]:
        got = _anchor_fasta_gpu(ctx, tbl, fastas[g], colsums=False)
        assert len(got["bitmap1"]) == int(fx[f"a{g}_len_1"])
        assert H.sha(got["bitmap1"]) == str(fx[f"a{g}_sha_1"])
        assert H.sha(got["bitmap100"]) == str(fx[f"a{g}_sha_100"])
        assert got["bins_tsv"].encode() == fx[f"a{g}_bitsum.bins.tsv"].tobytes()
        assert got["chrs_tsv"].encode() == fx[f"a{g}_chrs.tsv"].tobytes()
    # same answers from a much denser table (70 % slot load: overflow chains + retry queue)
    tbl.rehash(6.0)
    g = int(fx["anchors"][0])
    got = _anchor_fasta_gpu(ctx, tbl, fastas[g], colsums=False)
    assert H.sha(got["bitmap1"]) == str(fx[f"a{g}_sha_1"])
    assert got["bins_tsv"].encode() == fx[f"a{g}_bitsum.bins.tsv"].tobytes()
    tbl.close()


def test_one_shot_contig_and_edge_cases(ctx):
    from panagram_amd import engine
    rng = np.random.default_rng(3)
    k, n = 21, 5
    gen = po.synth_genomes(n, [5000], 0.05, 99)
    genomes = [[po.codes_to_ascii(c) for c in g] for g in gen]
    dbs = po.build_bitvec_dbs(genomes, k)
    tbl = engine.PanTable(ctx, k, n)
    tbl.insert_keys(0, *dbs[0])
    seq = bytearray(genomes[2][0])
    seq[100:140] = b"N" * 40
    seq[3000:3200] = bytes(seq[3000:3200]).lower()
    seq = bytes(seq)
    rows, rows100, bins, cs = tbl.anchor_contig(seq)
    o_rows, o_rows100, o_bins, _, o_cs = po.anchor_contig(dbs, seq, k, n)
    assert np.array_equal(rows, o_rows) and np.array_equal(rows100, o_rows100)
    assert np.array_equal(bins.astype(np.int64), o_bins) and np.array_equal(cs.astype(np.int64), o_cs)
    # contig shorter than k: zero k-mers, no failure (reference underflows; documented)
    r = tbl.anchor_contig(b"ACGT")
    assert r[0].shape == (0, 1)
    # exactly k bases: one k-mer
    r = tbl.anchor_contig(genomes[0][0][:k])
    assert r[0].shape == (1, 1) and r[0][0, 0] & 1
    # all-N contig: all zero rows
    r = tbl.anchor_contig(b"N" * 300)
    assert not r[0].any() and r[2][:, 0].sum() == 300 - k + 1
    # sizes around multiples of the tile (PROBE_TILE = 512 positions: 2048 and 4096 are tile boundaries too)
    for L in (2048 + k - 1, 2049 + k - 1, 4096 + k - 1 - 1):
        s = genomes[1][0][:L]
        got = tbl.anchor_contig(s)
        want = po.anchor_contig(dbs, s, k, n)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[2].astype(np.int64), want[2])
    tbl.close()


_KW = [(k, None) for k in (1, 7, 15, 19, 20, 22, 25, 27, 30, 32)] + \
      [(k, w) for k in (20, 21, 26, 31, 32) for w in (0, 3, 4, 5, 6, 7, 8)]


@pytest.mark.parametrize("k,w", _KW)
def test_every_minimizer_window_and_direct_mode(ctx, k, w):
    """The table places a k-mer (k >= 20) by the smallest of its w = 3..8 canonical m-mers, any other
    k (or m = 0) by the k-mer itself; m-mers over 16 bases take the 64-bit path.  Every window is
    pinned in turn (w None: the library's own choice); tables built on the GPU, answers compared
    with the oracle (bit-exact)."""
    from panagram_amd import engine
    n = 3
    gen = po.synth_genomes(n, [6000, 1500], 0.03, 500 + k)
    genomes = [[po.codes_to_ascii(c) for c in g] for g in gen]
    seq = bytearray(genomes[1][0])
    seq[700:720] = b"N" * 20          # N run
    seq[2000:2300] = bytes(seq[2000:2300]).lower()
    genomes[1][0] = bytes(seq)
    dbs = po.build_bitvec_dbs(genomes, k)
    tbl = engine.PanTable(ctx, k, n)
    if w is not None:
        tbl.set_minimizer(k - w + 1 if w else 0)
    elif k >= 20:
        assert 3 <= k - tbl.minimizer + 1 <= 8
    else:
        assert tbl.minimizer == 0
    for g in range(n):
        ss = engine.SeqSet.from_host(ctx, genomes[g])
        tbl.insert_seqset(g, ss)
        ss.close()
    keys, vals = tbl.export(0)
    o = np.argsort(keys)
    assert np.array_equal(keys[o], dbs[0][0]) and np.array_equal(vals[o], dbs[0][1])
    m_before = tbl.minimizer
    tbl.rehash(5.0)  # dense: exercises the overflow queue and the inline chase
    if w is not None:
        assert tbl.minimizer == m_before  # pinned
    for g in (0, 1):
        for seq_ in genomes[g]:
            rows, rows100, bins, cs = tbl.anchor_contig(seq_)
            o_rows, o_rows100, o_bins, _, o_cs = po.anchor_contig(dbs, seq_, k, n)
            assert np.array_equal(rows, o_rows) and np.array_equal(rows100, o_rows100)
            assert np.array_equal(bins.astype(np.int64), o_bins) and np.array_equal(cs.astype(np.int64), o_cs)
    tbl.close()


def test_minimizer_settles_on_the_first_sequence_set(ctx):
    """A table created for many keys picks m from the key count (k=21, 3e8 keys: m = 16); the first sequence set that
    goes into the EMPTY table tells the pangenome's non-redundant length and m is settled again from it (short
    genomes: the widest window the key rule allows, m = 15 — never more than two bases under the key rule); a pinned
    m stays; pg_table_clear starts over.  Answers are the oracle's either way."""
    from panagram_amd import engine
    k, n = 21, 20
    gen = po.synth_genomes(n, [5000, 900], 0.02, 77)
    genomes = [[po.codes_to_ascii(c) for c in g] for g in gen]
    dbs = po.build_bitvec_dbs(genomes, k)

    def fill_and_check(tbl):
        for g in range(n):
            ss = engine.SeqSet.from_host(ctx, genomes[g])
            tbl.insert_seqset(g, ss)
            ss.close()
        for seq_ in genomes[7]:
            rows = tbl.anchor_contig(seq_)[0]
            assert np.array_equal(rows, po.anchor_contig(dbs, seq_, k, n)[0])

    tbl = engine.PanTable(ctx, k, n, expected_keys=300_000_000)
    assert tbl.minimizer == 16
    fill_and_check(tbl)
    assert tbl.minimizer == 15
    tbl.clear()
    fill_and_check(tbl)
    assert tbl.minimizer == 15
    tbl.close()
    tbl = engine.PanTable(ctx, k, n, expected_keys=300_000_000)
    tbl.set_minimizer(18)
    fill_and_check(tbl)
    assert tbl.minimizer == 18
    tbl.close()


def test_window_cap_environment_variable():
    """PG_TABLE_WMAX caps the minimizer window the library chooses (read once per process: checked in a child)."""
    import subprocess
    import sys
    code = ("from panagram_amd import engine\n"
            "c = engine.Context(0)\n"
            "print(engine.PanTable(c, 21, 27).minimizer, engine.PanTable(c, 31, 27).minimizer)\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for cap, want in (("4", "18 28"), ("6", "16 26"), ("", "16 24")):  # (no key count yet: m >= 16)
        env = dict(os.environ, PG_TABLE_WMAX=cap) if cap else {k: v for k, v in os.environ.items() if k != "PG_TABLE_WMAX"}
        out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        assert out.stdout.split("\n")[-2].strip() == want, (cap, out.stdout)


def test_three_subtables_n130(ctx):
    """N = 130 -> 5 bitvec groups -> 3 sub-tables (64 + 64 + 2 genomes), 17-byte rows"""
    from panagram_amd import engine
    n, k = 130, 21
    gen = po.synth_genomes(n, [1500], 0.01, 4242)
    genomes = [[po.codes_to_ascii(c) for c in g] for g in gen]
    dbs = po.build_bitvec_dbs(genomes, k)
    assert len(dbs) == 5
    tbl = engine.PanTable(ctx, k, n)
    for g in range(n):
        ss = engine.SeqSet.from_host(ctx, genomes[g])
        tbl.insert_seqset(g, ss)
        ss.close()
    for d, (fk, fm) in enumerate(dbs):
        keys, vals = tbl.export(d)
        o = np.argsort(keys)
        assert np.array_equal(keys[o], fk) and np.array_equal(vals[o], fm)
    for g in (0, 64, 129):
        seq = genomes[g][0]
        rows, rows100, bins, cs = tbl.anchor_contig(seq)
        o_rows, o_rows100, o_bins, _, o_cs = po.anchor_contig(dbs, seq, k, n)
        assert rows.shape[1] == 17
        assert np.array_equal(rows, o_rows) and np.array_equal(rows100, o_rows100)
        assert np.array_equal(bins.astype(np.int64), o_bins) and np.array_equal(cs.astype(np.int64), o_cs)
    tbl.close()


@pytest.mark.parametrize("n,k", [(70, 21), (96, 31), (65, 15)])
def test_inline_layout_table_operations(ctx, n, k, tmp_path):
    """65..96 genomes: the inline table layout (6 bare keys + their mask blocks per 128-byte line) through every operation that
    touches a table — wave-cooperative build, export per 32-genome group, key-array import, re-hash to a dense table (long
    chains, the overflow levels and the lane-by-lane chase), update-only insertion, GetCountersForRead look-ups, the KMC import
    and a cleared table built again — against the oracle's databases and rows (cpp/anchor.cpp:138-165; index.py:934-945).
    k = 15: direct mode (no minimizer)."""
    from panagram_amd import engine
    gen = po.synth_genomes(n, [9000, 2500], 0.02, 5151 + n)
    genomes = [[po.codes_to_ascii(c) for c in g] for g in gen]
    dbs = po.build_bitvec_dbs(genomes, k)
    tbl = engine.PanTable(ctx, k, n)
    sets = [engine.SeqSet.from_host(ctx, g) for g in genomes]
    for g in range(n):
        tbl.insert_seqset(g, sets[g])
    st = tbl.stats()
    assert st["nslots"] // st["nbuckets"] == 6, st  # the inline layout: six keys per line

    def check_export(t):
        for d, (fk, fm) in enumerate(dbs):
            keys, vals = t.export(d)
            o = np.argsort(keys)
            assert np.array_equal(keys[o], fk) and np.array_equal(vals[o], fm), d

    def check_rows(t, who):
        for g in who:
            for seq in genomes[g]:
                rows, rows100, bins, cs = t.anchor_contig(seq)
                o_rows, o_rows100, o_bins, _, o_cs = po.anchor_contig(dbs, seq, k, n)
                assert np.array_equal(rows, o_rows) and np.array_equal(rows100, o_rows100)
                assert np.array_equal(bins.astype(np.int64), o_bins) and np.array_equal(cs.astype(np.int64), o_cs)

    check_export(tbl)
    check_rows(tbl, (0, n - 1))
    for d in range(len(dbs)):  # GetCountersForRead per 32-genome group (single-lane look-ups)
        got = tbl.counters_for_read(d, genomes[1][0])
        assert np.array_equal(got, po.counters_for_read(dbs[d], genomes[1][0], k)), d
    tbl.rehash(4.5)  # 4.5 of 6 slots per line: most groups spill, chains of several lines
    check_export(tbl)
    check_rows(tbl, (n // 2,))
    # a second table from key arrays (k_insert_keys), then a cleared table built again with update-only passes
    t2 = engine.PanTable(ctx, k, n)
    for d, (fk, fm) in enumerate(dbs):
        t2.insert_keys(d, fk, fm)
    check_export(t2)
    check_rows(t2, (3,))
    t2.clear()
    t2.insert_seqset(0, sets[0])
    for g in range(1, n):
        t2.update_seqset(g, sets[g])  # bits for the keys genome 0 holds, no new keys
    rows = t2.anchor_contig(genomes[0][0])[0]
    assert np.array_equal(rows, po.anchor_contig(dbs, genomes[0][0], k, n)[0])  # an anchor only asks for its own k-mers
    t2.close()
    # KMC1 files of the groups (the writer's own prefix length), imported on the GPU (k_import_kmc)
    t3 = engine.PanTable(ctx, k, n)
    for d, (fk, fm) in enumerate(dbs):
        pfx = str(tmp_path / f"bitvec{d}")
        po.write_kmc1(pfx, fk, fm, k)
        t3.load_kmc_files(d, pfx)
    check_rows(t3, (n - 1,))
    t3.close()
    for ss in sets:
        ss.close()
    tbl.close()


@pytest.mark.parametrize("n", [12, 20, 27, 32, 44, 52, 80, 88, 96, 120])  # (80 / 88 / 120: 10-, 11- and 15-byte rows, whose tail is a dword overlapping the words before it)
def test_rows_of_whole_words(ctx, n):
    """4- and 12-byte rows (N = 25..32, 89..96) leave the probe kernel as aligned 32-bit stores and go
    through the carry-save column sums: compare everything with the oracle"""
    from panagram_amd import engine
    k = 21
    gen = po.synth_genomes(n, [5000, 700], 0.02, 900 + n)
    genomes = [[po.codes_to_ascii(c) for c in g] for g in gen]
    dbs = po.build_bitvec_dbs(genomes, k)
    tbl = engine.PanTable(ctx, k, n)
    for g in range(n):
        ss = engine.SeqSet.from_host(ctx, genomes[g])
        tbl.insert_seqset(g, ss)
        ss.close()
    for g in (0, n // 2, n - 1):
        for seq in genomes[g]:
            rows, rows100, bins, cs = tbl.anchor_contig(seq)
            o_rows, o_rows100, o_bins, _, o_cs = po.anchor_contig(dbs, seq, k, n)
            assert rows.shape[1] == (n + 7) // 8
            assert np.array_equal(rows, o_rows) and np.array_equal(rows100, o_rows100)
            assert np.array_equal(bins.astype(np.int64), o_bins) and np.array_equal(cs.astype(np.int64), o_cs)
    tbl.close()


def test_errors_are_loud(ctx):
    from panagram_amd import engine
    with pytest.raises(engine.PanagramHipError):
        engine.PanTable(ctx, 33, 2)  # k > 32
    tbl = engine.PanTable(ctx, 21, 2)
    with pytest.raises(engine.PanagramHipError):
        tbl.set_minimizer(5)  # window of 17 m-mers
    with pytest.raises(engine.PanagramHipError):
        tbl.load_kmc1(0, b"garbage-not-a-kmc-file" * 8, b"KMCSKMCS")
    with pytest.raises(engine.PanagramHipError):
        tbl.insert_keys(3, np.zeros(1, np.uint64), np.ones(1, np.uint32))  # db index out of range
    tbl.close()


def test_chain_overflow_and_growth(ctx):
    """Many keys into a table created tiny: growth + bucket-overflow chains stay exact."""
    from panagram_amd import engine
    rng = np.random.default_rng(17)
    k, n = 31, 64
    keys = np.unique(rng.integers(0, 1 << 62, 400000, dtype=np.uint64))
    seqs = []
    m0 = rng.integers(1, 1 << 32, len(keys), dtype=np.uint64).astype(np.uint32)
    m1 = rng.integers(0, 1 << 32, len(keys), dtype=np.uint64).astype(np.uint32)
    tbl = engine.PanTable(ctx, k, n, expected_keys=1000)
    tbl.insert_keys(0, keys, m0)
    tbl.insert_keys(1, keys[::2], m1[::2])
    tbl.rehash(6.2)  # dense: ~78 % of the 4-slot buckets' capacity -> long chains
    for d, (kk, mm) in enumerate([(keys, m0), (keys[::2], m1[::2])]):
        ek, ev = tbl.export(d)
        o = np.argsort(ek)
        nz = mm != 0
        assert np.array_equal(ek[o], kk[nz]) and np.array_equal(ev[o], mm[nz])
    tbl.close()


def _simulate_reads(rng, genome: bytes, nreads: int, rlen: int, err: float):
    g = np.frombuffer(genome, np.uint8)
    reads = []
    for _ in range(nreads):
        p = int(rng.integers(0, len(g) - rlen))
        r = g[p:p + rlen].copy()
        e = rng.random(rlen) < err
        r[e] = rng.choice(np.frombuffer(b"ACGTN", np.uint8), size=int(e.sum()))
        if rng.random() < 0.5:  # reverse strand
            comp = np.zeros(256, np.uint8)
            comp[list(b"ACGTN")] = list(b"TGCAN")
            r = comp[r][::-1]
        reads.append(r.tobytes())
    return reads


@pytest.mark.parametrize("k,min_count", [(21, 2), (31, 2), (15, 3)])
def test_min_count_insertion_equals_kmc_ci(ctx, k, min_count):
    """kmc -ci<c> (the reference counts FASTQ samples with -ci2): canonical k-mers seen fewer than c
    times in the read set do not enter the table; sample 1 is an assembly (-ci1)."""
    from panagram_amd import engine
    rng = np.random.default_rng(5 + k)
    gen = po.synth_genomes(2, [12000], 0.02, 77 + k)
    asm = [po.codes_to_ascii(c) for c in gen[1]]
    reads = _simulate_reads(rng, po.codes_to_ascii(gen[0][0]), 400, 150, 0.01)
    dbs = po.build_bitvec_dbs([reads, asm], k, min_counts=[min_count, 1])
    tbl = engine.PanTable(ctx, k, 2)
    ss = engine.SeqSet.from_host(ctx, [b"N".join(reads)])
    tbl.insert_seqset(0, ss, min_count=min_count)
    ss.close()
    ss = engine.SeqSet.from_host(ctx, asm)
    tbl.insert_seqset(1, ss)
    ss.close()
    keys, vals = tbl.export(0)
    o = np.argsort(keys)
    assert np.array_equal(keys[o], dbs[0][0]) and np.array_equal(vals[o], dbs[0][1])
    # and the same read set at -ci1 holds strictly more keys (the singletons)
    tbl1 = engine.PanTable(ctx, k, 2)
    ss = engine.SeqSet.from_host(ctx, [b"N".join(reads)])
    tbl1.insert_seqset(0, ss)
    ss.close()
    assert tbl1.stats()["nkeys"] > int((dbs[0][1] & 1).sum())
    tbl1.close()
    tbl.close()


@pytest.mark.parametrize("n,k", [(5, 21), (40, 31), (65, 21)])
def test_window_stats_equal_oracle(ctx, n, k):
    """per-window occupancy histograms and column sums (genes, bins of any length) from rows in HBM"""
    from panagram_amd import engine
    rng = np.random.default_rng(n)
    gen = po.synth_genomes(n, [90000, 3000], 0.03, 400 + n)
    genomes = [[po.codes_to_ascii(c) for c in g] for g in gen]
    dbs = po.build_bitvec_dbs(genomes, k)
    tbl = engine.PanTable(ctx, k, n)
    for g in range(n):
        ss = engine.SeqSet.from_host(ctx, genomes[g])
        tbl.insert_seqset(g, ss)
        ss.close()
    ss = engine.SeqSet.from_host(ctx, genomes[1])
    res = engine.AnchorResult(tbl, ss)
    res.run()
    for ci, seq in enumerate(genomes[1]):
        o_rows, o_rows100 = po.anchor_contig(dbs, seq, k, n)[:2]
        nk = len(o_rows)
        starts = np.concatenate([rng.integers(0, nk, 40), [0, 0, nk - 1, nk, 5]])
        lens = np.concatenate([rng.integers(0, 5000, 40), [nk, 1, 1, 10, 0]])
        ends = starts + lens  # some run past the contig end (clipped), some are empty
        h, cs = res.window_stats(ci, starts, ends)
        oh, ocs = po.window_stats(o_rows, n, starts, ends)
        assert np.array_equal(h.astype(np.int64), oh) and np.array_equal(cs.astype(np.int64), ocs)
        # one window over everything, long enough to be split into pieces; and the 1-in-100 rows
        h, cs = res.window_stats(ci, [0], [nk], colsums=False)
        assert cs is None and np.array_equal(h.astype(np.int64), po.window_stats(o_rows, n, [0], [nk])[0])
        s100 = np.arange(0, len(o_rows100), 37)
        h, cs = res.window_stats(ci, s100, s100 + 50, step=100)
        oh, ocs = po.window_stats(o_rows100, n, s100, s100 + 50)
        assert np.array_equal(h.astype(np.int64), oh) and np.array_equal(cs.astype(np.int64), ocs)
    res.close()
    ss.close()
    tbl.close()


@pytest.mark.parametrize("k", [21, 31, 12])
def test_repeat_family_heavy_minimizer_groups(ctx, k):
    """Hundreds of diverged copies of one element put hundreds of distinct k-mers behind one
    minimizer: beyond GROUP_CHAIN lines the keys continue on their own probe sequence.  Table
    contents, GetCountersForRead and anchored rows stay bit-exact."""
    from panagram_amd import engine
    rng = np.random.default_rng(k)
    n = 3
    elem = rng.integers(0, 4, 400, dtype=np.uint8)
    genomes = []
    for g in range(n):
        parts = []
        for c in range(500):
            e = elem.copy()
            mut = rng.random(len(e)) < 0.04
            e[mut] = (e[mut] + rng.integers(1, 4, int(mut.sum()), dtype=np.uint8)) % 4
            parts += [e, rng.integers(0, 4, 30, dtype=np.uint8)]
        genomes.append([po.codes_to_ascii(np.concatenate(parts))])
    dbs = po.build_bitvec_dbs(genomes, k)
    tbl = engine.PanTable(ctx, k, n)
    for g in range(n):
        ss = engine.SeqSet.from_host(ctx, genomes[g])
        tbl.insert_seqset(g, ss)
        ss.close()
    keys, vals = tbl.export(0)
    o = np.argsort(keys)
    assert np.array_equal(keys[o], dbs[0][0]) and np.array_equal(vals[o], dbs[0][1])
    for dense in (False, True):
        if dense:
            tbl.rehash(4.0)
        seq = genomes[1][0]
        rows, rows100, bins, cs = tbl.anchor_contig(seq)
        o_rows, o_rows100, o_bins, _, o_cs = po.anchor_contig(dbs, seq, k, n)
        assert np.array_equal(rows, o_rows) and np.array_equal(rows100, o_rows100)
        assert np.array_equal(bins.astype(np.int64), o_bins) and np.array_equal(cs.astype(np.int64), o_cs)
        got = tbl.counters_for_read(0, seq[:5000])
        assert np.array_equal(got, po.counters_for_read(dbs[0], seq[:5000], k))
    tbl.close()


@pytest.mark.parametrize("k,period", [(21, 7), (21, 30), (31, 45), (21, 200)])
def test_tandem_repeats_equal_kmers_within_a_batch(ctx, k, period):
    """tandem repeats put EQUAL new k-mers side by side — into one 64-lane batch of the table build, whose lanes claim
    distinct slots of a line at once (the later copy of a key is retired) — and into neighbouring tiles, built by
    different waves at the same time.  The exported key set must be the oracle's, each key once; anchoring too."""
    from panagram_amd import engine
    rng = np.random.default_rng(period)
    n = 3
    genomes = []
    for g in range(n):
        parts = []
        for _ in range(12):
            unit = rng.integers(0, 4, period, dtype=np.uint8)
            parts.append(np.tile(unit, int(rng.integers(300, 3000)) // period + 2))
            parts.append(rng.integers(0, 4, int(rng.integers(50, 600)), dtype=np.uint8))
        genomes.append([po.codes_to_ascii(np.concatenate(parts)), po.codes_to_ascii(np.tile(rng.integers(0, 4, period, dtype=np.uint8), 40000 // period))])
    dbs = po.build_bitvec_dbs(genomes, k)
    tbl = engine.PanTable(ctx, k, n)
    for g in range(n):
        ss = engine.SeqSet.from_host(ctx, genomes[g])
        tbl.insert_seqset(g, ss)
        ss.close()
    keys, vals = tbl.export(0)
    o = np.argsort(keys)
    assert np.array_equal(keys[o], dbs[0][0]) and np.array_equal(vals[o], dbs[0][1])
    assert tbl.stats()["nkeys"] == len(dbs[0][0])
    for g in range(n):
        for seq in genomes[g]:
            rows, rows100, bins, cs = tbl.anchor_contig(seq)
            o_rows, o_rows100, o_bins, _, o_cs = po.anchor_contig(dbs, seq, k, n)
            assert np.array_equal(rows, o_rows) and np.array_equal(bins.astype(np.int64), o_bins)
    tbl.rehash(3.0)  # retired copies stay behind
    keys, vals = tbl.export(0)
    o = np.argsort(keys)
    assert np.array_equal(keys[o], dbs[0][0]) and np.array_equal(vals[o], dbs[0][1])
    tbl.close()


@pytest.mark.parametrize("n,k,dense", [(9, 21, False), (40, 31, False), (70, 21, False), (9, 21, True)])
def test_table_of_the_anchors_kmers_only(ctx, n, k, dense):
    """pg_table_update_seqset: a table built from the anchor genomes, the other genomes only setting bits in it, holds
    exactly the anchors' k-mers (each with the full presence mask) and anchors those genomes like the table of all
    genomes — also when it is dense enough for keys to sit in overflow lines (the update follows the chain)."""
    from panagram_amd import engine
    gen = po.synth_genomes(n, [6000, 1800, 300], 0.03, 4100 + n)
    genomes = [[po.codes_to_ascii(c) for c in g] for g in gen]
    anchors = [1, n - 1]
    dbs = po.build_bitvec_dbs(genomes, k)
    sets = [engine.SeqSet.from_host(ctx, g) for g in genomes]
    own = engine.KmerSketch(ctx, k)
    for g in anchors:
        own.add(sets[g])
    tbl = engine.PanTable(ctx, k, n, expected_keys=max(1024, own.estimate() // (6 if dense else 1)))
    own.close()
    for g in anchors:
        tbl.insert_seqset(g, sets[g])
    nk = tbl.stats()["nkeys"]
    for g in range(n):
        if g not in anchors:
            tbl.update_seqset(g, sets[g])
    assert tbl.stats()["nkeys"] == nk  # nothing was added
    akeys = np.unique(np.concatenate([po.build_bitvec_dbs([genomes[g]], k)[0][0] for g in anchors]))
    for d, (fk, fm) in enumerate(dbs):
        keys, vals = tbl.export(d)
        o = np.argsort(keys)
        sel = np.isin(fk, akeys) & (fm != 0)
        assert np.array_equal(keys[o], fk[sel]) and np.array_equal(vals[o], fm[sel])
    for g in anchors:
        for seq in genomes[g]:
            rows, rows100, bins, cs = tbl.anchor_contig(seq)
            o_rows, o_rows100, o_bins, _, o_cs = po.anchor_contig(dbs, seq, k, n)
            assert np.array_equal(rows, o_rows) and np.array_equal(bins.astype(np.int64), o_bins)
            assert np.array_equal(cs.astype(np.int64), o_cs)
    for ss in sets:
        ss.close()
    tbl.close()


@pytest.mark.parametrize("k", [21, 31])
def test_cooperative_build_equals_per_thread_build(ctx, k, monkeypatch):
    """k_insert_tile (a run's lanes claim distinct slots in one round, later copies of a key retired) against the
    one-thread-per-k-mer build (PG_INSERT_PER_THREAD) on repeat-rich genomes — identical and diverged copies of
    elements, tandem arrays of periods 2..700, at a size where many waves build at once: the same (key, mask) set, every
    key once, the same key count (tools/insert_equiv.py is the larger version)."""
    from panagram_amd import engine
    rng = np.random.default_rng(70 + k)
    G = 3
    elems = [rng.integers(0, 4, n, dtype=np.uint8) for n in (300, 1100, 5000)]
    genomes = []
    for g in range(G):
        parts = []
        for _ in range(900):
            kind = rng.integers(0, 4)
            if kind == 0:
                parts.append(elems[int(rng.integers(0, 3))])
            elif kind == 1:
                e = elems[int(rng.integers(0, 3))].copy()
                mut = rng.random(len(e)) < 0.02
                e[mut] = (e[mut] + rng.integers(1, 4, int(mut.sum()), dtype=np.uint8)) % 4
                parts.append(e)
            elif kind == 2:
                period = int(rng.choice([2, 5, 13, 40, 150, 700]))
                parts.append(np.tile(rng.integers(0, 4, period, dtype=np.uint8), int(rng.integers(200, 4000)) // period + 2))
            else:
                parts.append(rng.integers(0, 4, int(rng.integers(100, 3000)), dtype=np.uint8))
        genomes.append([po.codes_to_ascii(np.concatenate(parts))])

    def build():
        tbl = engine.PanTable(ctx, k, G)
        for g in range(G):
            ss = engine.SeqSet.from_host(ctx, genomes[g])
            tbl.insert_seqset(g, ss)
            ss.close()
        keys, vals = tbl.export(0)
        o = np.argsort(keys, kind="stable")
        nk = tbl.stats()["nkeys"]
        tbl.close()
        return keys[o], vals[o], nk

    monkeypatch.delenv("PG_INSERT_PER_THREAD", raising=False)
    ka, va, na = build()
    monkeypatch.setenv("PG_INSERT_PER_THREAD", "1")
    kb, vb, nb = build()
    assert len(np.unique(ka)) == len(ka)
    assert np.array_equal(ka, kb) and np.array_equal(va, vb)
    assert na == nb == len(ka)


@pytest.mark.parametrize("n,k,piece", [(5, 21, 3), (40, 31, 0), (9, 21, 1)])
def test_coscheduled_result_over_all_anchor_genomes(ctx, n, k, piece):
    """one result over the concatenated contigs of every anchor genome, tiles interleaved genome by
    genome (pg_result_coschedule): rows, bitmap.100, bins and per-genome column sums equal the
    genome-by-genome results and the oracle, for any grouping"""
    from panagram_amd import engine
    lens = [40000, 2600, 700, 10]
    gen = po.synth_genomes(n, lens, 0.02, 1000 + n)
    genomes = [[po.codes_to_ascii(c) for c in g] for g in gen]
    dbs = po.build_bitvec_dbs(genomes, k)
    tbl = engine.PanTable(ctx, k, n)
    sets = []
    for g in range(n):
        ss = engine.SeqSet.from_host(ctx, genomes[g])
        tbl.insert_seqset(g, ss)
        sets.append(ss)
    anchors = [0, 2, n - 1]
    per = {}
    for g in anchors:
        r = engine.AnchorResult(tbl, sets[g])
        r.run()
        per[g] = ([r.download(ci) for ci in range(len(lens))], r.colsums())
        r.close()
    merged = engine.SeqSet.concat(ctx, [sets[g] for g in anchors])
    assert [int(x) for x in merged.lens] == lens * len(anchors)
    res = engine.AnchorResult(tbl, merged)
    groups = np.repeat(np.arange(len(anchors)), len(lens))
    rng = np.random.default_rng(n)
    classes = [None, None, None, None, np.tile(np.arange(len(lens)), len(anchors)), rng.integers(0, 3, len(groups)),
               rng.permutation(len(groups))]  # homology classes: by contig number, arbitrary, all distinct
    for grp, cls in zip((groups, None, (groups + 1) % 2, np.zeros(len(groups), int), groups, groups, groups), classes):
        res.coschedule(grp, piece, contig_class=cls)
        res.run()
        for gi, g in enumerate(anchors):
            for ci in range(len(lens)):
                got = res.download(gi * len(lens) + ci)
                want = per[g][0][ci]
                for x, y in zip(got[:3], want[:3]):
                    assert np.array_equal(x, y)
            ccs = res.contig_colsums(gi * len(lens), len(lens))
            assert np.array_equal(ccs.sum(axis=0), per[g][1])
        assert np.array_equal(res.colsums(), sum(per[g][1] for g in anchors))
    o = po.anchor_contig(dbs, genomes[2][0], k, n)
    assert np.array_equal(res.download(len(lens))[0], o[0]) and np.array_equal(per[2][1].astype(np.int64), sum(
        po.anchor_contig(dbs, s, k, n)[4] for s in genomes[2] if len(s) >= k))
    res.close()
    merged.close()
    for ss in sets:
        ss.close()
    tbl.close()


def test_handles_close_children_first():
    """dropping a whole object graph (or closing a parent early) must not free a table under a result"""
    import gc
    from panagram_amd import engine
    c = engine.Context(0)
    t = engine.PanTable(c, 21, 2)
    s = engine.SeqSet.from_host(c, [b"ACGT" * 50])
    t.insert_seqset(0, s)
    r = engine.AnchorResult(t, s)
    r.run()
    t.close()              # parent first: the result goes with it
    assert r._h is None
    r2 = engine.AnchorResult(engine.PanTable(c, 21, 2), s)
    c.close()              # closes tables, sequences and their results
    assert r2._h is None and s._h is None
    for _ in range(3):     # graphs dropped to the collector in one piece
        c = engine.Context(0)
        t = engine.PanTable(c, 21, 2)
        s = engine.SeqSet.from_host(c, [b"ACGT" * 50])
        r = engine.AnchorResult(t, s)
        cyc = [c, t, s, r]
        cyc.append(cyc)
        del c, t, s, r, cyc
        gc.collect()


def test_sixteen_slot_lines_knob(ctx, monkeypatch):
    """PG_TABLE_SLOTS=16 re-hashes into 256-byte lines (tuning knob): same answers"""
    from panagram_amd import engine
    n, k = 40, 31
    gen = po.synth_genomes(n, [20000, 900], 0.03, 4040)
    genomes = [[po.codes_to_ascii(c) for c in g] for g in gen]
    dbs = po.build_bitvec_dbs(genomes, k)
    tbl = engine.PanTable(ctx, k, n)
    for g in range(n):
        ss = engine.SeqSet.from_host(ctx, genomes[g])
        tbl.insert_seqset(g, ss)
        ss.close()
    for slots in ("16", "8"):
        monkeypatch.setenv("PG_TABLE_SLOTS", slots)
        tbl.rehash(3.0)
        frac, got_slots = tbl.spill()
        assert got_slots == int(slots) and 0.0 <= frac < 1.0
        for seq in genomes[7]:
            rows, rows100, bins, cs = tbl.anchor_contig(seq)
            o_rows, o_rows100, o_bins, _, o_cs = po.anchor_contig(dbs, seq, k, n)
            assert np.array_equal(rows, o_rows) and np.array_equal(rows100, o_rows100)
            assert np.array_equal(bins.astype(np.int64), o_bins) and np.array_equal(cs.astype(np.int64), o_cs)
    tbl.close()


@pytest.mark.parametrize("k", [21, 31, 9])
def test_kmer_sketch_registers_equal_oracle(ctx, k):
    """Table sizing: the GPU sketch's registers are bit-exact with the restatement, the estimate is
    within 5 standard errors of the exact distinct count, and adding an input twice changes nothing."""
    from panagram_amd import engine
    rng = np.random.default_rng(100 + k)
    base = rng.integers(0, 4, 400_000, dtype=np.uint8)
    other = base.copy()
    hit = rng.random(len(other)) < 0.02
    other[hit] = (other[hit] + rng.integers(1, 4, int(hit.sum()), dtype=np.uint8)) & 3
    a = bytearray(po.codes_to_ascii(base))
    a[1000:1300] = b"N" * 300
    a[50_000:50_010] = b"acgtnacgtn"
    seqs = [[bytes(a[:250_000]), bytes(a[250_000:])], [po.codes_to_ascii(other), b"ACGT"]]
    sk = engine.KmerSketch(ctx, k)
    for contigs in seqs:
        ss = engine.SeqSet.from_host(ctx, contigs)
        sk.add(ss)
        sk.add(ss)
        ss.close()
    want = po.sketch_registers([c for g in seqs for c in g], k)
    assert np.array_equal(sk.registers(), want)
    exact = len(np.unique(np.concatenate([v[ok] for v, ok in (po.canonical_kmers(c, k) for g in seqs for c in g)])))
    est = sk.estimate()
    assert est == po.sketch_estimate(want)
    assert abs(est - exact) <= 0.02 * exact
    # a table created from the estimate takes all keys without growing
    tbl = engine.PanTable(ctx, k, 2, expected_keys=est + est // 32 + 1024)
    before = tbl.stats()["nbuckets"]
    for g, contigs in enumerate(seqs):
        ss = engine.SeqSet.from_host(ctx, contigs)
        tbl.insert_seqset(g, ss)
        ss.close()
    st = tbl.stats()
    assert st["nkeys"] == exact and st["nbuckets"] == before
    sk.close()
    tbl.close()


@pytest.mark.parametrize("n,k,lens", [(1, 21, [700_123]), (5, 21, [520_000, 9_000]), (8, 31, [900_077, 450_321]),
                                      (8, 21, [50_020, 12_345, 7_000, 33_333]), (3, 21, [204_900, 6_700]),
                                      (2, 21, [3_400_000, 1_700_000, 250_000])])  # (bins of 34 k and 17 k rows: STREAKS of whole one-bin groups, k_epilogue)
def test_one_byte_rows_long_contigs_against_oracle(ctx, n, k, lens):
    """contigs long enough for the bit-sliced statistics path of one-byte rows (32 rows per thread over 8 whole tiles)
    in both of its kinds — all 4096 rows inside one bin; several bins of 66..2049 rows each, a thread's rows split at
    the boundary — next to the per-tile path (contig ends, bins under 66 rows): bins of nkmers / 100 = 67 rows .. 12.5 k;
    N runs and lower case included.  Rows, bitmap.100, bins and per-contig column sums against the oracle."""
    from panagram_amd import engine
    rng = np.random.default_rng(n * 1000 + k)
    gen = po.synth_genomes(n, lens, 0.01, 31 + n)
    genomes = [[bytearray(po.codes_to_ascii(c)) for c in g] for g in gen]
    for g in range(n):
        for c in genomes[g]:
            for _ in range(3):
                p, run = int(rng.integers(0, len(c) - 5000)), int(rng.integers(1, 3000))
                c[p:p + run] = b"N" * run
            q = int(rng.integers(0, len(c) - 500))
            c[q:q + 400] = bytes(c[q:q + 400]).lower()
    genomes = [[bytes(c) for c in g] for g in genomes]
    dbs = po.build_bitvec_dbs(genomes, k)
    tbl = engine.PanTable(ctx, k, n)
    for g in range(n):
        ss = engine.SeqSet.from_host(ctx, genomes[g])
        tbl.insert_seqset(g, ss)
        ss.close()
    for g in {0, n - 1}:
        ss = engine.SeqSet.from_host(ctx, genomes[g])
        res = engine.AnchorResult(tbl, ss, colsums=True)
        res.run()
        ccs = res.contig_colsums().astype(np.int64)
        for ci, seq in enumerate(genomes[g]):
            rows, rows100, bins, info = res.download(ci)
            o_rows, o_rows100, o_bins, _, o_cs = po.anchor_contig(dbs, seq, k, n)
            assert np.array_equal(rows, o_rows)
            assert np.array_equal(rows100, o_rows100)
            assert np.array_equal(bins.astype(np.int64), o_bins), np.argwhere(bins.astype(np.int64) != o_bins)[:5]
            assert np.array_equal(ccs[ci], o_cs)
        res.close()
        ss.close()
    tbl.close()


@pytest.mark.parametrize("n,k,lens", [(65, 21, [230_000, 9_000]), (128, 31, [150_000, 2_000]), (200, 21, [120_000]),
                                      (257, 21, [110_000, 5_000])])
def test_wide_rows_long_contigs_against_oracle(ctx, n, k, lens):
    """more than 64 genomes on contigs whose bins are longer than a tile (nkmers / 100 = 1099 .. 2299 rows): the
    chunk-parallel statistics pass then keeps the histogram of a tile's one or two bins in several LDS copies, folds them
    when the window moves on (every second or third tile here) and goes back to the plain window on the short contig
    behind.  One, two and three 16-byte chunks per row, exact and ragged.  Rows, bitmap.100, bins and per-contig column
    sums against the oracle."""
    from panagram_amd import engine
    rng = np.random.default_rng(n * 1000 + k)
    gen = po.synth_genomes(n, lens, 0.01, 57 + n)
    genomes = [[bytearray(po.codes_to_ascii(c)) for c in g] for g in gen]
    for g in (0, n // 2, n - 1):
        for c in genomes[g]:
            p, run = int(rng.integers(0, len(c) - 1500)), int(rng.integers(1, 1200))
            c[p:p + run] = b"N" * run
            q = int(rng.integers(0, len(c) - 500))
            c[q:q + 400] = bytes(c[q:q + 400]).lower()
    genomes = [[bytes(c) for c in g] for g in genomes]
    dbs = po.build_bitvec_dbs(genomes, k)
    tbl = engine.PanTable(ctx, k, n)
    for g in range(n):
        ss = engine.SeqSet.from_host(ctx, genomes[g])
        tbl.insert_seqset(g, ss)
        ss.close()
    for g in {0, n - 1}:
        ss = engine.SeqSet.from_host(ctx, genomes[g])
        res = engine.AnchorResult(tbl, ss, colsums=True)
        res.run()
        ccs = res.contig_colsums().astype(np.int64)
        for ci, seq in enumerate(genomes[g]):
            rows, rows100, bins, info = res.download(ci)
            o_rows, o_rows100, o_bins, _, o_cs = po.anchor_contig(dbs, seq, k, n)
            assert np.array_equal(rows, o_rows)
            assert np.array_equal(rows100, o_rows100)
            assert np.array_equal(bins.astype(np.int64), o_bins), np.argwhere(bins.astype(np.int64) != o_bins)[:5]
            assert np.array_equal(ccs[ci], o_cs)
        res.close()
        ss.close()
    tbl.close()


@pytest.mark.parametrize("n", [66, 73, 80, 81, 88, 90, 99, 105, 112, 113, 120, 126])
def test_rows_of_9_to_16_bytes_against_oracle(ctx, n):
    """65..128 genomes, every row width from 9 to 16 bytes (three and four words per row, ragged and whole): the statistics
    pass k_epilogue_w reads 16 consecutive rows per thread as 4 x nbytes aligned words and cuts the rows out with static
    funnel shifts.  Contigs whose groups of four tiles lie inside one or two long bins (the histogram in 8 LDS copies),
    across several bins of a few hundred rows (the plain window), a contig of three tiles (no group: one tile at a time,
    bins shorter than the window's minimum) and one whose last tile is partial.  Rows, bitmap.100, bins and per-contig column
    sums against the oracle (cpp/anchor.cpp:150-189, index.py:1051,1169-1183)."""
    from panagram_amd import engine
    k = 21 if n % 2 else 31
    lens = [262_000 + 17 * n, 41_000 + n, 3_000 + n]
    rng = np.random.default_rng(n * 77 + k)
    gen = po.synth_genomes(n, lens, 0.01, 911 + n)
    genomes = [[bytearray(po.codes_to_ascii(c)) for c in g] for g in gen]
    for g in (0, n - 1):
        for c in genomes[g]:
            p, run = int(rng.integers(0, len(c) - 1500)), int(rng.integers(1, 700))
            c[p:p + run] = b"N" * run
    genomes = [[bytes(c) for c in g] for g in genomes]
    dbs = po.build_bitvec_dbs(genomes, k)
    tbl = engine.PanTable(ctx, k, n)
    for g in range(n):
        ss = engine.SeqSet.from_host(ctx, genomes[g])
        tbl.insert_seqset(g, ss)
        ss.close()
    g = 0 if n % 3 else n - 1
    ss = engine.SeqSet.from_host(ctx, genomes[g])
    res = engine.AnchorResult(tbl, ss, colsums=True)
    res.run()
    ccs = res.contig_colsums().astype(np.int64)
    for ci, seq in enumerate(genomes[g]):
        rows, rows100, bins, info = res.download(ci)
        o_rows, o_rows100, o_bins, _, o_cs = po.anchor_contig(dbs, seq, k, n)
        assert np.array_equal(rows, o_rows)
        assert np.array_equal(rows100, o_rows100)
        assert np.array_equal(bins.astype(np.int64), o_bins), np.argwhere(bins.astype(np.int64) != o_bins)[:5]
        assert np.array_equal(ccs[ci], o_cs)
    res.close()
    ss.close()
    tbl.close()


@pytest.mark.parametrize("n", [20, 36, 44, 52, 68, 76, 84, 100, 108, 116])  # rows of 3 / 5, 6, 7 / 9, 10, 11 / 13, 14, 15 bytes
def test_ragged_rows_keep_their_neighbours_first_bytes(ctx, n):
    """k_probe writes a ragged row (3, 5..7, 9..11, 13..15 bytes) as ONE store of the next word size: the row plus the first
    bytes of the NEXT position's row, which the next lane holds; the last lane of a batch has no successor in its batch and
    writes ZEROS there, which the next batch's own store — a later instruction of the same wave — puts right (DESIGN.md
    section 3, "ragged rows"; rows resolved by the overflow drain are stored exactly).  Directed at that invariant: every
    genome a near copy of the ancestor, so that nearly EVERY row has all of its first bytes set (a zero left behind at a
    batch boundary cannot hide), at every ragged width; half of the contig is one diverged repeat family, whose keys
    overflow their home lines (queue entries in every batch, queue-full drains in the middle of tiles, rows rewritten by
    the drain next to rows of the main batches).  Rows against the oracle (cpp/anchor.cpp:139-164)."""
    from panagram_amd import engine
    k = 21 if n % 8 else 31
    rng = np.random.default_rng(5150 + n)
    elem = rng.integers(0, 4, 500, dtype=np.uint8)
    parts = [rng.integers(0, 4, 6000, dtype=np.uint8)]
    for c in range(12):
        e = elem.copy()
        mut = rng.random(len(e)) < 0.05
        e[mut] = (e[mut] + rng.integers(1, 4, int(mut.sum()), dtype=np.uint8)) % 4
        parts += [e, rng.integers(0, 4, 25, dtype=np.uint8)]
    anc = [np.concatenate(parts), rng.integers(0, 4, 2500 + n, dtype=np.uint8)]
    genomes = []
    for g in range(n):
        cs = []
        for a in anc:
            c = a.copy()
            if g:
                mut = rng.random(len(c)) < 0.0015
                c[mut] = (c[mut] + rng.integers(1, 4, int(mut.sum()), dtype=np.uint8)) % 4
            cs.append(po.codes_to_ascii(c))
        genomes.append(cs)
    dbs = po.build_bitvec_dbs(genomes, k)
    tbl = engine.PanTable(ctx, k, n)
    for g in range(n):
        ss = engine.SeqSet.from_host(ctx, genomes[g])
        tbl.insert_seqset(g, ss)
        ss.close()
    for dense in (False, True):
        if dense:
            tbl.rehash(6.0)  # (denser lines: more keys outside their home line, more drain work)
        for g in (0, n - 1):
            ss = engine.SeqSet.from_host(ctx, genomes[g])
            res = engine.AnchorResult(tbl, ss, colsums=True)
            res.run()
            for ci, seq in enumerate(genomes[g]):
                rows = res.download(ci)[0]
                o_rows = po.anchor_contig(dbs, seq, k, n)[0]
                full = (o_rows[:, 0] == 0xFF).mean()
                assert full > 0.5, full  # (the directed part: most rows do start with a set byte)
                assert np.array_equal(rows, o_rows), (n, dense, g, ci, np.argwhere(rows != o_rows)[:4])
            res.close()
            ss.close()
    tbl.close()


@pytest.mark.parametrize("n,k,kpl", [(2, 21, 5.5), (8, 31, 6.4), (40, 21, 4.0), (1, 15, 5.0), (8, 21, 1.5), (27, 31, 2.0), (70, 21, 1.5), (100, 31, 1.5)])
def test_dense_tables_hold_the_same_sets_and_answer_the_same(ctx, n, k, kpl):
    """pg_table_create_dense (round 6): a table created at ``kpl`` keys per 128-byte line instead of 3 — denser (the genome-sharded
    mode's block tables: more keys outside their home lines, longer probe sequences, the overflow queue at work in every tile) or
    sparser (what Index.build_table asks for where HBM is plentiful) — holds exactly the oracle's k-mer sets,
    answers GetCountersForRead and the anchor step bit for bit, and is not grown back while it fills; fed more keys than it was
    created for it grows like any other table.  (KMC's sorted arrays have one density: cpp/anchor.cpp:28-31 is the call site.)"""
    from panagram_amd import engine
    gen = po.synth_genomes(n, [60_000, 9_000], 0.05, 1234 + n)
    genomes = [[po.codes_to_ascii(c) for c in g] for g in gen]
    dbs = po.build_bitvec_dbs(genomes, k)
    nkeys = len(np.unique(np.concatenate([d[0] for d in dbs])))  # (ONE table of all genomes: the union of the 32-genome groups' sets)
    tbl = engine.PanTable(ctx, k, n, expected_keys=int(nkeys * 1.03) + 1024, keys_per_line=kpl)
    bytes0 = tbl.stats()["bytes"]
    per_line = kpl if n <= 64 else kpl / 8.0 * (6 if n <= 96 else 16)        # (inline lines hold 6 keys, split lines 16: the same load)
    line_bytes = 128.0 if n <= 96 else 16 * (8 + 4 * ((n + 31) // 32))        # (split layout: 16 bare keys + their mask words in a second array)
    assert abs(bytes0 / (line_bytes * (int(nkeys * 1.03) + 1024) / per_line) - 1) < 0.02
    big = 3_000_000_000  # (the planner's arithmetic, host only: pg_table_bytes_for_dense)
    assert abs(engine.PanTable.bytes_for(k, n, big, keys_per_line=kpl) / (line_bytes * big / per_line) - 1) < 0.01
    for g in range(n):
        ss = engine.SeqSet.from_host(ctx, genomes[g])
        tbl.insert_seqset(g, ss)
        ss.close()
    st = tbl.stats()
    assert st["bytes"] == bytes0, "a dense table must not be grown back to the library's density while it fills"
    assert st["nkeys"] == nkeys and 0.9 * per_line / 1.05 < st["nkeys"] / st["nbuckets"] < per_line
    keys, vals = tbl.export(0)
    o = np.argsort(keys)
    assert np.array_equal(keys[o], dbs[0][0]) and np.array_equal(vals[o], dbs[0][1])
    for g in (0, n - 1):
        for seq in genomes[g]:
            rows, rows100, bins, cs = tbl.anchor_contig(seq)
            o_rows, o_rows100, o_bins, _, o_cs = po.anchor_contig(dbs, seq, k, n)
            assert np.array_equal(rows, o_rows) and np.array_equal(rows100, o_rows100)
            assert np.array_equal(bins.astype(np.int64), o_bins) and np.array_equal(cs.astype(np.int64), o_cs)
    assert np.array_equal(tbl.counters_for_read(0, genomes[0][0][:4000]), po.counters_for_read(dbs[0], genomes[0][0][:4000], k))
    if kpl < 3:  # (a sparse table has room for several times its keys: nothing to grow for here)
        tbl.close()
        return
    # far more keys than it was created for: it grows (and still answers)
    extra = po.codes_to_ascii(np.random.default_rng(5).integers(0, 4, 2_500_000, dtype=np.uint8))
    ss = engine.SeqSet.from_host(ctx, [extra])
    tbl.insert_seqset(0, ss)
    ss.close()
    assert tbl.stats()["bytes"] > bytes0
    rows = tbl.anchor_contig(genomes[n - 1][1])[0]
    dbs2 = po.build_bitvec_dbs([genomes[0] + [extra]] + genomes[1:], k)
    assert np.array_equal(rows, po.anchor_contig(dbs2, genomes[n - 1][1], k, n)[0])
    with pytest.raises(Exception):
        engine.PanTable(ctx, k, n, expected_keys=1000, keys_per_line=7.5)
    tbl.close()


def _bins_of(rows, n, binlen):
    """popcount histogram per bin of ``binlen`` rows (cpp/anchor.cpp:179-189) from oracle rows"""
    pc = np.unpackbits(rows, axis=1, bitorder="little")[:, :n].sum(axis=1)
    nb = (len(rows) + binlen - 1) // binlen
    out = np.zeros((nb, n + 1), np.int64)
    for b in range(nb):
        out[b] = np.bincount(pc[b * binlen:(b + 1) * binlen], minlength=n + 1)
    return out


@pytest.mark.parametrize("n", [9, 12, 20, 27, 32, 40, 47, 52, 64, 65, 73, 80, 90, 96, 100, 108, 116, 128])
def test_fused_statistics_equal_the_pass_and_the_oracle(ctx, n, monkeypatch):
    """Round 6 (opt-in, PG_FUSE_STATS=1): rows of 2..8 bytes (..16 in a -DPG_FUSE_WIDE=1 build) end every tile INSIDE k_probe with
    the tile's popcount histogram, column sums and 1-in-100 rows (read back from the rows just written; k_tile_reduce adds the
    tiles' counters up) for contigs whose bins are at least a tile (1024 rows) long; contigs with shorter bins stay with the statistics pass, launched over their tiles
    only.  Bins of 1024 / 1500 / 2500 rows here (AnchorResult(max_bin_len=..., min_bin_count=1)): tiles inside one bin, tiles
    across two bins, a contig that is one short bin (not fused: mixed launch), partial last tiles, N runs; a lowres step other
    than 100 (k_lowres) once.  bitmap.100, bins and per-contig column sums against the oracle (cpp/anchor.cpp:156-189,
    index.py:1051) AND against the unfused pass (PG_FUSE_STATS=0) in the same process; the result says which path ran."""
    from panagram_amd import engine
    k = 21 if n % 2 else 31
    lens = [7000 + 13 * n, 2600, 1100 + k, 700, 4096 + k - 1]
    rng = np.random.default_rng(31 * n + k)
    gen = po.synth_genomes(n, lens, 0.02, 1700 + n)
    genomes = [[bytearray(po.codes_to_ascii(c)) for c in g] for g in gen]
    for g in (0, n - 1):
        c = genomes[g][0]
        p_ = int(rng.integers(100, len(c) - 400))
        c[p_:p_ + 37] = b"N" * 37
    genomes = [[bytes(c) for c in g] for g in genomes]
    dbs = po.build_bitvec_dbs(genomes, k)
    tbl = engine.PanTable(ctx, k, n)
    for g in range(n):
        ss = engine.SeqSet.from_host(ctx, genomes[g])
        tbl.insert_seqset(g, ss)
        ss.close()
    for max_bin, step in ((1024, 100), (1500, 100), (2500, 7)):
        ga = n - 1 if max_bin == 1500 else 0
        ss = engine.SeqSet.from_host(ctx, genomes[ga])
        got = {}
        for fuse in ("1", "0"):
            monkeypatch.setenv("PG_FUSE_STATS", fuse)
            res = engine.AnchorResult(tbl, ss, colsums=True, max_bin_len=max_bin, min_bin_count=1, lowres_step=step)
            res.run()
            res.run()  # (a second run over the same buffers: the tiles' counters are overwritten, not added to)
            # (rows of 9..16 bytes have fused instantiations only in a -DPG_FUSE_WIDE=1 build: csrc/pg_kernels.h)
            want = (0,) if fuse == "0" else (2,) if (n + 7) // 8 <= 8 else (0, 2)
            assert res.fused_runs() in want, (n, max_bin, fuse, res.fused_runs())
            ccs = res.contig_colsums().astype(np.int64)
            got[fuse] = [res.download(ci) + (ccs[ci],) for ci in range(len(lens))]
            res.close()
        for ci, seq in enumerate(genomes[ga]):
            o_rows, _, _, _, o_cs = po.anchor_contig(dbs, seq, k, n)
            nk = len(o_rows)
            binlen = max_bin if nk // max_bin >= 1 else nk
            for fuse in ("1", "0"):
                rows, rows_lo, bins, info, cs = got[fuse][ci]
                assert info["binlen"] == binlen and info["nkmers"] == nk
                assert np.array_equal(rows, o_rows), (n, max_bin, fuse, ci)
                assert np.array_equal(rows_lo, o_rows[::step]), (n, max_bin, fuse, ci)
                assert np.array_equal(bins.astype(np.int64), _bins_of(o_rows, n, binlen)), (n, max_bin, fuse, ci)
                assert np.array_equal(cs, o_cs), (n, max_bin, fuse, ci)
        ss.close()
    tbl.close()


def test_fused_statistics_default_bins_long_contigs_coscheduled(ctx, monkeypatch):
    """The same with the reference's own bin rule (200 000 rows, or a hundredth of the contig: cpp/anchor.cpp:114-118) on contigs
    long enough for it to give bins of more than a tile — 150 000 and 260 000 k-mers: bins of 1500 and 2600 rows, most tiles
    across two bins — next to a contig of 30 000 (bins of 300 rows: the pass), two anchor genomes co-scheduled in one launch
    (bench.py's mode), for rows of 4, 8, 9 and 16 bytes."""
    from panagram_amd import engine
    for n, k in ((27, 21), (64, 31), (65, 21), (128, 21)):
        lens = [150_000 + k - 1, 30_000, 260_000 + k - 1]
        gen = po.synth_genomes(n, lens, 0.01, 4400 + n)
        genomes = [[po.codes_to_ascii(c) for c in g] for g in gen]
        dbs = po.build_bitvec_dbs(genomes, k)
        tbl = engine.PanTable(ctx, k, n)
        for g in range(n):
            ss = engine.SeqSet.from_host(ctx, genomes[g])
            tbl.insert_seqset(g, ss)
            ss.close()
        picks = [1, n - 1]
        sss = [engine.SeqSet.from_host(ctx, genomes[g]) for g in picks]
        both = engine.SeqSet.concat(ctx, sss)
        monkeypatch.setenv("PG_FUSE_STATS", "1")
        res = engine.AnchorResult(tbl, both, colsums=True)
        res.coschedule(np.repeat(np.arange(2), len(lens)), 2)
        res.run()
        assert res.fused_runs() in ((1,) if (n + 7) // 8 <= 8 else (0, 1))
        ccs = res.contig_colsums().astype(np.int64)
        for gi, g in enumerate(picks):
            for ci, seq in enumerate(genomes[g]):
                rows, rows100, bins, info = res.download(gi * len(lens) + ci)
                o_rows, o_rows100, o_bins, _, o_cs = po.anchor_contig(dbs, seq, k, n)
                assert np.array_equal(rows, o_rows) and np.array_equal(rows100, o_rows100), (n, g, ci)
                assert np.array_equal(bins.astype(np.int64), o_bins), (n, g, ci, np.argwhere(bins.astype(np.int64) != o_bins)[:5])
                assert np.array_equal(ccs[gi * len(lens) + ci], o_cs), (n, g, ci)
        for x in (res, both, *sss, tbl):
            x.close()


_FUZZ_N = [2, 8, 9, 16, 17, 24, 25, 32, 33, 40, 63, 64, 65, 72, 96, 97, 127, 128, 129, 160, 193, 256, 257, 300]


@pytest.mark.parametrize("seed", range(int(os.environ.get("PG_FUZZ_SEEDS", "64"))))  # (more seeds: PG_FUZZ_SEEDS=600)
def test_random_shapes_against_oracle(ctx, seed):
    """Randomised sweep over genome counts around every row-width boundary (1..38-byte rows, 1..5
    sub-tables), k, and contig lengths around the tile / bin / 1-in-100 boundaries, N runs included:
    rows, bitmap.100, bins and column sums of a multi-contig launch against the oracle."""
    from panagram_amd import engine
    rng = np.random.default_rng(7000 + seed)
    n = int(_FUZZ_N[(2 * seed + int(rng.integers(0, 2))) % len(_FUZZ_N)])
    k = int(rng.choice([15, 21, 31, 32]))
    # (contigs of fewer than 100 k-mers divide by zero in the reference, cpp/anchor.cpp:114-118: not pinned)
    special = [k + 99, k + 100, k + 198, k + 199, k + 298, k + 510, k + 511, k + 512, k + 1023, k + 1500, k + 3700, k + 5199, k + 9999]
    lens = [int(x) for x in rng.choice(special, size=int(rng.integers(3, 8)))] + [int(rng.integers(k + 99, 30000))]
    gen = po.synth_genomes(n, lens, float(rng.choice([0.003, 0.02, 0.1])), 100 + seed)
    genomes = [[bytearray(po.codes_to_ascii(c)) for c in g] for g in gen]
    for g in range(0, n, max(1, n // 5)):  # N runs, lower case
        for c in genomes[g]:
            if len(c) > 200 and rng.random() < 0.5:
                p = int(rng.integers(0, len(c) - 60))
                c[p:p + int(rng.integers(1, 60))] = b"N" * 1
                q = int(rng.integers(0, len(c) - 50))
                c[q:q + 40] = bytes(c[q:q + 40]).lower()
    genomes = [[bytes(c) for c in g] for g in genomes]
    dbs = po.build_bitvec_dbs(genomes, k)
    tbl = engine.PanTable(ctx, k, n)
    if k >= 20 and seed % 2 == 1:  # (the library's own choice on short genomes is the widest window: pin the others too)
        tbl.set_minimizer(k - int(rng.integers(3, 9)) + 1)
    for g in range(n):
        ss = engine.SeqSet.from_host(ctx, genomes[g])
        tbl.insert_seqset(g, ss)
        ss.close()
    if seed % 3 == 0:
        tbl.rehash(3.0)
    for g in {0, int(rng.integers(0, n)), n - 1}:
        ss = engine.SeqSet.from_host(ctx, genomes[g])
        res = engine.AnchorResult(tbl, ss, colsums=True)
        res.run()
        ccs = res.contig_colsums().astype(np.int64)
        for ci, seq in enumerate(genomes[g]):
            rows, rows100, bins, info = res.download(ci)
            o_rows, o_rows100, o_bins, _, o_cs = po.anchor_contig(dbs, seq, k, n)
            assert np.array_equal(rows, o_rows), (n, k, len(seq))
            assert np.array_equal(rows100, o_rows100), (n, k, len(seq))
            assert np.array_equal(bins.astype(np.int64), o_bins), (n, k, len(seq))
            assert np.array_equal(ccs[ci], o_cs), (n, k, len(seq))
        res.close()
        ss.close()
    tbl.close()


@pytest.mark.parametrize("name", H.kmc2_cases())
def test_kmc2_layout_databases_load_on_the_gpu(ctx, name, tmp_path):
    """KMC2-layout databases (kmc_version 0x200: one prefix LUT per signature bin, what `kmc` itself writes) —
    file images the reference binary's own KMC reader accepted — through k_import_kmc: the table then anchors to the
    reference's golden outputs, and exports exactly the databases' k-mers.  Also through memory-mapped files
    (PanTable.load_kmc_files), the way Index.build_table and the run_anchor CLI open them."""
    from panagram_amd import engine
    fx = H.load_case(name)
    ref = H.load_case(str(fx["ref_case"]))
    n, k = int(ref["ngenomes"]), int(ref["k"])
    for via_files in (False, True):
        tbl = engine.PanTable(ctx, k, n)
        for i in range((n + 31) // 32):
            if via_files:
                p = str(tmp_path / f"bitvec{i}")
                fx[f"db{i}_pre"].tofile(p + ".kmc_pre")
                fx[f"db{i}_suf"].tofile(p + ".kmc_suf")
                tbl.load_kmc_files(i, p)
            else:
                tbl.load_kmc(i, fx[f"db{i}_pre"].tobytes(), fx[f"db{i}_suf"].tobytes())
        for i, (keys, masks) in enumerate(H.case_dbs(ref)):
            gk, gm = tbl.export(i)
            order = np.argsort(gk)
            assert np.array_equal(gk[order], keys) and np.array_equal(gm[order], masks)
        for g in ref["anchors"]:
            got = _anchor_fasta_gpu(ctx, tbl, ref[f"fasta_{g}"].tobytes())
            assert got["bitmap1"] == ref[f"a{g}_bitmap1"].tobytes()
            assert got["bitmap100"] == ref[f"a{g}_bitmap100"].tobytes()
        tbl.close()


def test_kmc_import_streams_in_chunks(ctx, tmp_path):
    """A database larger than one upload chunk (several import launches, double-buffered) and with a one-byte
    counter: every key arrives, none twice."""
    from panagram_amd import engine
    rng = np.random.default_rng(3)
    k, n = 27, 5
    nkeys = 45_000_000  # x (5 suffix bytes + 1 counter byte) = 270 MB > the 256 MiB chunk
    keys = np.unique(rng.integers(0, 1 << 54, nkeys, dtype=np.uint64))
    masks = rng.integers(1, 32, len(keys), dtype=np.uint32)
    p = str(tmp_path / "big")
    po.write_kmc1(p, keys, masks, k, lut_prefix_len=7, counter_size=1)
    tbl = engine.PanTable(ctx, k, n, expected_keys=len(keys))
    tbl.load_kmc_files(0, p)
    assert tbl.stats()["nkeys"] == len(keys)
    gk, gm = tbl.export(0)
    order = np.argsort(gk)
    assert np.array_equal(gk[order], keys) and np.array_equal(gm[order], masks)
    tbl.close()


def test_corrupt_kmc_total_is_a_format_error_not_a_crash(ctx):
    """A total_kmers whose product with the record size wraps 64 bits (a corrupt .kmc_pre) must come back as
    PG_E_FORMAT before anything is allocated or read — not pass the truncation check and read past the mapping."""
    from panagram_amd import engine
    fx = H.load_case(H.kmc2_cases()[0])
    ref = H.load_case(str(fx["ref_case"]))
    n, k = int(ref["ngenomes"]), int(ref["k"])
    pre = bytearray(fx["db0_pre"].tobytes())
    suf = fx["db0_suf"].tobytes()
    hoff = int.from_bytes(pre[-8:-4], "little")
    h = len(pre) - 8 - hoff
    sb = (k - int.from_bytes(pre[h + 12:h + 16], "little")) // 4
    rec = sb + int.from_bytes(pre[h + 8:h + 12], "little")
    total_at = h + 16 + 4 + 8  # KMC2: signature_len, min_count, max_count, then total_kmers
    real = int.from_bytes(pre[total_at:total_at + 8], "little")
    assert 8 + real * rec <= len(suf)  # (the field really is where this test patches it)
    wrapped = ((1 << 64) + len(suf) - 8) // rec - 1  # 8 + wrapped * rec wraps to a value below len(suf)
    assert (8 + wrapped * rec) % (1 << 64) <= len(suf) and wrapped > real
    pre[total_at:total_at + 8] = wrapped.to_bytes(8, "little")
    tbl = engine.PanTable(ctx, k, n)
    with pytest.raises(engine.PanagramHipError) as ei:
        tbl.load_kmc(0, bytes(pre), suf)
    assert "truncated" in str(ei.value)
    tbl.close()


@pytest.mark.parametrize("n,k,chunks", [(7, 21, 3), (27, 21, 5), (64, 31, 4), (70, 21, 6), (100, 21, 4)])
def test_a_run_in_chunks_equals_the_run_in_one_launch(ctx, n, k, chunks, monkeypatch):
    """pg_anchor_run as several probe launches (slices of the co-schedule) with the statistics pass of every slice on
    the side stream over the tile ranges the slice touches (PG_RUN_CHUNKS; the default for huge 8-byte-row runs): rows,
    bitmap.100, bins and column sums are those of the single launch + single pass — every row width class, schedules
    by group and by class, and in plain tile order."""
    from panagram_amd import engine
    lens = [30000, 9000, 2600, 700]
    gen = po.synth_genomes(n, lens, 0.02, 2000 + n)
    genomes = [[po.codes_to_ascii(c) for c in g] for g in gen]
    tbl = engine.PanTable(ctx, k, n)
    sets = []
    for g in range(n):
        ss = engine.SeqSet.from_host(ctx, genomes[g])
        tbl.insert_seqset(g, ss)
        sets.append(ss)
    anchors = [0, 1, n - 1]
    merged = engine.SeqSet.concat(ctx, [sets[g] for g in anchors])
    groups = np.repeat(np.arange(len(anchors)), len(lens))
    ncontigs = len(groups)

    def outputs(schedule):
        res = engine.AnchorResult(tbl, merged, colsums=True)
        if schedule == "groups":
            res.coschedule(groups, 2)
        elif schedule == "classes":
            res.coschedule(groups, 3, contig_class=np.tile(np.arange(len(lens))[::-1], len(anchors)))
        res.run()
        res.run()  # (a second run over the same result: the side stream's previous pass is joined first)
        out = [res.download(ci)[:3] for ci in range(ncontigs)], res.contig_colsums().copy()
        res.close()
        return out

    monkeypatch.setenv("PG_RUN_CHUNKS", "1")
    want = {s_: outputs(s_) for s_ in ("groups", "classes", "order")}
    monkeypatch.setenv("PG_RUN_CHUNKS", str(chunks))
    monkeypatch.setenv("PG_CHUNK_MIN_TILES", "4")
    for s_ in ("groups", "classes", "order"):
        got = outputs(s_)
        for ci in range(ncontigs):
            for x, y in zip(got[0][ci], want[s_][0][ci]):
                assert np.array_equal(x, y), (s_, ci)
        assert np.array_equal(got[1], want[s_][1]), s_
    o = po.anchor_contig(po.build_bitvec_dbs(genomes, k), genomes[1][0], k, n)
    assert np.array_equal(want["groups"][0][len(lens)][0], o[0])
    merged.close()
    for ss in sets:
        ss.close()
    tbl.close()
