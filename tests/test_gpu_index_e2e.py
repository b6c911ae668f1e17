"""GPU: the `panagram index` counterpart end to end — files on disk equal what the reference
binary wrote (decompressed payloads, TSV texts), and read back through the reference's
addressing rule."""
import gzip
import os

import numpy as np
import pandas as pd
import pytest

from oracle import pyoracle as po
from tests import helpers as H

pytestmark = pytest.mark.gpu


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


@pytest.mark.parametrize("batch_bytes", [None, 1])
@pytest.mark.parametrize("name", ["n2_k21", "n9_k21", "n40_k31", "n65_k21"])
def test_index_run_writes_reference_identical_tree(name, batch_bytes, tmp_path):
    """batch_bytes None: all anchors of the case in one co-scheduled launch; 1: one batch per anchor
    (the batches pipeline: anchoring of batch b+1 beside the file writers of batch b)"""
    from panagram_amd import index as pidx
    fx = H.load_case(name)
    n, k = int(fx["ngenomes"]), int(fx["k"])
    anchors = [f"g{g}" for g in fx["anchors"]]
    s = _write_case(tmp_path, fx)
    out = tmp_path / "idx"
    idx = pidx.Index(str(s), prefix=str(out), k=k, anchor_genomes=anchors, export_kmc=True)
    if batch_bytes is not None:
        idx.batch_bytes = batch_bytes
    idx.run()
    dbs = H.case_dbs(fx)
    for g in fx["anchors"]:
        adir = out / "anchor" / f"g{g}"
        for step in (1, 100):
            payload = gzip.open(adir / f"bitmap.{step}.gz", "rb").read()
            assert payload == fx[f"a{g}_bitmap{step}"].tobytes()
            gzi = np.fromfile(adir / f"bitmap.{step}.gzi", "<u8")
            ref = np.frombuffer(fx[f"a{g}_gzi{step}"].tobytes(), "<u8")
            assert gzi[0] == ref[0]
        assert (adir / "bitsum.bins.tsv").read_bytes() == fx[f"a{g}_bitsum.bins.tsv"].tobytes()
        assert (adir / "chrs.tsv").read_bytes() == fx[f"a{g}_chrs.tsv"].tobytes()
        # total_paircounts.csv (python path only, index.py:1068-1074): derived from the golden payload
        ora = po.anchor_fasta(dbs, fx[f"fasta_{g}"].tobytes(), k, n)
        tp = pd.read_csv(adir / "total_paircounts.csv", index_col="name", float_precision="round_trip")
        assert list(tp.index) == [f"g{i}" for i in range(n)]
        assert np.array_equal(tp["count"].to_numpy(), ora["colsums"])
        assert np.allclose(tp["frac"].to_numpy(), ora["colsums"] / ora["colsums"][g], rtol=0, atol=0)
    # exported kmc/bitvec{i} are KMC1 files holding exactly the fixture's DBs
    for i, (fk, fm) in enumerate(dbs):
        db = po.read_kmc1(str(out / "kmc" / f"bitvec{i}"))
        assert np.array_equal(db["keys"], fk) and np.array_equal(db["counters"], fm)
    # read side (what `panagram view` does): query == unpackbits of the golden payload
    ridx = pidx.Index(str(out), mode="r")
    g = int(fx["anchors"][0])
    gen = ridx.genomes[f"g{g}"]
    gen.init_read()
    recs = po.parse_fasta_cpp(fx[f"fasta_{g}"].tobytes())
    nb = (n + 7) // 8
    rows = np.frombuffer(fx[f"a{g}_bitmap1"].tobytes(), np.uint8).reshape(-1, nb)
    bits = np.unpackbits(rows, axis=1, bitorder="little")[:, :n]
    off = 0
    for nm, seq in recs:
        nk = len(seq) - k + 1
        df = ridx.query_bitmap(f"g{g}", nm, 100, min(nk, 700))
        assert np.array_equal(df.to_numpy(), bits[off + 100: off + min(nk, 700)])
        df100 = ridx.query_bitmap(f"g{g}", nm, 0, nk, 100)
        assert np.array_equal(df100.to_numpy(), bits[off: off + nk: 100])
        off += nk
    # second run re-using the exported DBs (kmc.use_existing) gives identical payloads
    idx2 = pidx.Index(str(out), mode="w")
    idx2.kmc.use_existing = True
    idx2.genomes[anchors[0]].run_anchor(idx2.build_table())
    payload = gzip.open(out / "anchor" / anchors[0] / "bitmap.1.gz", "rb").read()
    assert payload == fx[f"a{int(fx['anchors'][0])}_bitmap1"].tobytes()
    idx2.close()


@pytest.mark.parametrize("layout,batch_bytes", [("kmc1", None), ("kmc2", None), ("kmc2", 1)])
def test_run_anchor_cli_matches_reference_binary(layout, batch_bytes, tmp_path, monkeypatch):
    """same argv + file contract as cpp/run_anchor, fed the same KMC files (either layout); the anchors share one
    co-scheduled launch (or, with a tiny HBM budget, one launch each, written while the next is anchored)"""
    from panagram_amd import index as pidx
    from panagram_amd.__main__ import main
    if batch_bytes is not None:
        monkeypatch.setattr(pidx.Index, "batch_bytes", batch_bytes)
    if layout == "kmc1":
        fx = H.load_case("n33_k16")
    else:
        k2 = H.load_case("kmc2_n40_k31")
        fx = H.load_case(str(k2["ref_case"]))
    n, k = int(fx["ngenomes"]), int(fx["k"])
    root = tmp_path / "root"
    (root / "kmc").mkdir(parents=True)
    for i, (keys, masks) in enumerate(H.case_dbs(fx)):
        if layout == "kmc1":
            po.write_kmc1(str(root / "kmc" / f"bitvec{i}"), keys, masks, k, lut_prefix_len=4)
        else:  # the very images the reference binary accepted
            k2[f"db{i}_pre"].tofile(str(root / "kmc" / f"bitvec{i}.kmc_pre"))
            k2[f"db{i}_suf"].tofile(str(root / "kmc" / f"bitvec{i}.kmc_suf"))
    args = ["run_anchor", str(n), str(root)]
    for g in fx["anchors"]:
        fa = tmp_path / f"g{g}.fa"
        fa.write_bytes(fx[f"fasta_{g}"].tobytes())
        args += [f"g{g}", str(fa)]
    assert main(args) == 0
    for g in fx["anchors"]:
        adir = root / "anchor" / f"g{g}"
        assert gzip.open(adir / "bitmap.1.gz", "rb").read() == fx[f"a{g}_bitmap1"].tobytes()
        assert gzip.open(adir / "bitmap.100.gz", "rb").read() == fx[f"a{g}_bitmap100"].tobytes()
        assert (adir / "bitsum.bins.tsv").read_bytes() == fx[f"a{g}_bitsum.bins.tsv"].tobytes()
        assert (adir / "chrs.tsv").read_bytes() == fx[f"a{g}_chrs.tsv"].tobytes()
        gzi = np.fromfile(adir / "bitmap.1.gzi", "<u8")
        assert gzi[0] == np.frombuffer(fx[f"a{g}_gzi1"].tobytes(), "<u8")[0]


def test_kmc_api_mirror(tmp_path):
    """py_kmc_api-shaped seam: KMCFile().OpenForRA / CountVec / GetCountersForRead"""
    from panagram_amd import kmc_api as py_kmc_api
    fx = H.load_case("n4_k21_minmax")
    k = int(fx["k"])
    keys, masks = H.case_dbs(fx)[0]
    p = str(tmp_path / "bitvec0")
    po.write_kmc1(p, keys, masks, k, min_count=int(fx["min_count"]), max_count=int(fx["max_count"]))
    db = py_kmc_api.KMCFile()
    assert db.OpenForRA(p) is True and db.KmerLength() == k
    assert py_kmc_api.KMCFile().OpenForRA(str(tmp_path / "missing")) is False
    for _, seq in po.parse_fasta_cpp(fx["fasta_3"].tobytes()):
        vec = py_kmc_api.CountVec()
        db.GetCountersForRead(seq.decode(), vec)
        pac32 = np.array(vec, dtype="uint32")  # exactly what index.py:937 does
        want = po.counters_for_read((keys, masks), seq, k, int(fx["min_count"]), int(fx["max_count"]))
        assert np.array_equal(pac32, want)
    db.Close()


# ---------------------------------------------------------------------------
# FASTA text parsed on the GPU, payloads streamed from HBM into BGZF files
# ---------------------------------------------------------------------------
_FASTA_CASES = {
    "plain": b">chr1 some description\nACGTACGTAC\nGGGTTTAAAC\nAC\n>chr2\nTTTTGGGGCCCCAAAA\n",
    "no_trailing_newline": b">a\nACGT\nAC",
    "crlf_and_blank_lines": b">a desc\r\nACGTNNNN\r\n\r\nacgtacgt\r\n>b\r\n\r\nGG\r\n",
    "junk_before_header": b"; comment\nACGT\n>x\nAAAA\nCC\n",
    "empty_record": b">e1\n>e2\nACGTTT\n>e3\n",
    "spaces_tabs_inside": b">s\nAC GT\tAC\x0bGT\x0cA\n",
    "gt_inside_header": b">h a>b >c\nACGT\n",
    "header_only_ws_name": b">   \nACGT\n>\t name2 x\nGG\n",
    "no_header": b"ACGT\nACGT\n",
    "empty": b"",
}


@pytest.mark.parametrize("case", sorted(_FASTA_CASES))
def test_gpu_fasta_parser_matches_host_reader(ctx, case, tmp_path):
    from panagram_amd import engine, index as pidx
    text = _FASTA_CASES[case]
    fa = tmp_path / "x.fa"
    fa.write_bytes(text)
    want = list(pidx.read_fasta(str(fa)))
    ss = engine.SeqSet.from_fasta(ctx, text)
    assert ss.names == [n for n, _ in want]
    assert [int(x) for x in ss.lens] == [len(s) for _, s in want]
    for i, (_, s) in enumerate(want):
        exp = bytes(c if c in b"ACGT" else ord("N") for c in s.upper())
        assert ss.unpack(i) == exp
    ss.close()
    ss2 = engine.SeqSet.from_fasta(ctx, str(fa))  # same through the file path
    assert ss2.names == [n for n, _ in want]
    ss2.close()


def test_gpu_fasta_header_lines_found_on_the_device(ctx, tmp_path, monkeypatch):
    """Texts of 4 MB and more have their header lines looked for on the device (k_text_headers): '>' at offset 0, behind
    '\\n' and '\\r\\n', NOT in the middle of a line or at a line's end; headers at every alignment of the 16-byte groups the
    kernel reads, next to each other, and as the text's last byte — the host's reader and the host's scan
    (PG_FASTA_HOST_SCAN=1) agree with it."""
    from panagram_amd import engine, index as pidx
    rng = np.random.default_rng(78)
    out = bytearray()
    recs = []
    for r in range(60):
        n = int(rng.choice([0, 1, 15, 16, 17, 33, 100003, 250001])) if r % 3 else 250001 + r
        seq = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=n).tobytes()
        if r % 5 == 1 and n > 50:  # a '>' inside a sequence line and at its end: not headers (bytes outside ACGT, like any other)
            seq = seq[:20] + b">" + seq[21:-1] + b">"
        pad = b"x" * int(rng.integers(0, 16))  # (moves the next '>' through the 16 alignments)
        out += b">r%d %s\n" % (r, pad)
        recs.append((f"r{r}", seq))
        p = 0
        while p < n:
            w = int(rng.integers(30, 90))
            out += seq[p:p + w] + (b"\r\n" if r % 7 == 3 else b"\n")
            p += w
    out += b">last_without_a_line_end"
    recs.append(("last_without_a_line_end", b""))
    out += b"\n>"  # ... and a '>' as the very last byte: a header without a name
    recs.append(("", b""))
    assert len(out) > (4 << 20)
    fa = tmp_path / "h.fa"
    fa.write_bytes(bytes(out))
    assert [(n, s) for n, s in pidx.read_fasta(str(fa))] == recs
    got = []
    for host_scan in (False, True):
        if host_scan:
            monkeypatch.setenv("PG_FASTA_HOST_SCAN", "1")
        ss = engine.SeqSet.from_fasta(ctx, bytes(out))
        got.append((list(ss.names), [int(x) for x in ss.lens], [ss.unpack(i) for i in range(len(ss.names))]))
        ss.close()
    assert got[0] == got[1]
    assert got[0][0] == [n for n, _ in recs] and got[0][1] == [len(s) for _, s in recs]
    for i, (_, s) in enumerate(recs):
        assert got[0][2][i] == bytes(c if c in b"ACGT" else ord("N") for c in s)


def test_gpu_fasta_parser_large_random_wrapping(ctx, tmp_path):
    """multi-chunk records, line widths from 1 to 200, N runs and lower case, .gz input"""
    import gzip
    from panagram_amd import engine, index as pidx
    rng = np.random.default_rng(77)
    recs = []
    out = bytearray()
    for r, n in enumerate([70001, 1, 4096, 4095, 123457, 31, 33]):
        seq = rng.choice(np.frombuffer(b"ACGTacgtN", np.uint8), size=n, p=[.22, .22, .22, .22, .02, .02, .02, .02, .04]).tobytes()
        recs.append((f"r{r}", seq))
        out += f">r{r} len={n}\n".encode()
        p = 0
        while p < n:
            w = int(rng.integers(1, 200))
            out += seq[p:p + w] + (b"\r\n" if rng.random() < 0.1 else b"\n")
            p += w
    fa = tmp_path / "big.fa.gz"
    with gzip.open(fa, "wb") as f:
        f.write(bytes(out))
    assert [(n, s) for n, s in pidx.read_fasta(str(fa))] == recs
    ss = engine.SeqSet.from_fasta(ctx, str(fa))
    assert ss.names == [n for n, _ in recs]
    for i, (_, s) in enumerate(recs):
        assert ss.unpack(i) == bytes(c if c in b"ACGT" else ord("N") for c in s.upper())
    ss.close()


def test_fasta_files_read_into_page_locked_buffers_or_pageable_arrays_give_the_same_tree(tmp_path, monkeypatch):
    """Index.load_inputs reads plain FASTA files into a pool of page-locked buffers (engine.HostBuffer: fewer buffers than
    files here, so they go round; files of different sizes) and compressed ones, or everything with PG_PINNED_READS=0, into
    pageable arrays: the same index either way — and the golden one."""
    from panagram_amd import engine, index as pidx
    fx = H.load_case("n9_k21")
    k = int(fx["k"])
    s = _write_case(tmp_path, fx)
    lines = s.read_text().splitlines()
    name, fa = lines[3].split("\t")  # one sample as .gz: read through gzip into a pageable array beside the pooled ones
    with open(fa, "rb") as f, gzip.open(fa + ".gz", "wb") as z:
        z.write(f.read())
    lines[3] = f"{name}\t{fa}.gz"
    s.write_text("\n".join(lines) + "\n")
    monkeypatch.setenv("PG_PINNED_BUFFERS", "3")  # (three buffers for nine files)
    monkeypatch.setenv("PG_PINNED_READS", "2")  # (whatever the files' size: the pool is for jobs of 256 MB and more)
    pidx.Index(str(s), prefix=str(tmp_path / "pinned"), k=k).run()
    monkeypatch.setenv("PG_PINNED_READS", "0")
    pidx.Index(str(s), prefix=str(tmp_path / "pageable"), k=k).run()
    _same_index_payload(str(tmp_path / "pinned"), str(tmp_path / "pageable"))
    for g in fx["anchors"]:
        assert gzip.open(tmp_path / "pinned" / "anchor" / f"g{g}" / "bitmap.1.gz", "rb").read() == fx[f"a{g}_bitmap1"].tobytes()
    # the buffer itself: a numpy view of page-locked memory that the parser takes like any other text
    ctx = engine.Context(0)
    text = fx["fasta_0"].tobytes()
    hb = ctx.host_buffer(len(text) + 100)
    hb.array[:len(text)] = np.frombuffer(text, np.uint8)
    a, b = engine.SeqSet.from_fasta(ctx, hb.array[:len(text)]), engine.SeqSet.from_fasta(ctx, text)
    assert a.names == b.names and np.array_equal(a.lens, b.lens)
    for x in (a, b, hb, hb):  # (closing a buffer twice is harmless)
        x.close()
    ctx.close()


def test_write_bgzf_from_hbm_equals_host_writer(ctx, tmp_path):
    import gzip
    from panagram_amd import engine
    n, k = 9, 21
    gen = po.synth_genomes(n, [300000, 70000, 25], 0.02, 31)
    genomes = [[po.codes_to_ascii(c) for c in g] for g in gen]
    tbl = engine.PanTable(ctx, k, n)
    for g in range(n):
        ss = engine.SeqSet.from_host(ctx, genomes[g])
        tbl.insert_seqset(g, ss)
        ss.close()
    ss = engine.SeqSet.from_host(ctx, genomes[2])
    res = engine.AnchorResult(tbl, ss)
    res.run()
    parts = [res.download(ci) for ci in range(3)]
    for step, col in ((1, 0), (100, 1)):
        gz, gzi = str(tmp_path / f"b{step}.gz"), str(tmp_path / f"b{step}.gzi")
        res.write_bgzf(step, gz, gzi, level=6, threads=3)
        # (write_bgzf picks the encoder by row width: 2-byte rows here -> the row-aware one)
        w = engine.BgzfWriter(str(tmp_path / f"h{step}.gz"), level=6 | engine.BgzfWriter.ROWS(2), threads=3)
        for p in parts:
            w.write(p[col])
        w.close(str(tmp_path / f"h{step}.gzi"))
        payload = b"".join(p[col].tobytes() for p in parts)
        assert gzip.open(gz, "rb").read() == payload
        assert open(gz, "rb").read() == open(tmp_path / f"h{step}.gz", "rb").read()
        assert open(gzi, "rb").read() == open(tmp_path / f"h{step}.gzi", "rb").read()
        # level -2: the blocks are compressed on the GPU (k_row_deflate) — other bytes, same payload and
        # the same uncompressed geometry; and its stored-block fallback
        for stored in (False, True):
            if stored:
                os.environ["PG_DEFLATE_FORCE_STORED"] = "1"
            try:
                gz2, gzi2 = str(tmp_path / f"g{step}{stored}.gz"), str(tmp_path / f"g{step}{stored}.gzi")
                res.write_bgzf(step, gz2, gzi2, level=-2)
            finally:
                os.environ.pop("PG_DEFLATE_FORCE_STORED", None)
            assert gzip.open(gz2, "rb").read() == payload
            a, b = np.fromfile(gzi2, "<u8"), np.fromfile(gzi, "<u8")
            assert a[0] == b[0] and np.array_equal(a[2::2], b[2::2])
            if not stored and step == 1:
                assert os.path.getsize(gz2) < 0.5 * len(payload)
    res.close()
    ss.close()
    tbl.close()


def test_fastq_sample_is_counted_with_ci2(tmp_path):
    """a FASTQ sample (non-anchor) enters the pan table with kmc -ci2 semantics
    (workflow/Snakefile:88-89); the anchor's bitmap equals the oracle's"""
    import gzip as gz
    from panagram_amd import index as pidx
    from tests.test_gpu_parity import _simulate_reads
    k = 21
    rng = np.random.default_rng(3)
    gen = po.synth_genomes(2, [9000, 4000], 0.02, 99)
    asm = [po.codes_to_ascii(c) for c in gen[0]]
    reads = _simulate_reads(rng, po.codes_to_ascii(gen[1][0]), 300, 120, 0.01)
    fa = tmp_path / "a.fa"
    fa.write_bytes(po.fasta_text(["c1", "c2"], asm))
    fq = tmp_path / "r.fq.gz"
    with gz.open(fq, "wb") as f:
        for i, r in enumerate(reads):
            f.write(b"@r%d\n" % i + r + b"\n+\n" + b"I" * len(r) + b"\n")
    (tmp_path / "samples.tsv").write_text(f"name\tfasta\tanchor\nasm\t{fa}\tTrue\nreads\t{fq}\tFalse\n")
    idx = pidx.Index(str(tmp_path / "samples.tsv"), prefix=str(tmp_path / "idx"), k=k)
    idx.run()
    dbs = po.build_bitvec_dbs([asm, reads], k, min_counts=[1, 2])
    want = b"".join(po.anchor_contig(dbs, s, k, 2)[0].tobytes() for s in asm)
    got = gzip.open(tmp_path / "idx" / "anchor" / "asm" / "bitmap.1.gz").read()
    assert got == want
    assert not (tmp_path / "idx" / "anchor" / "reads").exists()


def test_gene_tabulation_from_gff(tmp_path):
    """bitsum.genes.tsv and chrs.tsv gene_count for an annotated anchor (index.py:1027-1033,
    1055-1064, 1079-1082): per-gene occupancy counted on the GPU equals the oracle's rows"""
    from panagram_amd import index as pidx
    k, n = 21, 3
    gen = po.synth_genomes(n, [20000, 8000, 1000], 0.02, 5)
    genomes = [[po.codes_to_ascii(c) for c in g] for g in gen]
    names = ["chrB", "chrA", "chrC"]  # the GFF is tabulated in sorted-chr order, the FASTA is not sorted
    rows = ["name\tfasta\tgff"]
    for g in range(n):
        fa = tmp_path / f"g{g}.fa"
        fa.write_bytes(po.fasta_text(names, genomes[g]))
        rows.append(f"g{g}\t{fa}\t" + (str(tmp_path / "g0.gff") if g == 0 else ""))
    genes = [("chrA", 100, 900), ("chrA", 850, 2000), ("chrB", 1, 19000), ("chrB", 500, 400),  # end <= start: skipped
             ("chrB", 19000, 20001), ("chrZ", 10, 20), ("chrC", 0, 980)]                       # past the end / unknown chr
    with open(tmp_path / "g0.gff", "w") as f:
        f.write("##gff-version 3\n")
        for c, s, e in genes:
            f.write(f"{c}\tsrc\tgene\t{s}\t{e}\t.\t+\t.\tID=g{s};Name=n{s}\n")
            f.write(f"{c}\tsrc\texon\t{s}\t{e}\t.\t+\t.\tParent=g{s}\n")
    (tmp_path / "samples.tsv").write_text("\n".join(rows) + "\n")
    idx = pidx.Index(str(tmp_path / "samples.tsv"), prefix=str(tmp_path / "idx"), k=k, anchor_genomes=["g0"])
    idx.run()
    dbs = po.build_bitvec_dbs(genomes, k)
    want = {}
    for c, s, e in genes:
        want.setdefault(c, np.zeros(n + 1, np.int64))
        if c in names:
            o_rows = po.anchor_contig(dbs, genomes[0][names.index(c)], k, n)[0]
            if e > s and s >= 0 and e <= len(o_rows):
                want[c] += po.window_stats(o_rows, n, [s], [e])[0][0]
    got = pd.read_table(tmp_path / "idx" / "anchor" / "g0" / "bitsum.genes.tsv").set_index("chr")
    assert list(got.index) == sorted(want)
    for c in want:
        assert np.array_equal(got.loc[c].to_numpy(), want[c])
    chrs = pd.read_table(tmp_path / "idx" / "anchor" / "g0" / "chrs.tsv").set_index("name")
    assert chrs["gene_count"].to_dict() == {"chrB": 3, "chrA": 2, "chrC": 1}


def _messy_pangenome(rng, n):
    """derived genomes with substitutions, indels, a repeat family, N runs, soft-masked stretches,
    contigs of unequal length — and a different contig order in one genome"""
    elem = rng.integers(0, 4, 600, dtype=np.uint8)
    base = []
    for L in (70000, 26000, 9000):
        parts, left = [], L
        while left > 0:
            parts.append(rng.integers(0, 4, int(rng.integers(500, 3000)), dtype=np.uint8))
            e = elem.copy()
            mut = rng.random(len(e)) < 0.06
            e[mut] = (e[mut] + rng.integers(1, 4, int(mut.sum()), dtype=np.uint8)) % 4
            parts.append(e)
            left -= len(parts[-1]) + len(parts[-2])
        base.append(np.concatenate(parts))
    acgt = np.frombuffer(b"ACGT", np.uint8)
    genomes = []
    for g in range(n):
        contigs = []
        for b in base:
            x = b.copy()
            if g:
                mut = rng.random(len(x)) < 0.015
                x[mut] = (x[mut] + rng.integers(1, 4, int(mut.sum()), dtype=np.uint8)) % 4
                pieces, cur = [], 0
                for p in np.sort(rng.integers(0, len(x), 12)):
                    if p < cur:
                        continue
                    pieces.append(x[cur:p])
                    ln = int(rng.integers(1, 400))
                    if rng.random() < 0.5:
                        cur = min(len(x), p + ln)
                    else:
                        pieces.append(rng.integers(0, 4, ln, dtype=np.uint8))
                        cur = p
                pieces.append(x[cur:])
                x = np.concatenate(pieces)
            s = bytearray(acgt[x].tobytes())
            for _ in range(2):
                p = int(rng.integers(0, len(s) - 300))
                s[p:p + int(rng.integers(1, 200))] = b"N" * 1  # a single N and longer runs below
                q = int(rng.integers(0, len(s) - 300))
                s[q:q + 150] = b"N" * 150
                u = int(rng.integers(0, len(s) - 2000))
                s[u:u + 1500] = bytes(s[u:u + 1500]).lower()
            contigs.append(bytes(s))
        genomes.append(contigs)
    return genomes


def test_messy_pangenome_through_index_run(tmp_path):
    """indels, repeats, N runs, soft masking, unequal contigs and a shuffled contig order in one
    genome: every anchor's files from the co-scheduled Index.run() equal the oracle's"""
    from panagram_amd import index as pidx
    rng = np.random.default_rng(2024)
    n, k = 6, 21
    genomes = _messy_pangenome(rng, n)
    names = ["chr1", "chr2", "chr3"]
    order = {g: [0, 1, 2] for g in range(n)}
    order[3] = [2, 0, 1]
    rows = ["name\tfasta"]
    fastas = {}
    for g in range(n):
        fastas[g] = po.fasta_text([names[i] for i in order[g]], [genomes[g][i] for i in order[g]], width=61 + g)
        if g % 2:  # every other input gzip-compressed (read ahead by the host threads like the plain ones)
            fa = tmp_path / f"g{g}.fa.gz"
            with gzip.open(fa, "wb") as f:
                f.write(fastas[g])
        else:
            fa = tmp_path / f"g{g}.fa"
            fa.write_bytes(fastas[g])
        rows.append(f"g{g}\t{fa}")
    (tmp_path / "samples.tsv").write_text("\n".join(rows) + "\n")
    idx = pidx.Index(str(tmp_path / "samples.tsv"), prefix=str(tmp_path / "idx"), k=k)
    idx.run()
    dbs = po.build_bitvec_dbs([[genomes[g][i] for i in order[g]] for g in range(n)], k)
    for g in range(n):
        ora = po.anchor_fasta(dbs, fastas[g], k, n)
        adir = tmp_path / "idx" / "anchor" / f"g{g}"
        assert gzip.open(adir / "bitmap.1.gz", "rb").read() == ora["bitmap1"]
        assert gzip.open(adir / "bitmap.100.gz", "rb").read() == ora["bitmap100"]
        assert (adir / "bitsum.bins.tsv").read_text() == ora["bins_tsv"]
        assert (adir / "chrs.tsv").read_text() == ora["chrs_tsv"]
        tp = pd.read_csv(adir / "total_paircounts.csv", index_col="name")
        assert np.array_equal(tp["count"].to_numpy(), ora["colsums"])


def test_gpu_bgzf_multi_batch(ctx, tmp_path):
    """a payload of more than 1024 BGZF blocks goes through several k_row_deflate launches and both
    staging buffers; odd tail block"""
    from panagram_amd import engine
    rng = np.random.default_rng(8)
    L = 70_000_123
    a = rng.integers(0, 4, L, dtype=np.uint8)
    b = a.copy()
    mut = rng.random(L) < 0.02
    b[mut] = (b[mut] + 1) & 3
    acgt = np.frombuffer(b"ACGT", np.uint8)
    ga, gb = acgt[a].tobytes(), acgt[b].tobytes()
    tbl = engine.PanTable(ctx, 21, 2, expected_keys=int(L * 1.5))
    sa, sb = engine.SeqSet.from_host(ctx, [ga]), engine.SeqSet.from_host(ctx, [gb])
    tbl.insert_seqset(0, sa)
    tbl.insert_seqset(1, sb)
    res = engine.AnchorResult(tbl, sb)
    res.run()
    rows = res.download(0)[0]
    gz, gzi = str(tmp_path / "m.gz"), str(tmp_path / "m.gzi")
    res.write_bgzf(1, gz, gzi, level=-2)
    assert gzip.open(gz, "rb").read() == rows.tobytes()
    g = np.fromfile(gzi, "<u8")
    nblocks = (len(rows) + 65279) // 65280
    assert nblocks > 1024 and g[0] == nblocks - 1 and np.array_equal(g[2::2], np.arange(1, nblocks, dtype=np.uint64) * 65280)
    assert np.array_equal(g[1::2][1:] > g[1::2][:-1], np.ones(nblocks - 2, bool))
    for x in (res, sa, sb, tbl):
        x.close()


@pytest.mark.parametrize("ngenomes", [8, 27])
def test_gpu_bgzf_one_code_per_file_covers_what_the_sample_did_not_see(ctx, ngenomes, tmp_path):
    """k_row_deflate codes a whole file with ONE Huffman code built from a sample of its blocks (every fourth block here: 2048
    blocks, 512 sampled).  Blocks the sample never saw hold byte values, run lengths and incompressible stretches of their own:
    every symbol must have a code (k_df_build_code counts each at least once), a block that does not fit falls back to a stored
    block, and the file inflates to the rows."""
    import torch
    from panagram_amd import engine
    nb = (ngenomes + 7) // 8
    npos = 2048 * 65280 // nb + 12345
    ss = engine.SeqSet(ctx, [npos + 20])
    res = engine.AnchorResult.rows_container(ctx, 21, ngenomes, ss)
    rows = res.rows_tensor()[:npos * nb]  # (the buffer is padded behind the last row)
    gen = torch.Generator(device=rows.device)
    gen.manual_seed(5)
    # what the sampled blocks look like: two row values in long runs
    base = torch.where(torch.rand(rows.numel() // (64 * nb) + 1, device=rows.device, generator=gen) < 0.3, 0x81, 0x7F).to(torch.uint8)
    rows.copy_(base.repeat_interleave(64 * nb)[:rows.numel()])
    blk = 65280
    # unsampled blocks (the sample takes blocks 0, 4, 8, ...): every byte value, incompressible
    for b in (1, 2, 1027):
        rows[b * blk:(b + 1) * blk] = torch.randint(0, 256, (blk,), dtype=torch.uint8, device=rows.device, generator=gen)
    # ... every run length 1..300 between rare separator bytes (all 29 length symbols), in an unsampled block
    lens = torch.arange(1, 301, device=rows.device)
    runs = torch.full((int(lens.sum()) + 300,), 0x33, dtype=torch.uint8, device=rows.device)
    runs[torch.cumsum(lens + 1, 0) - 1] = torch.arange(300, device=rows.device).to(torch.uint8) | 0x80
    rows[5 * blk:5 * blk + min(blk, runs.numel())] = runs[:blk]
    # ... and rare literals sprinkled over another one
    idx = 9 * blk + torch.arange(0, blk, 97, device=rows.device)
    rows[idx] = (torch.arange(idx.numel(), device=rows.device) % 251 + 3).to(torch.uint8)
    torch.cuda.synchronize()
    want = rows.cpu().numpy().tobytes()
    res.rows_epilogue()  # (a rows container is written once its statistics have been enqueued)
    gz, gzi = str(tmp_path / "h.gz"), str(tmp_path / "h.gzi")
    res.write_bgzf(1, gz, gzi, level=-2)
    assert gzip.open(gz, "rb").read() == want
    g = np.fromfile(gzi, "<u8")
    nblocks = (len(want) + 65279) // 65280
    assert nblocks > 2048 and g[0] == nblocks - 1 and np.array_equal(g[2::2], np.arange(1, nblocks, dtype=np.uint64) * 65280)
    assert os.path.getsize(gz) < 0.05 * len(want)  # (the long runs compress; the three random blocks are stored)
    res.close()
    ss.close()


def _same_index_payload(one, many, steps=(1, 100)):
    """two index trees hold the same index: decompressed bitmaps and every table byte for byte (the compressed
    bytes and the .gzi follow the BGZF block boundaries, which may differ), the .gzi addressing its own payload"""
    from panagram_amd import index as pidx
    for dirpath, _, files in os.walk(one):
        if os.path.basename(dirpath) == "logs":
            continue
        for f in files:
            a = os.path.join(dirpath, f)
            b = os.path.join(many, os.path.relpath(a, one))
            if f.endswith(".gzi"):
                continue
            if f.endswith(".gz"):
                raw = gzip.open(b, "rb").read()
                assert gzip.open(a, "rb").read() == raw, f
                blocks = pidx.load_bgz_blocks(b + "i")
                for start in (0, len(raw) // 3, max(0, len(raw) - 5)):
                    assert pidx.bgzf_read(b, blocks, start, 5) == raw[start:start + 5]
            else:
                assert open(a, "rb").read() == open(b, "rb").read(), f
    assert not [d for d, _, _ in os.walk(many) if os.path.basename(d) == ".parts"]


@pytest.mark.parametrize("name,world,partition", [("n9_k21", 2, "pieces"), ("n40_k31", 3, "pieces"), ("n9_k21", 2, "genomes")])
def test_index_run_on_several_ranks_writes_the_same_tree(name, world, partition, tmp_path, monkeypatch):
    """Multi-GPU Index.run (one process per GPU; here the ranks run one after the other on the one GPU of the box).
    Default partition: pieces of homology classes — every rank anchors its share of EVERY genome, whoever completes a
    genome assembles it (the last rank at the latest).  PG_PARTITION=genomes: whole anchor genomes dealt to the ranks.
    Either way the index is the single-rank one — it does not depend on the GPU count."""
    from panagram_amd import index as pidx
    monkeypatch.setenv("PG_PARTITION", partition)
    fx = H.load_case(name)
    k = int(fx["k"])
    anchors = [f"g{g}" for g in fx["anchors"]]
    s = _write_case(tmp_path, fx)
    one = tmp_path / "one"
    pidx.Index(str(s), prefix=str(one), k=k, anchor_genomes=anchors).run()
    many = tmp_path / "many"
    seen = []
    for rank in range(world):
        idx = pidx.Index(str(s), prefix=str(many), k=k, anchor_genomes=anchors, rank=rank, world=world)
        mine = idx.my_anchor_genomes()
        seen += mine
        idx.run()
        if partition == "genomes":
            for nm in mine:  # a rank leaves exactly its genomes' directories complete
                assert (many / "anchor" / nm / "total_paircounts.csv").exists()
    assert sorted(seen) == sorted(anchors) and len(seen) == len(anchors)
    if partition == "genomes":
        for dirpath, _, files in os.walk(one):
            if os.path.basename(dirpath) == "logs":
                continue
            for f in files:
                a = os.path.join(dirpath, f)
                b = os.path.join(many, os.path.relpath(a, one))
                assert open(a, "rb").read() == open(b, "rb").read(), f
    else:
        _same_index_payload(str(one), str(many))


def test_seqset_slice_and_pieces_equal_whole_contigs(ctx):
    """pg_seqset_slice: pieces of packed contigs (starts on 32-base words) unpack to the contig's bases; anchored with
    k - 1 bases of overlap they give exactly the contig's rows — N runs and lower case across the cuts included."""
    from panagram_amd import engine
    rng = np.random.default_rng(9)
    k, n = 21, 5
    gen = po.synth_genomes(n, [9000, 2100], 0.02, 31)
    genomes = [[po.codes_to_ascii(c) for c in g] for g in gen]
    a = bytearray(genomes[2][0])
    a[3190:3215] = b"N" * 25          # an N run across the cut at 3200
    a[6380:6420] = bytes(a[6380:6420]).lower()
    genomes[2][0] = bytes(a)
    tbl = engine.PanTable(ctx, k, n)
    for g in range(n):
        ss = engine.SeqSet.from_host(ctx, genomes[g])
        tbl.insert_seqset(g, ss)
        ss.close()
    whole = engine.SeqSet.from_host(ctx, genomes[2])
    cuts = [(0, 0, 3200 + k - 1), (0, 3200, 3200 + k - 1), (0, 6400, 9000 - 6400), (1, 0, 2100), (1, 2080, 20), (0, 8992, 8)]
    pieces = whole.slice(cuts)
    assert list(pieces.lens) == [c[2] for c in cuts]
    want = [whole.unpack(0), whole.unpack(1)]
    for i, (ci, s0, ln) in enumerate(cuts):
        assert pieces.unpack(i) == want[ci][s0:s0 + ln]
    r0 = engine.AnchorResult(tbl, whole, colsums=True)
    r0.run()
    r1 = engine.AnchorResult(tbl, pieces, colsums=True)
    r1.run()
    rows = r0.download(0)[0]
    got = np.concatenate([r1.download(i)[0] for i in range(3)])
    assert np.array_equal(got, rows)
    assert np.array_equal(r1.download(3)[0], r0.download(1)[0])
    assert r1.download(4)[0].shape[0] == 0 and r1.download(5)[0].shape[0] == 0  # pieces shorter than k: no rows
    with pytest.raises(engine.PanagramHipError):
        whole.slice([(0, 100, 50)])  # not on a 32-base word
    with pytest.raises(engine.PanagramHipError):
        whole.slice([(0, 8992, 100)])  # past the end
    for x in (r0, r1, pieces, whole, tbl):
        x.close()


def test_index_run_pieces_cut_chromosomes_on_the_gpu(tmp_path, monkeypatch):
    """The default multi-rank partition on the real engine with chromosomes long enough to be cut (small minimum piece):
    3 ranks one after the other; pieces that start inside a bin, genes across cuts, a genome with other record ids."""
    from panagram_amd import distributed as pdist
    from panagram_amd import index as pidx
    from tests.test_distributed_cpu import _piece_pangenome
    s = _piece_pangenome(tmp_path)
    geo = dict(k=21, lowres_step=50, max_bin_kbp=3, min_bin_count=5)
    pidx.Index(str(s), prefix=str(tmp_path / "one"), **geo).run()
    monkeypatch.setattr(pdist, "MIN_PIECE", 3000)
    for rank in range(3):
        pidx.Index(str(s), prefix=str(tmp_path / "many"), rank=rank, world=3, **geo).run()
    _same_index_payload(str(tmp_path / "one"), str(tmp_path / "many"), steps=(1, 50))
    assert (tmp_path / "many" / "anchor" / "g0" / "bitsum.genes.tsv").exists()


@pytest.mark.parametrize("n", [3, 12, 20, 27, 33, 50, 64, 96, 130, 300])
def test_gpu_bgzf_every_row_width(ctx, n, tmp_path):
    """k_row_deflate on rows of 1..38 bytes (matches at distance = row width, unaligned for odd widths):
    the decompressed files equal the rows, for bitmap.1 and bitmap.100, and compress"""
    import gzip
    from panagram_amd import engine
    k = 21
    gen = po.synth_genomes(n, [40000, 9000], 0.01, 500 + n)
    genomes = [[po.codes_to_ascii(c) for c in g] for g in gen]
    tbl = engine.PanTable(ctx, k, n)
    for g in range(n):
        ss = engine.SeqSet.from_host(ctx, genomes[g])
        tbl.insert_seqset(g, ss)
        ss.close()
    ss = engine.SeqSet.from_host(ctx, genomes[n // 2])
    res = engine.AnchorResult(tbl, ss)
    res.run()
    parts = [res.download(ci) for ci in range(2)]
    for step, col in ((1, 0), (100, 1)):
        gz, gzi = str(tmp_path / f"b{step}.gz"), str(tmp_path / f"b{step}.gzi")
        res.write_bgzf(step, gz, gzi, level=-2)
        payload = b"".join(p[col].tobytes() for p in parts)
        assert gzip.open(gz, "rb").read() == payload
        if step == 1:
            assert os.path.getsize(gz) < 0.6 * len(payload)
    res.close()
    ss.close()
    tbl.close()


def test_degenerate_inputs_produce_valid_empty_outputs(tmp_path):
    """An empty FASTA, a header without sequence, sequence without a header, contigs shorter than k and an
    all-N contig as anchors: no crash, valid (possibly empty) files."""
    from panagram_amd import index as pidx
    rng = np.random.default_rng(77)
    good = b">c1\n" + po.codes_to_ascii(rng.integers(0, 4, 3000, dtype=np.uint8)) + b"\n"
    files = {"a": good, "empty": b"", "hdr": b">x\n", "nohdr": b"ACGTACGTACGTACGTACGTACGTACGTACGT\n",
             "short": b">x\nACGT\n>y\nACGTACGT\n", "alln": b">n\n" + b"N" * 500 + b"\n"}
    rows = ["name\tfasta"]
    for nm, txt in files.items():
        (tmp_path / f"{nm}.fa").write_bytes(txt)
        rows.append(f"{nm}\t{tmp_path / (nm + '.fa')}")
    (tmp_path / "s.tsv").write_text("\n".join(rows) + "\n")
    pidx.Index(str(tmp_path / "s.tsv"), prefix=str(tmp_path / "o"), k=21).run()
    nb = 1
    want = {"a": 3000 - 20, "empty": 0, "hdr": 0, "nohdr": 0, "short": 0, "alln": 480}
    for nm, nk in want.items():
        adir = tmp_path / "o" / "anchor" / nm
        payload = gzip.open(adir / "bitmap.1.gz", "rb").read()
        assert len(payload) == nk * nb, nm
        assert len(gzip.open(adir / "bitmap.100.gz", "rb").read()) == ((nk + 99) // 100) * nb, nm
        chrs = pd.read_table(adir / "chrs.tsv")
        assert int(chrs["size"].clip(lower=0).sum()) == nk, nm
        if nm == "alln":
            assert not any(payload)
    tp = pd.read_csv(tmp_path / "o" / "anchor" / "a" / "total_paircounts.csv", index_col="name")
    assert int(tp.loc["a", "count"]) == 2980 and int(tp.loc["alln", "count"]) == 0


def _bars_loop(occ, num_samples, bin_size, step):
    """scripts/make_bins_bits.py:34-59 as the plain loop it is"""
    x, cntr = [], bin_size
    while cntr < len(occ) * step:
        x.append(cntr)
        cntr += bin_size
    z_unique, z_univ = [0] * (len(x) + 1), [0] * (len(x) + 1)
    c = 0
    for i in occ:
        t = int(c / bin_size)
        if i == 1:
            z_unique[t] += 1
        elif i == num_samples:
            z_univ[t] += 1
        c += step
    return x, z_univ[:len(x)], z_unique[:len(x)]


@pytest.mark.parametrize("name", ["n9_k21", "n2_k21"])
def test_bins_bits_on_a_hip_written_index(name, tmp_path, capsys):
    """the make_bins_bits.py counterpart reads an index the HIP path wrote (bitmap.100.gz / .gzi
    through the reference's addressing rule) and must give what the script's loop gives on the golden
    bitmap.100 payload of the reference binary."""
    from panagram_amd import bins_bits
    from panagram_amd import index as pidx
    fx = H.load_case(name)
    n, k = int(fx["ngenomes"]), int(fx["k"])
    s = _write_case(tmp_path, fx)
    out = tmp_path / "idx"
    pidx.Index(str(s), prefix=str(out), k=k, anchor_genomes=[f"g{g}" for g in fx["anchors"]]).run()
    ridx = pidx.Index(str(out), mode="r")
    nb = (n + 7) // 8
    for g in fx["anchors"]:
        rows100 = np.frombuffer(fx[f"a{g}_bitmap100"].tobytes(), np.uint8).reshape(-1, nb)
        occ_all = np.unpackbits(rows100, axis=1, bitorder="little")[:, :n].sum(axis=1)
        off = 0
        for nm, seq in po.parse_fasta_cpp(fx[f"fasta_{g}"].tobytes()):
            n100 = (len(seq) - k + 1 + 99) // 100
            occ = occ_all[off:off + n100]
            off += n100
            for bin_size in (200000, 1000, 300):
                assert bins_bits.chromosome_bars(ridx, f"g{g}", nm, bin_size, 100) == _bars_loop(occ, n, bin_size, 100)
    # the command-line form prints the script's three comma-terminated lines per chromosome
    g0 = f"g{int(fx['anchors'][0])}"
    capsys.readouterr()  # (drop what Index.run() printed)
    assert bins_bits.main([str(out), g0]) == 0
    # (the fixtures' contigs are shorter than the default 200 kb window: three empty lines each, as the script prints)
    assert capsys.readouterr().out.count("\n") == 3 * len(po.parse_fasta_cpp(fx[f"fasta_{int(fx['anchors'][0])}"].tobytes()))


@pytest.mark.parametrize("name,step,kbp,minbins", [("n9_k21", 10, 1, 7), ("n40_k31", 250, 2, 3), ("n2_k21", 7, 200, 100)])
def test_index_run_with_non_default_resolution_and_bins(name, step, kbp, minbins, tmp_path):
    """lowres_step / max_bin_kbp / min_bin_count of the Python path (index.py:101-106, 1169-1183): bitmap.<step> =
    every step-th row of the golden bitmap.1, bins of max_bin_kbp*1000 positions unless the contig would get fewer
    than min_bin_count of them; the read side addresses bitmap.<step> by the same rule."""
    from panagram_amd import index as pidx
    fx = H.load_case(name)
    n, k = int(fx["ngenomes"]), int(fx["k"])
    s = _write_case(tmp_path, fx)
    out = tmp_path / "idx"
    idx = pidx.Index(str(s), prefix=str(out), k=k, anchor_genomes=[f"g{g}" for g in fx["anchors"]],
                     lowres_step=step, max_bin_kbp=kbp, min_bin_count=minbins)
    idx.run()
    ridx = pidx.Index(str(out), mode="r")
    assert ridx.steps == (1, step)
    nb = (n + 7) // 8
    for g in fx["anchors"]:
        adir = out / "anchor" / f"g{g}"
        rows = np.frombuffer(fx[f"a{g}_bitmap1"].tobytes(), np.uint8).reshape(-1, nb)
        assert gzip.open(adir / "bitmap.1.gz", "rb").read() == rows.tobytes()
        low, lines, off = [], ["chr\tstart" + "".join(f"\t{i}" for i in range(n + 1)) + "\n"], 0
        for ci, (nm, seq) in enumerate(po.parse_fasta_cpp(fx[f"fasta_{g}"].tobytes())):
            nk = len(seq) - k + 1
            r = rows[off:off + nk]
            off += nk
            low.append(r[::step].tobytes())
            binlen = kbp * 1000
            if nk // binlen < minbins:
                binlen = nk // minbins
            popc = np.unpackbits(r, axis=1, bitorder="little")[:, :n].sum(axis=1)
            for b0 in range(0, nk, binlen):
                h = np.bincount(popc[b0:b0 + binlen], minlength=n + 1)
                lines.append(f"{ci}\t{b0}" + "".join(f"\t{c}" for c in h) + "\n")
            df = ridx.query_bitmap(f"g{g}", nm, 0, nk, step)
            assert np.array_equal(df.to_numpy(), np.unpackbits(r[::step], axis=1, bitorder="little")[:, :n])
        assert gzip.open(adir / f"bitmap.{step}.gz", "rb").read() == b"".join(low)
        assert (adir / "bitsum.bins.tsv").read_text() == "".join(lines)


def test_mean_launch_timing_over_several_runs(ctx):
    """pg_result_timing_mean: every run since the reset is counted (bench.py's avg_launch_ms)"""
    from panagram_amd import engine
    fx = H.load_case("n9_k21")
    t = engine.PanTable(ctx, int(fx["k"]), int(fx["ngenomes"]))
    for d, (keys, masks) in enumerate(H.case_dbs(fx)):
        t.insert_keys(d, keys, masks)
    ss = engine.SeqSet.from_fasta(ctx, fx["fasta_0"].tobytes())
    res = engine.AnchorResult(t, ss, colsums=True)
    for _ in range(3):
        res.run()
    res.timing_reset()
    assert res.timing_mean()[2] == 0
    for _ in range(150):  # more than the event ring holds
        res.run()
    p_ms, e_ms, n = res.timing_mean()
    assert n == 150 and p_ms > 0 and e_ms > 0
    last = res.timing()
    assert 0 < last[0] < 50 * p_ms
    assert res.timing_mean()[2] == 150  # reading does not consume
    res.close(); ss.close(); t.close()


@pytest.mark.parametrize("name", ["n9_k21", "n65_k21", "n2_k21"])
def test_contig_sharded_fragments_on_the_gpu(name, tmp_path):
    """run_index_sharded (fine-grained contig-sharded mode) with one rank: every (genome, contig) unit leaves BGZF
    fragments compressed on the GPU out of HBM, the owner concatenates them; decompressed payloads, tables and the
    read side equal the reference's"""
    from panagram_amd import index as pidx
    from panagram_amd.distributed import run_index_sharded
    fx = H.load_case(name)
    n, k = int(fx["ngenomes"]), int(fx["k"])
    s = _write_case(tmp_path, fx)
    out = tmp_path / "idx"
    idx = pidx.Index(str(s), prefix=str(out), k=k, anchor_genomes=[f"g{g}" for g in fx["anchors"]])
    run_index_sharded(idx, 0, 1, lambda: None)
    idx.close()
    dbs = H.case_dbs(fx)
    for g in fx["anchors"]:
        adir = out / "anchor" / f"g{g}"
        for step in (1, 100):
            assert gzip.open(adir / f"bitmap.{step}.gz", "rb").read() == fx[f"a{g}_bitmap{step}"].tobytes()
        assert (adir / "bitsum.bins.tsv").read_bytes() == fx[f"a{g}_bitsum.bins.tsv"].tobytes()
        assert (adir / "chrs.tsv").read_bytes() == fx[f"a{g}_chrs.tsv"].tobytes()
        ora = po.anchor_fasta(dbs, fx[f"fasta_{g}"].tobytes(), k, n)
        tp = pd.read_csv(adir / "total_paircounts.csv", index_col="name")
        assert np.array_equal(tp["count"].to_numpy(), ora["colsums"])
        assert not (adir / ".parts").exists()
    ridx = pidx.Index(str(out), mode="r")
    g = int(fx["anchors"][0])
    nb = (n + 7) // 8
    rows = np.frombuffer(fx[f"a{g}_bitmap1"].tobytes(), np.uint8).reshape(-1, nb)
    bits = np.unpackbits(rows, axis=1, bitorder="little")[:, :n]
    off = 0
    for nm, seq in po.parse_fasta_cpp(fx[f"fasta_{g}"].tobytes()):
        nk = len(seq) - k + 1
        assert np.array_equal(ridx.query_bitmap(f"g{g}", nm, 5, nk - 3).to_numpy(), bits[off + 5: off + nk - 3])
        off += nk


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_gpus_2_spawns_two_ranks_with_disjoint_work(scaling):
    """`python bench.py --gpus 2` with WORLD_SIZE unset must become two ranks by itself (torch.distributed.run) and say
    so on its line: n_gpus == 2 == ranks_observed.  On a one-GPU box both ranks share the device and talk over gloo
    (PG_BENCH_ONE_DEVICE / PG_BENCH_BACKEND: the numbers mean nothing, the code path is the 2-GPU one).  Weak: each rank
    its own contigs of a 2x longer pangenome.  Strong: ONE pangenome cut into pieces of homology classes whose planned
    positions add up to it exactly — the ranks' shares are disjoint and cover it."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PG_BENCH_ONE_DEVICE="1", PG_BENCH_BACKEND="gloo", PG_MIN_PIECE="100000")  # (2-Mb contigs are cut)
    for v in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(v, None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--genomes", "4",
           "--genome-mb", "10", "--scaling", scaling, "--no-sharded-leg", "--no-cpu-baseline"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_observed"] == 2 and d["scaling"] == scaling
    k, L, G = 21, 10_000_000, 4
    per_contig = L // 5 - k + 1
    if scaling == "weak":
        assert d["config"]["positions_per_step_per_gpu"] == G * 5 * per_contig
        assert abs(d["value"] - 2 * G * 5 * per_contig * d["steps"] / (d["ms_per_step"] * 1e-3 * d["steps"])) < 1e-3 * d["value"]
    else:
        total = G * 5 * per_contig
        assert abs(d["value"] - total / (d["ms_per_step"] * 1e-3)) < 1e-3 * d["value"]
        loads = [int(x) for x in d["config"]["parallelism"].split("[")[1].split("]")[0].split(",")]
        assert len(loads) == 2 and sum(loads) == total and max(loads) <= 1.1 * total / 2


def test_bench_gpus_8_dress_rehearsal_on_one_device():
    """`python bench.py --gpus 8 --scaling strong` on a configs[2]-shaped pangenome (27 genomes x 5 chromosomes, a tenth of
    the length) with all 8 ranks on the one device over gloo (PG_BENCH_ONE_DEVICE / PG_BENCH_BACKEND: the numbers mean
    nothing, the code path is the 8-GPU one the driver will launch): eight ranks answer, their pieces of homology classes are
    disjoint and cover the pangenome exactly, and the fullest rank carries at most 15 % more than the mean."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PG_BENCH_ONE_DEVICE="1", PG_BENCH_BACKEND="gloo", PG_MIN_PIECE="100000")  # (a tenth of the default 2^20)
    for v in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(v, None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--genomes", "27",
           "--genome-mb", "13.5", "--scaling", "strong", "--no-sharded-leg", "--no-cpu-baseline"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["ranks_observed"] == 8 and d["scaling"] == "strong"
    total = 27 * 5 * (13_500_000 // 5 - 21 + 1)
    loads = [int(x) for x in d["config"]["parallelism"].split("[")[1].split("]")[0].split(",")]
    assert len(loads) == 8 and sum(loads) == total and min(loads) > 0
    assert max(loads) <= 1.15 * total / 8, loads
    assert abs(d["value"] - total / (d["ms_per_step"] * 1e-3)) < 1e-3 * d["value"]
