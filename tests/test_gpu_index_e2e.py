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


@pytest.mark.parametrize("name", ["n2_k21", "n9_k21", "n40_k31", "n65_k21"])
def test_index_run_writes_reference_identical_tree(name, tmp_path):
    from panagram_amd import index as pidx
    fx = H.load_case(name)
    n, k = int(fx["ngenomes"]), int(fx["k"])
    anchors = [f"g{g}" for g in fx["anchors"]]
    s = _write_case(tmp_path, fx)
    out = tmp_path / "idx"
    idx = pidx.Index(str(s), prefix=str(out), k=k, anchor_genomes=anchors, export_kmc=True)
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


def test_run_anchor_cli_matches_reference_binary(tmp_path):
    """same argv + file contract as cpp/run_anchor, fed the same KMC1 files"""
    from panagram_amd.__main__ import main
    fx = H.load_case("n33_k16")
    n, k = int(fx["ngenomes"]), int(fx["k"])
    root = tmp_path / "root"
    (root / "kmc").mkdir(parents=True)
    for i, (keys, masks) in enumerate(H.case_dbs(fx)):
        po.write_kmc1(str(root / "kmc" / f"bitvec{i}"), keys, masks, k, lut_prefix_len=4)
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
