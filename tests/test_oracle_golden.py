"""CPU: the oracle restatement vs golden vectors produced by the reference binary."""
import numpy as np
import pytest

from oracle import pyoracle as po
from tests import helpers as H


@pytest.mark.parametrize("name", H.payload_cases())
def test_oracle_matches_reference_outputs(name):
    fx = H.load_case(name)
    n, k = int(fx["ngenomes"]), int(fx["k"])
    dbs = H.case_dbs(fx)
    for g in fx["anchors"]:
        r = po.anchor_fasta(dbs, fx[f"fasta_{g}"].tobytes(), k, n, int(fx["min_count"]), int(fx["max_count"]))
        assert r["bitmap1"] == fx[f"a{g}_bitmap1"].tobytes()
        assert r["bitmap100"] == fx[f"a{g}_bitmap100"].tobytes()
        assert r["bins_tsv"].encode() == fx[f"a{g}_bitsum.bins.tsv"].tobytes()
        assert r["chrs_tsv"].encode() == fx[f"a{g}_chrs.tsv"].tobytes()


@pytest.mark.parametrize("name", H.payload_cases())
def test_set_construction_reproduces_fixture_dbs(name):
    fx = H.load_case(name)
    n, k = int(fx["ngenomes"]), int(fx["k"])
    genomes = [[s for _, s in po.parse_fasta_cpp(fx[f"fasta_{g}"].tobytes())] for g in range(n)]
    dbs = po.build_bitvec_dbs(genomes, k)
    for (keys, masks), (fk, fm) in zip(dbs, H.case_dbs(fx)):
        assert np.array_equal(keys, fk) and np.array_equal(masks, fm)


def test_kmc1_roundtrip(tmp_path):
    fx = H.load_case("n40_k31")
    for i, (keys, masks) in enumerate(H.case_dbs(fx)):
        p = str(tmp_path / f"bitvec{i}")
        po.write_kmc1(p, keys, masks, 31, lut_prefix_len=7)
        db = po.read_kmc1(p)
        assert db["k"] == 31 and np.array_equal(db["keys"], keys) and np.array_equal(db["counters"], masks)


def test_canonical_edge_cases():
    # palindrome (even k): fwd == revcomp
    keys, valid = po.canonical_kmers(b"ACGT", 4)
    assert valid.all() and keys[0] == 0b00011011
    # all-T canonicalises to all-A = 0, so ~0 is never a key (EMPTY sentinel of the GPU table)
    keys, _ = po.canonical_kmers(b"T" * 32, 32)
    assert keys[0] == 0
    # non-ACGT byte invalidates every window that covers it; lower case == upper case
    keys_u, valid_u = po.canonical_kmers(b"ACGTNACGTA", 3)
    assert list(valid_u) == [True, True, False, False, False, True, True, True]
    keys_l, _ = po.canonical_kmers(b"acgtnacgta", 3)
    assert np.array_equal(keys_l[valid_u], keys_u[valid_u])
    # shorter than k
    assert len(po.canonical_kmers(b"ACG", 5)[0]) == 0


def test_row_byte_counts():
    assert po.row_byte_counts(1) == [1] and po.row_byte_counts(9) == [2] and po.row_byte_counts(32) == [4]
    assert po.row_byte_counts(33) == [4, 1] and po.row_byte_counts(40) == [4, 1]
    assert po.row_byte_counts(64) == [4, 4] and po.row_byte_counts(65) == [4, 4, 1]


@pytest.mark.parametrize("name", H.kmc2_cases())
def test_kmc2_layout_restatement_reads_the_reference_accepted_files(name):
    """The KMC2-layout images (kmc_version 0x200) the reference binary accepted hold exactly the k-mers of their
    KMC1 twin; the library's header parser (host only) agrees on k."""
    from panagram_amd import engine
    fx = H.load_case(name)
    ref = H.load_case(str(fx["ref_case"]))
    for i, (keys, masks) in enumerate(H.case_dbs(ref)):
        d = po.parse_kmc2(fx[f"db{i}_pre"].tobytes(), fx[f"db{i}_suf"].tobytes())
        assert d["k"] == int(ref["k"]) and d["signature_len"] == int(fx["signature_len"]) and d["nbins"] == int(fx["nbins"])
        assert np.array_equal(d["keys"], keys) and np.array_equal(d["counters"], masks)
        assert engine.kmc_kmer_length(fx[f"db{i}_pre"]) == int(ref["k"])
