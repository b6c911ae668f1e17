"""CPU: the C restatement (the CPU baseline) vs golden vectors from the reference binary."""
import numpy as np
import pytest

from oracle import coracle, pyoracle as po
from tests import helpers as H


@pytest.mark.parametrize("name", H.payload_cases())
def test_c_oracle_matches_reference_outputs(name, tmp_path):
    fx = H.load_case(name)
    n, k = int(fx["ngenomes"]), int(fx["k"])
    dbs = []
    for i, (keys, masks) in enumerate(H.case_dbs(fx)):
        if i % 2 == 0:  # exercise both constructors
            dbs.append(coracle.OracleDB.from_arrays(keys, masks, k, None, int(fx["min_count"]), int(fx["max_count"])))
        else:
            p = str(tmp_path / f"bitvec{i}")
            po.write_kmc1(p, keys, masks, k, min_count=int(fx["min_count"]), max_count=int(fx["max_count"]))
            dbs.append(coracle.OracleDB.from_files(p))
    for g in fx["anchors"]:
        b1, b100, bins_rows = [], [], []
        for ci, (_, seq) in enumerate(po.parse_fasta_cpp(fx[f"fasta_{g}"].tobytes())):
            rows, rows100, bins, starts = coracle.write_bits(dbs, n, seq, k)
            b1.append(rows.tobytes())
            b100.append(rows100.tobytes())
            for s, r in zip(starts, bins):
                bins_rows.append(f"{ci}\t{int(s)}" + "".join(f"\t{int(c)}" for c in r) + "\n")
        assert b"".join(b1) == fx[f"a{g}_bitmap1"].tobytes()
        assert b"".join(b100) == fx[f"a{g}_bitmap100"].tobytes()
        hdr = "chr\tstart" + "".join(f"\t{i}" for i in range(n + 1)) + "\n"
        assert (hdr + "".join(bins_rows)).encode() == fx[f"a{g}_bitsum.bins.tsv"].tobytes()


def test_c_oracle_config1_sha256():
    """config-1 shape (2 x 1 Mb, k=21): sha256 of the payload the reference binary wrote."""
    fx = H.load_case("c1_2x1mb_k21")
    n, k = int(fx["ngenomes"]), int(fx["k"])
    genomes, fastas = H.regen_seed_case(fx)
    dbs_np = po.build_bitvec_dbs(genomes, k)
    dbs = [coracle.OracleDB.from_arrays(kk, mm, k) for kk, mm in dbs_np]
    for g in fx["anchors"]:
        rows, rows100, bins, starts = coracle.write_bits(dbs, n, genomes[g][0], k)
        assert H.sha(rows.tobytes()) == str(fx[f"a{g}_sha_1"])
        assert H.sha(rows100.tobytes()) == str(fx[f"a{g}_sha_100"])
