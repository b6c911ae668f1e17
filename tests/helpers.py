"""Shared helpers for the parity tests (oracle side only; never imported by the product)."""
import glob
import hashlib
import os

import numpy as np

from oracle import pyoracle as po

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _anchor_fixtures():
    return [f for f in glob.glob(os.path.join(GOLDEN, "*.npz")) if not os.path.basename(f).startswith(("kmc2_", "kmc1_"))]


def payload_cases():
    return sorted(os.path.basename(f)[:-4] for f in _anchor_fixtures() if bool(np.load(f)["store_payload"]))


def seed_cases():
    return sorted(os.path.basename(f)[:-4] for f in _anchor_fixtures() if not bool(np.load(f)["store_payload"]))


def kmc2_cases():
    """KMC2-layout database images accepted by the reference binary (tests/golden/make_golden.py)"""
    return sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(GOLDEN, "kmc2_*.npz")))


def load_case(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def case_dbs(fx):
    n = int(fx["ngenomes"])
    return [(fx[f"db{i}_keys"], fx[f"db{i}_masks"]) for i in range((n + 31) // 32)]


def regen_seed_case(fx):
    """Re-create the FASTA texts of a seed-only fixture exactly as make_golden.py did."""
    n, k = int(fx["ngenomes"]), int(fx["k"])
    gen = po.synth_genomes(n, [int(x) for x in fx["contig_lens"]], float(fx["d"]), int(fx["seed"]))
    wrap = [int(x) for x in fx["wrap"]]
    fastas, genomes = [], []
    for g, contigs in enumerate(gen):
        seqs = [po.codes_to_ascii(c) for c in contigs]
        genomes.append(seqs)
        names = [f"chr{ci + 1}" for ci in range(len(seqs))]
        fastas.append(po.fasta_text(names, seqs, wrap[g % len(wrap)]))
    return genomes, fastas


def sha(b) -> str:
    return hashlib.sha256(bytes(b)).hexdigest()


def bins_text(ngenomes, per_contig_bins, per_contig_binlen):
    """bitsum.bins.tsv exactly as cpp/anchor.cpp:59-63,184-188 writes it."""
    out = ["chr\tstart" + "".join(f"\t{i}" for i in range(ngenomes + 1)) + "\n"]
    for ci, (bins, binlen) in enumerate(zip(per_contig_bins, per_contig_binlen)):
        for b, row in enumerate(bins):
            out.append(f"{ci}\t{b * binlen}" + "".join(f"\t{int(c)}" for c in row) + "\n")
    return "".join(out)
