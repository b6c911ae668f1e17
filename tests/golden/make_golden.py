#!/usr/bin/env python3
"""Generate golden vectors by RUNNING THE REFERENCE'S OWN BINARY.

Runs only in the build container (needs /root/reference/cpp/run_anchor, the
prebuilt `cpp/anchor.cpp` + KMC API + htslib).  Nothing of the reference travels:
the fixtures hold only inputs (FASTA text, the k-mer DB as key/counter arrays or
a seed) and the outputs the reference produced (decompressed bitmap payloads or
their sha256, TSV texts, .gzi bytes).

    python tests/golden/make_golden.py            # regenerate all fixtures

Procedure: seed -> synthetic FASTAs -> k-mer sets (canonical,
-ci1, one-hot OR per 32-genome group) -> KMC1 files (oracle.pyoracle.write_kmc1)
-> `run_anchor N root name fasta ...` -> gunzip -> fixture.
"""
import gzip
import hashlib
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import pyoracle as po  # noqa: E402

RUN_ANCHOR = "/root/reference/cpp/run_anchor"


def decorate(seq: bytes, rng, n_runs=True, lower=True) -> bytes:
    """Add an N-run, a lowercase stretch and an IUPAC byte to a contig."""
    s = bytearray(seq)
    L = len(s)
    if n_runs and L > 400:
        a = int(rng.integers(50, L - 100))
        ln = int(rng.integers(1, 40))
        s[a:a + ln] = b"N" * ln
        b = int(rng.integers(0, L - 1))
        s[b] = ord("R")
    if lower and L > 400:
        a = int(rng.integers(0, L - 250))
        s[a:a + 200] = bytes(s[a:a + 200]).lower()
    return bytes(s)


def make_case(name, ngenomes, k, contig_lens, d, seed, anchors, wrap=(80, 70, 60),
              messy=True, store_payload=True, lut=None, min_count=1, max_count=0xFFFFFFFF):
    rng = np.random.default_rng(seed + 777)
    gen = po.synth_genomes(ngenomes, contig_lens, d, seed)
    genomes = []
    fastas = []
    for g, contigs in enumerate(gen):
        seqs = [po.codes_to_ascii(c) for c in contigs]
        if messy and g % 2 == 1:
            seqs = [decorate(s, rng) for s in seqs]
        genomes.append(seqs)
        names = [f"chr{ci + 1}" + (" some description" if (messy and ci == 0) else "")
                 for ci in range(len(seqs))]
        fastas.append(po.fasta_text(names, seqs, wrap[g % len(wrap)]))
    dbs = po.build_bitvec_dbs(genomes, k)

    root = tempfile.mkdtemp(prefix="golden_")
    try:
        os.makedirs(os.path.join(root, "kmc"))
        for i, (keys, masks) in enumerate(dbs):
            po.write_kmc1(os.path.join(root, "kmc", f"bitvec{i}"), keys, masks, k,
                          lut_prefix_len=lut, min_count=min_count, max_count=max_count)
        args = [RUN_ANCHOR, str(ngenomes), root]
        for g in anchors:
            nm = f"g{g}"
            os.makedirs(os.path.join(root, "anchor", nm))
            fa = os.path.join(root, nm + ".fa")
            with open(fa, "wb") as f:
                f.write(fastas[g])
            args += [nm, fa]
        env = dict(os.environ, OMP_NUM_THREADS="1")
        subprocess.run(args, check=True, stdout=subprocess.DEVNULL, env=env)

        fx = dict(ngenomes=ngenomes, k=k, anchors=np.array(anchors), seed=seed, d=d,
                  contig_lens=np.array(contig_lens), min_count=min_count, max_count=max_count,
                  messy=messy, store_payload=store_payload)
        if store_payload:
            for g in range(ngenomes):
                fx[f"fasta_{g}"] = np.frombuffer(fastas[g], np.uint8)
            for i, (keys, masks) in enumerate(dbs):
                fx[f"db{i}_keys"] = keys
                fx[f"db{i}_masks"] = masks
        else:
            fx["wrap"] = np.array(wrap)
        for g in anchors:
            adir = os.path.join(root, "anchor", f"g{g}")
            for step in (1, 100):
                with gzip.open(os.path.join(adir, f"bitmap.{step}.gz"), "rb") as f:
                    payload = f.read()
                fx[f"a{g}_sha_{step}"] = hashlib.sha256(payload).hexdigest()
                fx[f"a{g}_len_{step}"] = len(payload)
                if store_payload:
                    fx[f"a{g}_bitmap{step}"] = np.frombuffer(payload, np.uint8)
                    with open(os.path.join(adir, f"bitmap.{step}.gzi"), "rb") as f:
                        fx[f"a{g}_gzi{step}"] = np.frombuffer(f.read(), np.uint8)
            for t in ("bitsum.bins.tsv", "chrs.tsv"):
                with open(os.path.join(adir, t), "rb") as f:
                    fx[f"a{g}_{t}"] = np.frombuffer(f.read(), np.uint8)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **fx)
        print("wrote", name, {k_: v for k_, v in fx.items() if k_.startswith("a") and "len" in k_})
    finally:
        shutil.rmtree(root)


ONLY = set(sys.argv[1:])  # optional: fixture names to (re)generate; default all


def make_kmc2_case(name, ref_case, lut, sig_len=7, nbins=13, guard=False):
    """KMC2-LAYOUT databases (kmc_version 0x200, what `kmc` itself writes) holding the k-mers of an existing
    fixture, written by oracle.pyoracle.write_kmc2 and READ BY THE REFERENCE BINARY: its outputs over them must be
    the ones it gave over the KMC1 files of `ref_case` (stored there).  Only then are the file images committed —
    they are what the product's KMC2 reader is tested on."""
    fx = np.load(os.path.join(HERE, ref_case + ".npz"))
    n, k = int(fx["ngenomes"]), int(fx["k"])
    ndbs = (n + 31) // 32
    root = tempfile.mkdtemp(prefix="golden_kmc2_")
    try:
        os.makedirs(os.path.join(root, "kmc"))
        out = dict(ref_case=ref_case, ngenomes=n, k=k, lut_prefix_len=lut, signature_len=sig_len, nbins=nbins, guard=guard)
        for i in range(ndbs):
            pre = os.path.join(root, "kmc", f"bitvec{i}")
            po.write_kmc2(pre, fx[f"db{i}_keys"], fx[f"db{i}_masks"], k, lut, sig_len=sig_len, nbins=nbins, guard=guard)
            out[f"db{i}_pre"] = np.fromfile(pre + ".kmc_pre", np.uint8)
            out[f"db{i}_suf"] = np.fromfile(pre + ".kmc_suf", np.uint8)
        args = [RUN_ANCHOR, str(n), root]
        for g in fx["anchors"]:
            os.makedirs(os.path.join(root, "anchor", f"g{g}"))
            fa = os.path.join(root, f"g{g}.fa")
            with open(fa, "wb") as f:
                f.write(fx[f"fasta_{g}"].tobytes())
            args += [f"g{g}", fa]
        subprocess.run(args, check=True, stdout=subprocess.DEVNULL, env=dict(os.environ, OMP_NUM_THREADS="1"))
        for g in fx["anchors"]:
            adir = os.path.join(root, "anchor", f"g{g}")
            for step in (1, 100):
                with gzip.open(os.path.join(adir, f"bitmap.{step}.gz"), "rb") as f:
                    assert f.read() == fx[f"a{g}_bitmap{step}"].tobytes(), "the reference reads our KMC2 files differently"
            with open(os.path.join(adir, "bitsum.bins.tsv"), "rb") as f:
                assert f.read() == fx[f"a{g}_bitsum.bins.tsv"].tobytes()
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print("wrote", name, "(reference binary's outputs over the KMC2 files == its outputs over the KMC1 files)")
    finally:
        shutil.rmtree(root)


def pin_product_kmc1_writer(name, cases):
    """The PRODUCT's KMC1 writer (panagram_amd.index.write_kmc1, behind `panagram_amd index --export_kmc`) pinned by
    the reference's own reader (CKMCFile::OpenForRA, cpp/anchor.cpp:26-31): the databases of existing fixtures are
    written with it — once with the prefix length the fixture was made with, once with the writer's own choice —
    `run_anchor` is run over them and must reproduce the fixture's golden outputs.  Only then are the sha256 of the
    files it wrote committed: tests/test_host_logic.py checks the writer still produces exactly those bytes."""
    from panagram_amd import index as pidx
    out = dict(cases=np.array(cases))
    for ref_case, lut in cases:
        fx = np.load(os.path.join(HERE, ref_case + ".npz"))
        n, k = int(fx["ngenomes"]), int(fx["k"])
        for tag, p in (("lut", int(lut)), ("auto", None)):
            root = tempfile.mkdtemp(prefix="golden_pw_")
            try:
                os.makedirs(os.path.join(root, "kmc"))
                for i in range((n + 31) // 32):
                    pre = os.path.join(root, "kmc", f"bitvec{i}")
                    pidx.write_kmc1(pre, fx[f"db{i}_keys"], fx[f"db{i}_masks"], k, lut_prefix_len=p)
                    for ext in ("kmc_pre", "kmc_suf"):
                        with open(pre + "." + ext, "rb") as f:
                            out[f"{ref_case}_{tag}_db{i}_{ext}_sha"] = hashlib.sha256(f.read()).hexdigest()
                args = [RUN_ANCHOR, str(n), root]
                for g in fx["anchors"]:
                    os.makedirs(os.path.join(root, "anchor", f"g{g}"))
                    fa = os.path.join(root, f"g{g}.fa")
                    with open(fa, "wb") as f:
                        f.write(fx[f"fasta_{g}"].tobytes())
                    args += [f"g{g}", fa]
                subprocess.run(args, check=True, stdout=subprocess.DEVNULL, env=dict(os.environ, OMP_NUM_THREADS="1"))
                for g in fx["anchors"]:
                    adir = os.path.join(root, "anchor", f"g{g}")
                    for step in (1, 100):
                        with gzip.open(os.path.join(adir, f"bitmap.{step}.gz"), "rb") as f:
                            assert f.read() == fx[f"a{g}_bitmap{step}"].tobytes(), \
                                f"{ref_case}/{tag}: the reference reads the product's KMC1 files differently"
                    for t in ("bitsum.bins.tsv", "chrs.tsv"):
                        with open(os.path.join(adir, t), "rb") as f:
                            assert f.read() == fx[f"a{g}_{t}"].tobytes(), (ref_case, tag, t)
            finally:
                shutil.rmtree(root)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, "(reference binary's outputs over the product-written KMC1 files == the golden outputs)")


def make_case_if(name, *a, **kw):
    if not ONLY or name in ONLY:
        make_case(name, *a, **kw)


def main():
    if not os.path.exists(RUN_ANCHOR):
        sys.exit("reference binary not found (run in the build container)")
    # N=2, k=21: N-run, lower case, IUPAC byte, description in header, ragged wrap
    make_case_if("n2_k21", 2, 21, [1700, 1300], 0.02, 11, [0, 1], lut=5)
    # nbytes=2
    make_case_if("n9_k21", 9, 21, [2500, 900], 0.03, 12, [0, 4, 8], lut=5)
    # two DBs, nbytes=5 -> rows [db0 b0..b3][db1 b0]
    make_case_if("n40_k31", 40, 31, [2500], 0.01, 13, [0, 39], lut=7)
    # even k (palindromic k-mers exist), nbytes=5 with N=33
    make_case_if("n33_k16", 33, 16, [3000, 400], 0.05, 14, [1, 32], lut=4)
    # k=32 (largest single-word k), N=3
    make_case_if("n3_k32", 3, 32, [2200], 0.02, 15, [0, 2], lut=8)
    # N=64: nbytes=8, both DBs give 4 bytes;  N=65: 3 DBs, rows 4+4+1
    make_case_if("n64_k31", 64, 31, [1500], 0.01, 16, [0, 63], lut=7)
    make_case_if("n65_k21", 65, 21, [1500, 700], 0.01, 17, [64, 5], lut=5)
    # min/max counter filter in the DB header (counters outside read as 0)
    make_case_if("n4_k21_minmax", 4, 21, [2000], 0.05, 18, [0, 3], lut=5, min_count=2, max_count=7)
    # small k, heavy collisions between strands / genomes
    make_case_if("n5_k9", 5, 9, [5000], 0.1, 19, [0, 2], lut=5)
    # config-1 shaped (2 x 1 Mb, k=21, d=0.01, seed 1234): seed-only fixture, sha256 of payloads
    make_case_if("c1_2x1mb_k21", 2, 21, [1000000], 0.01, 1234, [0, 1], wrap=(80,), messy=False,
              store_payload=False, lut=9)
    # config-5 shaped (8 genomes, k=21, one-byte rows): the genome-sharded mode's one-genome-per-GPU layout
    make_case_if("n8_k21", 8, 21, [3000, 1200], 0.02, 20, [0, 3, 7], lut=5)
    # KMC2-layout databases of two of the cases above, accepted by the reference's own KMC reader
    if not ONLY or "kmc2_n9_k21" in ONLY:
        make_kmc2_case("kmc2_n9_k21", "n9_k21", lut=5, sig_len=7, nbins=13)
    if not ONLY or "kmc2_n40_k31" in ONLY:
        make_kmc2_case("kmc2_n40_k31", "n40_k31", lut=7, sig_len=9, nbins=64, guard=True)
    # the product's own KMC1 writer, read by the reference binary (two row widths, one and two databases)
    if not ONLY or "kmc1_writer_pins" in ONLY:
        pin_product_kmc1_writer("kmc1_writer_pins", [("n9_k21", 5), ("n40_k31", 7), ("n3_k32", 8), ("n5_k9", 5)])
    # >= 100 bins of 200000: exercises binlen=200000 + tail bin + multi-block .gzi
    make_case_if("big_n3_k21", 3, 21, [20300000, 150000], 0.01, 4321, [1], wrap=(80,), messy=False,
              store_payload=False, lut=9)


if __name__ == "__main__":
    main()
