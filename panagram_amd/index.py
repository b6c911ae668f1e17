"""Host-side counterpart of the reference's write path (``panagram/index.py``), scoped to
the anchor hot path and driving the HIP kernels through the C-ABI.

Mirrors, with the same names / argument meaning / on-disk layout:

* ``Index``  — config + orchestration (``panagram/index.py:85-466``): ``samples.tsv``
  schema ``name fasta gff id anchor`` (``:269-295``), ``config.yaml`` keys (``:347-357``),
  ``kmc_bitvec_count`` / ``bitvec_prefixes`` / ``steps`` (``:391-405``), ``run()``.
  Where the reference launches Snakemake -> kmc -> kmc_tools -> one anchor job per genome,
  ``Index.run()`` builds the pan-kmer table on the GPU (or loads existing
  ``kmc/bitvec{i}.kmc_{pre,suf}``) and anchors every anchor genome in-process.
* ``Genome`` — ``run_anchor`` (``:1012-1097``), ``iter_fasta`` (``:922-930``),
  ``set_chrs`` offsets (``:596-604``) and the read side ``load_bgz_blocks`` /
  ``_query_bytes`` / ``query`` (``:793-845``) so that what we write can be read back the way
  ``panagram view`` reads it.

Output tree (identical to the reference's):
    <prefix>/config.yaml, samples.tsv
    <prefix>/anchor/<name>/bitmap.1.gz(.gzi) bitmap.100.gz(.gzi) bitsum.bins.tsv chrs.tsv
                           total_paircounts.csv

Out of scope here: GFF annotation, UMAPs, mash distances, the viewer.
"""
from __future__ import annotations

import dataclasses
import gzip
import logging
import os
import re
import struct
import zlib
from typing import Dict, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import pandas as pd
import yaml

from . import engine

logger = logging.getLogger(__name__)

NAME_REGEX = "[A-Za-z0-9_-]+"
BGZ_SUFFIX = "gz"
IDX_SUFFIX = "gzi"
ANCHOR_DIR = "anchor"
_WS = b" \t\r\n\x0b\x0c"


# ---------------------------------------------------------------------------
# FASTA
# ---------------------------------------------------------------------------
def read_fasta(path: str) -> Iterator[Tuple[str, bytes]]:
    """Yield ``(record id, sequence bytes)``.  Semantics of the reference's Python path
    (``Bio.SeqIO`` via ``iter_fasta``, index.py:922-930): id = header up to the first
    whitespace, sequence lines joined with all whitespace removed, ``.gz``/``.bgz`` read
    through gzip.  (The C++ variant keeps ``\\r`` bytes in the sequence, cpp/anchor.cpp:77-93;
    the two agree on well-formed FASTA.)"""
    opn = gzip.open if path.endswith((".gz", ".bgz")) else open
    with opn(path, "rb") as f:
        data = f.read()
    pos = 0 if data[:1] == b">" else data.find(b"\n>") + 1
    if pos == 0 and data[:1] != b">":
        return
    n = len(data)
    while pos < n:
        eol = data.find(b"\n", pos)
        if eol < 0:
            eol = n
        header = data[pos + 1:eol].strip()
        nxt = data.find(b"\n>", eol)
        end = n if nxt < 0 else nxt + 1
        seq = data[eol + 1:end].translate(None, _WS)
        name = header.split(None, 1)[0].decode("latin-1") if header else ""
        yield name, seq
        pos = end


FASTQ_SUFFIXES = (".fastq", ".fastq.gz", ".fq", ".fq.gz")  # workflow/Snakefile:88-89


def _read_fasta_image(path: str) -> np.ndarray:
    """the bytes of a FASTA file (``.gz`` / ``.bgz`` inflated) for ``SeqSet.from_fasta``"""
    if path.endswith((".gz", ".bgz")):
        import gzip
        with gzip.open(path, "rb") as f:
            return np.frombuffer(f.read(), dtype=np.uint8)
    return np.fromfile(path, dtype=np.uint8)


def is_fastq(path) -> bool:
    return isinstance(path, str) and path.endswith(FASTQ_SUFFIXES)


def read_fastq_joined(path: str) -> bytes:
    """The reads of a 4-line-record FASTQ joined by ``N``: no k-mer window spans two reads, which is
    how kmc -fq counts them."""
    opn = gzip.open if path.endswith(".gz") else open
    with opn(path, "rb") as f:
        lines = f.read().split(b"\n")
    return b"N".join(ln.rstrip(b"\r") for ln in lines[1::4])


# ---------------------------------------------------------------------------
# KMC1 database files (format: SURVEY.md Appendix A) — writer, so that tables built on
# the GPU can be handed to the reference (`kmc/bitvec{i}.kmc_pre|.kmc_suf`)
# ---------------------------------------------------------------------------
def write_kmc1(prefix: str, keys: np.ndarray, counters: np.ndarray, k: int,
               lut_prefix_len: Optional[int] = None) -> None:
    keys = np.ascontiguousarray(keys, np.uint64)
    counters = np.ascontiguousarray(counters, np.uint32)
    order = np.argsort(keys, kind="stable")
    keys, counters = keys[order], counters[order]
    if lut_prefix_len is None:
        cands = [p for p in range(1, min(k, 13)) if (k - p) % 4 == 0]
        lut_prefix_len = cands[0]
        for p in cands:
            if 4 ** p <= max(len(keys), 1) * 4:
                lut_prefix_len = p
    p = lut_prefix_len
    if (k - p) % 4:
        raise ValueError("(k - lut_prefix_len) must be a multiple of 4")
    suf_bytes = (k - p) // 4
    pref = (keys >> np.uint64(2 * (k - p))).astype(np.int64)
    lut = np.searchsorted(pref, np.arange(4 ** p, dtype=np.int64), side="left").astype(np.uint64)
    with open(prefix + ".kmc_pre", "wb") as f:
        f.write(b"KMCP")
        f.write(lut.tobytes())
        f.write(struct.pack("<IIIIIIQB3x24xI", k, 0, 4, p, 1, 0xFFFFFFFF, len(keys), 0, 0))
        f.write(struct.pack("<I", 64))
        f.write(b"KMCP")
    rec = np.zeros((len(keys), suf_bytes + 4), np.uint8)
    for b in range(suf_bytes):
        rec[:, b] = ((keys >> np.uint64(8 * (suf_bytes - 1 - b))) & np.uint64(0xFF)).astype(np.uint8)
    rec[:, suf_bytes:] = counters.view(np.uint8).reshape(-1, 4)
    with open(prefix + ".kmc_suf", "wb") as f:
        f.write(b"KMCS")
        f.write(rec.tobytes())
        f.write(b"KMCS")


# ---------------------------------------------------------------------------
# BGZF read side (Bio.bgzf is not a dependency here)
# ---------------------------------------------------------------------------
def load_bgz_blocks(fname: str) -> np.ndarray:
    """index.py:793-799: (rstart, dstart) per block, block 0 implicit."""
    raw = np.fromfile(fname, "<u8")
    n = int(raw[0])
    blocks = np.zeros((n + 1, 2), np.int64)
    blocks[1:] = raw[1:1 + 2 * n].reshape(n, 2)
    return blocks


def bgzf_read(path: str, blocks: np.ndarray, byte_start: int, length: int) -> bytes:
    """Random access as in index.py:827-845: locate the block through the .gzi, then inflate."""
    blk = int(np.searchsorted(blocks[:, 1], byte_start, side="right") - 1)
    out = bytearray()
    skip = byte_start - int(blocks[blk, 1])
    with open(path, "rb") as f:
        f.seek(int(blocks[blk, 0]))
        while len(out) < length:
            hdr = f.read(18)
            if len(hdr) < 18:
                break
            bsize = struct.unpack("<H", hdr[16:18])[0] + 1
            body = f.read(bsize - 18)
            data = zlib.decompress(body[:-8], -15)
            if not data:
                break  # EOF block
            out += data[skip:]
            skip = 0
    return bytes(out[:length])


# ---------------------------------------------------------------------------
# config dataclasses (same field names and defaults as index.py:63-138)
# ---------------------------------------------------------------------------
@dataclasses.dataclass
class KMC:
    memory: int = 8
    threads: int = 1
    use_existing: bool = False


@dataclasses.dataclass
class UMAP:
    neighbors: int = 4
    dist: float = 0
    eps: float = 1
    samples: int = 1
    bin_size: int = 100000


@dataclasses.dataclass
class Index:
    """Anchor k-mer bitvectors to reference FASTA files to create the pan-kmer bitmap."""

    input: str
    mode: Optional[str] = None
    prefix: Optional[str] = None
    k: int = 21
    cores: int = 1
    lowres_step: int = 100
    max_bin_kbp: int = 200
    min_bin_count: int = 100
    max_view_chrs: int = 50
    gff_gene_types: List[str] = dataclasses.field(default_factory=lambda: ["gene"])
    gff_anno_types: Optional[List[str]] = None
    gff_name: str = "Name"
    anchor_genomes: Optional[List[str]] = None
    prepare: bool = False
    kmc: KMC = dataclasses.field(default_factory=KMC)
    genome_umap: UMAP = dataclasses.field(default_factory=UMAP)
    chrom_umap: UMAP = dataclasses.field(default_factory=UMAP)
    use_existing: int = 1
    threads: int = 1
    memory: int = 1
    # -- not part of the reference's schema: which GPU this process drives, and whether to
    #    also export kmc/bitvec{i} in KMC1 layout for the reference to read
    device: int = 0
    export_kmc: bool = False
    rank: int = dataclasses.field(default_factory=lambda: int(os.environ.get("RANK", "0")))
    world: int = dataclasses.field(default_factory=lambda: int(os.environ.get("WORLD_SIZE", "1")))

    _EXTRA = ("device", "export_kmc", "rank", "world")

    def __post_init__(self):
        if not (self.mode is None or self.mode in {"r", "w"}):
            raise ValueError(f"Invalid mode '{self.mode}', must be 'r' or 'w'")
        if self.lowres_step != 100 or self.max_bin_kbp != 200 or self.min_bin_count != 100:
            # the kernels fuse the reference's defaults (cpp/anchor.cpp hard-codes them too)
            raise ValueError("lowres_step / max_bin_kbp / min_bin_count other than 100 / 200 / 100 "
                             "are not supported by the GPU path")
        self.write_mode = os.path.isfile(self.input) if self.mode is None else self.mode == "w"
        if self.write_mode:
            if os.path.isdir(self.input):
                self.prefix = self.input
                if not (os.path.isfile(self.config_fname) and os.path.isfile(self.samples_fname)):
                    raise ValueError("Index write directory not initialized")
                self.input = self.samples_fname
                self.load_config()
            elif os.path.isfile(self.input):
                if self.prefix is None:
                    self.prefix = os.path.dirname(self.input)
                if len(self.prefix) == 0:
                    self.prefix = "."
                os.makedirs(self.prefix, exist_ok=True)
                self.init_config()
            else:
                raise ValueError("Index input must be sample TSV or initialized directory")
        else:
            if not os.path.isdir(self.input):
                raise ValueError("Index input must be directory mode='r'")
            self.prefix = self.input
            self.load_config()

        self.samples = pd.read_table(self.samples_fname)
        missing = {"name", "id"}.difference(self.samples.columns)
        if missing:
            raise ValueError(f"{self.samples_fname} is missing required column(s): {', '.join(sorted(missing))}. "
                             "This directory does not look like a prepared Panagram index.")
        self.samples = self.samples.set_index("name")
        self.ngenomes = len(self.samples)
        self.genomes: Dict[str, Genome] = {}
        for name, row in self.samples.iterrows():
            self.genomes[name] = Genome(self, int(row["id"]), name, row["fasta"], row.get("gff"),
                                        bool(row["anchor"]), write=self.write_mode)
        if self.anchor_genomes is None:
            self.anchor_genomes = [n for n, g in self.genomes.items() if g.anchored]
        self._ctx = None
        self._table = None
        self._seqsets: Dict[str, engine.SeqSet] = {}

    # ---- naming (index.py:155-165, 359-405) ----
    @property
    def params(self):
        d = dataclasses.asdict(self)
        for e in self._EXTRA:
            d.pop(e, None)
        return d

    @property
    def config_fname(self):
        return os.path.join(self.prefix, "config.yaml")

    @property
    def samples_fname(self):
        return os.path.join(self.prefix, "samples.tsv")

    def get_subdir(self, name):
        return os.path.join(self.prefix, name)

    def kmc_prefix(self, *names):
        return os.path.join(self.get_subdir("kmc"), ".".join(names))

    @property
    def genome_names(self):
        return self.samples.index

    @property
    def kmc_bitvec_count(self):
        return int(np.ceil(len(self.samples) / 32.0))

    @property
    def bitvec_prefixes(self):
        return [self.kmc_prefix(f"bitvec{i}") for i in range(self.kmc_bitvec_count)]

    @property
    def steps(self):
        return (1, self.lowres_step)

    def __getitem__(self, genome):
        return self.genomes[genome]

    # ---- config files (index.py:269-295, 347-357) ----
    def init_config(self):
        samples = pd.read_table(self.input)
        if "name" not in samples.columns or "fasta" not in samples.columns:
            raise ValueError("Input samples must contain 'name' and 'fasta' column headers")
        if "gff" not in samples:
            samples["gff"] = pd.NA
        invalid = ~samples["name"].astype(str).str.fullmatch(NAME_REGEX)
        if np.any(invalid):
            bad = "', '".join(samples["name"][invalid])
            raise ValueError(f"Invalid genome names: '{bad}'\nMust match r'{NAME_REGEX}'.")
        keep = ["name", "fasta", "gff"] + (["anchor"] if "anchor" in samples else [])
        samples = samples[keep].set_index("name").dropna(how="all")
        samples["id"] = np.arange(len(samples), dtype=int)
        if self.anchor_genomes is None:
            if "anchor" in samples:
                self.anchor_genomes = list(samples.index[samples["anchor"].astype(bool)])
            else:
                self.anchor_genomes = list(samples["fasta"].dropna().index)
        samples["anchor"] = samples.index.isin(self.anchor_genomes)
        # (several ranks of one multi-GPU job write the same bytes: write-then-rename keeps readers whole)
        tmp = f"{self.samples_fname}.{os.getpid()}.tmp"
        samples[["fasta", "gff", "id", "anchor"]].to_csv(tmp, sep="\t")
        os.replace(tmp, self.samples_fname)
        self.write_config()

    def write_config(self, exclude=("prefix",)):
        prms = self.params
        for p in exclude:
            del prms[p]
        tmp = f"{self.config_fname}.{os.getpid()}.tmp"
        with open(tmp, "w") as conf_out:
            yaml.dump(prms, conf_out)
        os.replace(tmp, self.config_fname)

    def load_config(self):
        with open(self.config_fname) as f:
            vals = yaml.load(f, yaml.SafeLoader) or {}
        for key, val in vals.items():
            cur = getattr(self, key, None)
            if dataclasses.is_dataclass(cur) and isinstance(val, dict):
                for k2, v2 in val.items():
                    setattr(cur, k2, v2)
            elif key not in ("input", "mode", "prefix"):
                setattr(self, key, val)

    # ---- the table: replaces rules kmc_count / opdefs / kmc_bitvec (workflow/Snakefile:54-110) ----
    @property
    def context(self) -> engine.Context:
        if self._ctx is None:
            self._ctx = engine.Context(self.device)
        return self._ctx

    def build_table(self, keep: Optional[Sequence[str]] = None) -> engine.PanTable:
        """``keep``: the genomes whose packed sequences stay resident for the anchor step (default: all anchors)"""
        keep = set(self.anchor_genomes if keep is None else keep)
        if self._table is not None:
            return self._table
        have = all(os.path.exists(p + ".kmc_pre") and os.path.exists(p + ".kmc_suf") for p in self.bitvec_prefixes)
        if self.kmc.use_existing and have:
            tbl = engine.PanTable(self.context, self.k, self.ngenomes)
            for i, p in enumerate(self.bitvec_prefixes):
                with open(p + ".kmc_pre", "rb") as f:
                    pre = f.read()
                with open(p + ".kmc_suf", "rb") as f:
                    suf = f.read()
                tbl.load_kmc1(i, pre, suf)
            logger.info("KMC Database Loaded")
        else:
            # every input is parsed and packed once on the GPU (0.375 byte per base) and stays
            # resident through the build; a sketch of the distinct k-mers over all of them sizes the
            # table (and settles its minimizer length) once: no re-hash while it grows, no second
            # copy of the table in HBM.  Anchors keep their sequences for the anchor step.
            inputs = []
            sketch = engine.KmerSketch(self.context, self.k)
            # the FASTA files are read a few ahead by host threads while the GPU parses and sketches
            from concurrent.futures import ThreadPoolExecutor
            todo = [n for n, g in self.genomes.items() if not pd.isna(g.fasta) and not is_fastq(g.fasta) and n not in self._seqsets]
            nread = max(2, min(6, engine.usable_cpus() // 2))
            reader = ThreadPoolExecutor(max_workers=nread)
            ahead = {n: reader.submit(_read_fasta_image, self.genomes[n].fasta) for n in todo[:nread + 2]}
            nxt = nread + 2
            for name, g in self.genomes.items():
                if pd.isna(g.fasta):
                    continue
                if name in ahead:
                    self._seqsets[name] = engine.SeqSet.from_fasta(self.context, ahead.pop(name).result())
                    if nxt < len(todo):
                        ahead[todo[nxt]] = reader.submit(_read_fasta_image, self.genomes[todo[nxt]].fasta)
                        nxt += 1
                if is_fastq(g.fasta):
                    # read sets: kmc -ci2 -fq (workflow/Snakefile:88-89) — k-mers seen once are dropped
                    # (the sketch counts them too: the table is sized from above)
                    if name in self.anchor_genomes:
                        raise ValueError(f"{name}: a FASTQ sample cannot be an anchor genome")
                    ss = engine.SeqSet.from_host(self.context, [read_fastq_joined(g.fasta)])
                    inputs.append((name, g, ss, 2))
                else:
                    inputs.append((name, g, self.seqset_for(name), 1))
                sketch.add(inputs[-1][2])
            reader.shutdown()
            expected = sketch.estimate()
            sketch.close()
            self.context.trim()  # (the FASTA text buffer the parser kept for the next file)
            tbl = engine.PanTable(self.context, self.k, self.ngenomes, expected_keys=expected + expected // 32 + 1024)
            for name, g, ss, min_count in inputs:
                tbl.insert_seqset(g.id, ss, min_count=min_count)
                if min_count > 1:
                    ss.close()
                elif name not in keep:
                    self.drop_seqset(name)
            logger.info("k-mer table built on GPU (sketch: %d distinct k-mers): %s", expected, tbl.stats())
            if self.export_kmc:
                os.makedirs(self.get_subdir("kmc"), exist_ok=True)
                for i, p in enumerate(self.bitvec_prefixes):
                    keys, vals = tbl.export(i)
                    write_kmc1(p, keys, vals, self.k)
        self._table = tbl
        return tbl

    def seqset_for(self, name: str) -> engine.SeqSet:
        """The genome's FASTA, parsed and 2-bit packed in HBM (0.375 byte per base), cached."""
        ss = self._seqsets.get(name)
        if ss is None:
            ss = self._seqsets[name] = engine.SeqSet.from_fasta(self.context, self.genomes[name].fasta)
        return ss

    def drop_seqset(self, name: str) -> None:
        ss = self._seqsets.pop(name, None)
        if ss is not None:
            ss.close()

    # ---- panagram index command (index.py:172-191) ----
    def run(self):
        """Table build, then the anchors in batches: the anchor genomes of a batch share ONE
        co-scheduled launch (homologous regions side by side, table lines shared in L2 — the
        reference runs one thread per anchor FASTA instead, cpp/anchor.cpp:217-223), and host threads
        stream each genome's rows out of HBM into its BGZF files while the next batch is anchored."""
        print("Wrote config.yaml and samples.tsv")
        if self.prepare:
            print("Prepared. Run 'python -m panagram_amd index <dir>' to build the index")
            return
        from concurrent.futures import ThreadPoolExecutor
        os.makedirs(self.get_subdir("logs"), exist_ok=True)
        mine = self.my_anchor_genomes()
        if not mine:  # more ranks than anchor genomes
            logger.info("rank %d of %d: no anchor genome to write", self.rank, self.world)
            return
        tbl = self.build_table(keep=mine)
        nb = (self.ngenomes + 7) // 8
        # two batches of rows are resident at a time (one being written, one being anchored), next to the
        # table: with a table that fills most of the HBM the batches shrink to what is left
        free = self.context.mem_info()[0]
        batch_bytes = int(min(self.batch_bytes, max(1 << 30, (free - (6 << 30)) / 2.3)))
        batches, cur, cur_bytes = [], [], 0
        for name in mine:  # a batch's rows stay in HBM until written: bound them
            rows_bytes = int(self.seqset_for(name).lens.sum()) * nb
            if cur and cur_bytes + rows_bytes > batch_bytes:
                batches.append(cur)
                cur, cur_bytes = [], 0
            cur.append(name)
            cur_bytes += rows_bytes
        if cur:
            batches.append(cur)
        # (a writer job is bound by the file system — about 1.5 GB/s of compressed bytes each — not by the GPU;
        # each concurrent job pins its own staging at first use, which only pays for itself on big outputs)
        payload = sum(int(self.seqset_for(n).lens.sum()) for n in mine) * nb
        writers = int(os.environ.get("PG_WRITERS", "4" if payload > (4 << 30) else "2"))
        with ThreadPoolExecutor(max_workers=writers) as pool:
            previous = None
            for batch in batches:
                for name in batch:
                    self.genomes[name].setup_log(os.path.join(self.get_subdir("logs"), f"anchor.{name}.log.txt"))
                job = self._anchor_batch(tbl, batch)
                futs = [pool.submit(self.genomes[name].write_from_result, job, gi) for gi, name in enumerate(batch)]
                if previous is not None:  # at most two batches of rows resident
                    self._finish_batch(*previous)
                previous = (job, batch, futs)
            if previous is not None:
                self._finish_batch(*previous)
        self.close()

    def my_anchor_genomes(self) -> List[str]:
        """Multi-GPU (one process per GPU, e.g. under torchrun: RANK / WORLD_SIZE): the table is
        replicated — every rank builds it from all inputs — and the anchor GENOMES are dealt to the
        ranks longest-first (FASTA size); each rank writes the directories of its genomes.  Anchors are
        independent (cpp/anchor.cpp:217-223 runs them as OpenMP iterations), so there is no collective
        and the files do not depend on the GPU count."""
        if self.world <= 1:
            return list(self.anchor_genomes)
        from .distributed import plan_shards
        units = [(n, 0, os.path.getsize(self.genomes[n].fasta)) for n in self.anchor_genomes]
        mine = {u[0] for u in plan_shards(units, self.world)[self.rank]}
        return [n for n in self.anchor_genomes if n in mine]

    batch_bytes = 32 << 30  # rows of one batch of anchor genomes held in HBM (bitmap.1 payload bytes)
    # BGZF compression of the bitmaps: a zlib level (host threads), or -2 = on the GPU (k_row_deflate)
    bgzf_level = int(os.environ.get("PG_BGZF_LEVEL", "-2"))

    def _anchor_batch(self, tbl, batch):
        sets = [self.seqset_for(name) for name in batch]
        merged = engine.SeqSet.concat(self.context, sets) if len(sets) > 1 else sets[0]
        for nm, ln in zip(merged.names, merged.lens):
            if int(ln) < tbl.k:
                logger.warning(f"Contig {nm} is shorter than k={tbl.k}: 0 k-mers (the reference underflows here)")
        logger.info("Anchoring Started")
        res = engine.AnchorResult(tbl, merged, colsums=True)
        first = np.cumsum([0] + [len(s.names) for s in sets])
        if len(sets) > 1:
            res.coschedule(np.repeat(np.arange(len(sets)), [len(s.names) for s in sets]))
        res.run()
        small = [res.download(ci, want_bitmap1=False, want_bitmap100=False) for ci in range(len(merged.names))]
        ccs = res.contig_colsums().astype(np.int64)
        per_genome = []
        for gi, name in enumerate(batch):
            lo, hi = int(first[gi]), int(first[gi + 1])
            g = self.genomes[name]
            names = list(merged.names[lo:hi])
            gene_hists = g._tabulate_genes(res, names, small[lo:hi], lo) if g.annotated else None
            per_genome.append((lo, hi, names, small[lo:hi], ccs[lo:hi].sum(axis=0), gene_hists))
        return dict(res=res, merged=merged if len(sets) > 1 else None, genomes=per_genome)

    def _finish_batch(self, job, batch, futs):
        try:
            for f in futs:
                f.result()
        finally:
            job["res"].close()
            if job["merged"] is not None:
                job["merged"].close()
            # (the batch's packed sequences — 0.375 byte per base — stay until close(): every hipFree waits
            # for the device, i.e. for the writers' kernels of the other batch, 10-20 ms apiece)

    # ---- bitmap -> bins (index.py:438-465): what the viewer does with a queried bitmap ----
    @property
    def bitsum_index(self):
        return pd.RangeIndex(0, self.ngenomes + 1)

    @staticmethod
    def _bin_ids(bitmap: pd.DataFrame, binlen: int):
        bins = np.asarray(bitmap.index) // binlen
        ub, inv = np.unique(bins, return_inverse=True)
        return ub, inv

    def bitmap_to_bins(self, bitmap: pd.DataFrame, binlen: int):
        """(pancount_bins, paircount_bins) of a positions x genomes 0/1 bitmap: rows = occupancy 0..N
        and columns = bin number; rows = genomes and columns = bin start, every bin scaled by its
        largest genome count (index.py:438-449)."""
        ub, inv = self._bin_ids(bitmap, binlen)
        vals = bitmap.to_numpy()
        N = self.ngenomes
        flat = np.bincount(inv * (N + 1) + vals.sum(axis=1).astype(np.int64), minlength=len(ub) * (N + 1))
        pan = pd.DataFrame(flat.reshape(len(ub), N + 1).T, index=self.bitsum_index, columns=ub)
        return pan, self._paircount_bins(vals, ub, inv, binlen, bitmap.columns)

    @staticmethod
    def _paircount_bins(vals, ub, inv, binlen, columns):
        sums = np.zeros((len(ub), vals.shape[1]), np.int64)
        np.add.at(sums, inv, vals.astype(np.int64))
        pc = pd.DataFrame(sums.T, index=columns, columns=ub * binlen)
        return pc.div(pc.max(axis=0), axis=1)

    def bitmap_to_paircount_bins(self, bitmap: pd.DataFrame, binlen: int):
        ub, inv = self._bin_ids(bitmap, binlen)
        return self._paircount_bins(bitmap.to_numpy(), ub, inv, binlen, bitmap.columns)

    def bitmap_to_pancount(self, bitmap: pd.DataFrame) -> pd.Series:
        return pd.Series(bitmap.to_numpy().sum(axis=1), index=bitmap.index)

    def pancount_to_bins(self, pancnts: pd.Series, binlen: int) -> pd.DataFrame:
        bins = np.asarray(pancnts.index) // binlen
        ub, inv = np.unique(bins, return_inverse=True)
        N = self.ngenomes
        flat = np.bincount(inv * (N + 1) + pancnts.to_numpy().astype(np.int64), minlength=len(ub) * (N + 1))
        return pd.DataFrame(flat.reshape(len(ub), N + 1).T, index=self.bitsum_index, columns=ub)

    def query_bitmap(self, genome, chrom, start=None, end=None, step=1):
        return self.genomes[genome].query(chrom, start, end, step)

    def close(self):
        for nm in list(self._seqsets):
            self.drop_seqset(nm)
        if self._table is not None:
            self._table.close()
            self._table = None
        if self._ctx is not None:
            self._ctx.close()
            self._ctx = None


class Genome:
    """One sample (index.py:468-1189), write side = ``run_anchor``, read side = ``query``."""

    def __init__(self, idx: Index, id: int, name: str, fasta=None, gff=None, anchor=None, write=False):
        self.index = idx
        self.id = id
        self.name = name
        self.fasta = fasta
        self.gff = gff
        self.write_mode = write
        self.prefix = os.path.join(idx.prefix, ANCHOR_DIR, name)
        self.anchored = bool(anchor) if anchor is not None else (fasta is not None and not pd.isna(fasta))
        self.ngenomes = idx.ngenomes
        self.nbytes = int(np.ceil(self.ngenomes / 8))
        self.steps = list(idx.steps)
        self.chrs = None
        self.blocks = None
        if self.anchored and os.path.exists(self.chrs_fname):
            self.load_chrs()

    @property
    def chrs_fname(self):
        return os.path.join(self.prefix, "chrs.tsv")

    @property
    def chr_genes_fname(self):
        return os.path.join(self.prefix, "bitsum.genes.tsv")

    @property
    def annotated(self) -> bool:
        return self.gff is not None and not pd.isna(self.gff)

    def load_genes(self) -> pd.DataFrame:
        """The GFF records whose type is a gene type (index.py:669-691,731-736), sorted by (chr,
        start); ``start`` / ``end`` are used as they stand in the file, like the reference does
        when it slices ``bitsum[start:end]`` (index.py:1056-1063)."""
        df = pd.read_csv(self.gff, sep="\t", comment="#", header=None,
                         names=["chr", "source", "type", "start", "end", "score", "strand", "phase", "attr"],
                         usecols=["chr", "type", "start", "end"], dtype={"chr": str})
        df = df[df["type"].isin(self.index.gff_gene_types)]
        return df.sort_values(["chr", "start"], kind="stable").reset_index(drop=True)

    @property
    def bins_fname(self):
        return os.path.join(self.prefix, "bitsum.bins.tsv")

    def bitmap_gz_fname(self, step):
        return os.path.join(self.prefix, f"bitmap.{step}.{BGZ_SUFFIX}")

    def bitmap_gzi_fname(self, step):
        return os.path.join(self.prefix, f"bitmap.{step}.{IDX_SUFFIX}")

    @property
    def anchor_filenames(self):
        if not self.anchored:
            return []
        ret = [self.chrs_fname, self.bins_fname]
        for s in self.steps:
            ret += [self.bitmap_gz_fname(s), self.bitmap_gzi_fname(s)]
        return ret

    def iter_fasta(self):
        return read_fasta(self.fasta)

    # ---- chrs / offsets (index.py:576-604) ----
    def set_chrs(self, chrs: pd.DataFrame):
        self.chrs = chrs
        if "gene_count" not in self.chrs.columns:
            self.chrs["gene_count"] = 0
        self.sizes = chrs["size"]
        step_sizes = pd.DataFrame({step: np.ceil(self.sizes / step) for step in self.steps}, dtype=int)
        self.offsets = step_sizes.cumsum().shift(fill_value=0)

    def load_chrs(self):
        self.set_chrs(pd.read_table(self.chrs_fname, index_col="name"))

    # ---- WRITE: the hot path ----
    def anchor_contigs(self, table: engine.PanTable, seqs: Sequence[bytes]):
        """GPU compute for a list of contigs: returns ([(rows, rows100, bins, info)], colsums)."""
        ctx = table.ctx
        ss = engine.SeqSet.from_host(ctx, seqs)
        res = engine.AnchorResult(table, ss, colsums=True)
        res.run()
        out = [res.download(ci) for ci in range(len(seqs))]
        cs = res.colsums().astype(np.int64)
        res.close()
        ss.close()
        return out, cs

    def write_outputs(self, names: Sequence[str], results, paircount_sums: np.ndarray,
                      bgzf_threads: Optional[int] = None):
        """Write anchor/<name>/ exactly as the reference lays it out (cpp/anchor.cpp:37-109,
        index.py:1035-1094).  ``results[i] = (rows, rows100, bins, info)`` for contig i in
        FASTA order; files are written to temporaries and renamed, chrs.tsv last."""
        N = self.ngenomes
        os.makedirs(self.prefix, exist_ok=True)
        nthreads = bgzf_threads or self._bgzf_threads()
        tmp = {s: self.bitmap_gz_fname(s) + ".tmp" for s in self.steps}
        # one-byte rows: equal rows are byte runs (zlib RLE); wider rows: the row-aware encoder
        level = 6 | (engine.BgzfWriter.RLE if self.nbytes == 1 else engine.BgzfWriter.ROWS(self.nbytes) if self.nbytes < 256 else 0)
        writers = {s: engine.BgzfWriter(tmp[s], level=level, threads=nthreads) for s in self.steps}
        for rows, rows100, _, _ in results:
            writers[1].write(rows)
            writers[self.steps[1]].write(rows100)
        for s in self.steps:
            writers[s].close(self.bitmap_gzi_fname(s) + ".tmp")
            os.replace(tmp[s], self.bitmap_gz_fname(s))
            os.replace(self.bitmap_gzi_fname(s) + ".tmp", self.bitmap_gzi_fname(s))
        self._write_tables(names, [(b, info) for _, _, b, info in results], paircount_sums)

    def setup_log(self, logfile: Optional[str]):
        if logfile:
            logging.basicConfig(filename=logfile, level=logging.INFO,
                                format="[ %(asctime)s %(levelname)7s ] %(message)s", datefmt="%Y-%m-%d %H:%M:%S")

    def anchor_on_gpu(self, table: engine.PanTable, ss: engine.SeqSet):
        """Enqueue the anchor kernels for a packed FASTA and fetch the small outputs (bins, column
        sums); the bitmap rows stay in HBM for ``write_from_result``."""
        for nm, ln in zip(ss.names, ss.lens):
            if int(ln) < table.k:
                logger.warning(f"Contig {nm} is shorter than k={table.k}: 0 k-mers (the reference underflows here)")
        logger.info("Anchoring Started")
        res = engine.AnchorResult(table, ss, colsums=True)
        res.run()
        small = [res.download(ci, want_bitmap1=False, want_bitmap100=False) for ci in range(len(ss.names))]
        cs = res.colsums().astype(np.int64)
        gene_hists = self._tabulate_genes(res, list(ss.names), small) if self.annotated else None
        return dict(res=res, merged=None, genomes=[(0, len(ss.names), list(ss.names), small, cs, gene_hists)])

    def _tabulate_genes(self, res, names, small, first_contig: int = 0):
        """occupancy histogram of every gene's positions, summed per chromosome (index.py:1055-1064,
        1079-1082), from the rows in HBM: {chrom: (gene_count, hist[N+1])} in sorted-chr order"""
        genes = self.load_genes()
        out = {}
        for chrom, grp in genes.groupby("chr", sort=True):
            if chrom not in names:
                out[chrom] = (len(grp), np.zeros(self.ngenomes + 1, np.int64))
                continue
            ci = names.index(chrom)
            size = small[ci][3]["nkmers"]
            st, en = grp["start"].to_numpy(np.int64), grp["end"].to_numpy(np.int64)
            ok = (en > st) & (st >= 0) & (en <= size)
            for s_, e_ in zip(st[~ok], en[~ok]):
                logger.warning(f"Skipping gene at {chrom}:{s_}-{e_}, coordinates out-of-bounds")
            hist = np.zeros(self.ngenomes + 1, np.int64)
            if ok.any():
                h, _ = res.window_stats(first_contig + ci, st[ok], en[ok], step=1, colsums=False)
                hist = h.sum(axis=0).astype(np.int64)
            out[chrom] = (len(grp), hist)
            logger.info(f"Annotated {chrom}")
        return out

    def write_from_result(self, job, gi: int = 0, bgzf_threads: Optional[int] = None):
        """anchor/<name>/ exactly as the reference lays it out (cpp/anchor.cpp:37-109,
        index.py:1035-1094): this genome's contig range of the (possibly shared) result, the two
        bitmaps streamed from HBM by the library.  The caller closes the result."""
        res = job["res"]
        lo, hi, names, small, cs, gene_hists = job["genomes"][gi]
        os.makedirs(self.prefix, exist_ok=True)
        nthreads = bgzf_threads or self._bgzf_threads()
        for s in self.steps:
            gz, gzi = self.bitmap_gz_fname(s), self.bitmap_gzi_fname(s)
            res.write_bgzf(s, gz + ".tmp", gzi + ".tmp", level=self.index.bgzf_level, threads=nthreads, first_contig=lo,
                           ncontigs=hi - lo)
            os.replace(gz + ".tmp", gz)
            os.replace(gzi + ".tmp", gzi)
        self._write_tables(names, [(b, info) for _, _, b, info in small], cs, gene_hists)

    def _bgzf_threads(self) -> int:
        return max(1, min(64, self.index.cores if self.index.cores > 1 else engine.usable_cpus()))

    def _write_tables(self, names, bins_infos, paircount_sums, gene_hists=None):
        N = self.ngenomes
        bins_rows: List[str] = ["chr\tstart" + "".join(f"\t{i}" for i in range(N + 1)) + "\n"]
        chr_rows: List[Tuple[str, int, int, int]] = []
        for ci, (chrom, (bins, info)) in enumerate(zip(names, bins_infos)):
            starts = np.arange(info["nbins"], dtype=np.int64) * info["binlen"]
            body = np.column_stack([np.full(info["nbins"], ci, np.int64), starts, bins.astype(np.int64)])
            bins_rows.extend("\t".join(map(str, row)) + "\n" for row in body.tolist())
            gene_count = gene_hists[chrom][0] if gene_hists and chrom in gene_hists else 0
            chr_rows.append((chrom, ci, info["nkmers"], gene_count))
            logger.info(f"Anchored {chrom}")
        with open(self.bins_fname, "w") as f:
            f.writelines(bins_rows)
        if gene_hists is not None:  # bitsum.genes.tsv: one row per annotated chromosome (index.py:1079-1082)
            pd.DataFrame([h for _, h in gene_hists.values()], index=pd.Index(list(gene_hists), name="chr"),
                         columns=range(N + 1)).to_csv(self.chr_genes_fname, sep="\t")
        # total_paircounts.csv (index.py:1068-1074): count[g] = positions holding genome g's bit
        counts = pd.Series(np.asarray(paircount_sums, dtype=np.int64), index=self.index.genome_names)
        pd.DataFrame({"count": counts, "frac": counts / counts[self.name]}).to_csv(
            os.path.join(self.prefix, "total_paircounts.csv"))
        chrs = pd.DataFrame(chr_rows, columns=["name", "id", "size", "gene_count"]).set_index("name")
        self.set_chrs(chrs)
        self.chrs.to_csv(self.chrs_fname, sep="\t")  # written last: it is the rule's completion marker

    def run_anchor(self, table: engine.PanTable, logfile: Optional[str] = None, bgzf_threads: Optional[int] = None):
        """Counterpart of ``Genome.run_anchor(bitvecs, logfile)`` (index.py:1012-1097) and of
        ``KMCdb::anchor_fasta`` (cpp/anchor.cpp:37-109): walks the anchor FASTA, anchors every
        contig on the GPU and writes the anchor/<name>/ files.  ``table`` is the GPU-resident
        pan-kmer table standing in for the list of ``kmc/bitvec{i}`` prefixes."""
        self.setup_log(logfile)
        if not self.anchored:
            logger.info(f"Skipping non-anchor genome '{self.name}'")
            return
        ss = self.index.seqset_for(self.name)
        job = None
        try:
            job = self.anchor_on_gpu(table, ss)
            self.write_from_result(job, 0, bgzf_threads)
        finally:
            if job is not None:
                job["res"].close()
            self.index.drop_seqset(self.name)

    # ---- READ: what `panagram view` does with our files (index.py:615-658, 793-845) ----
    def init_read(self):
        if self.chrs is None:
            self.load_chrs()
        self.blocks = {s: load_bgz_blocks(self.bitmap_gzi_fname(s)) for s in self.steps}
        self.bitsum_bins = pd.read_table(self.bins_fname)
        self.total_paircounts = pd.read_csv(os.path.join(self.prefix, "total_paircounts.csv"), index_col="name")

    def seq_len(self, seq_name):
        return int(self.sizes.loc[seq_name])

    def _bytes_to_bits(self, pac):
        return np.unpackbits(pac, bitorder="little", axis=1)[:, : self.ngenomes]

    def _query_bytes(self, name, start, end, step, bstep):
        byte_start = self.nbytes * (int(self.offsets.loc[name, bstep]) + (start // bstep))
        length = int((end - start) // bstep) + 1
        step = step // bstep
        buf = bgzf_read(self.bitmap_gz_fname(bstep), self.blocks[bstep], byte_start, length * self.nbytes)
        pac = np.frombuffer(buf, "uint8").reshape((len(buf) // self.nbytes, self.nbytes))
        return pac[::step] if step > 1 else pac

    def query(self, name, start=None, end=None, step=1) -> pd.DataFrame:
        if self.blocks is None:
            self.init_read()
        bstep = 1
        for s in self.steps:
            if step % s == 0:
                bstep = max(bstep, s)
        start = 0 if start is None else start
        end = self.seq_len(name) if end is None else end
        pac = self._query_bytes(name, start, end - 1, step, bstep)
        bits = self._bytes_to_bits(pac)
        return pd.DataFrame(bits, index=pd.RangeIndex(start, end, step)[: len(bits)], columns=self.index.genome_names)


# ---------------------------------------------------------------------------
# process-level seam: `run_anchor <ngenomes> <root> [<name> <fasta>]...` (cpp/anchor.cpp:204-233)
# ---------------------------------------------------------------------------
def run_anchor_cli(argv: Sequence[str], device: int = 0) -> int:
    """Same argv and file contract as the reference's ``cpp/run_anchor`` binary: reads
    ``root/kmc/bitvec{i}`` (KMC1), writes ``root/anchor/<name>/{bitmap.1,bitmap.100}.gz(.gzi)``,
    ``bitsum.bins.tsv``, ``chrs.tsv``.  Unlike the reference, a missing or ill-formed DB is an
    error (cpp/anchor.cpp:29 ignores OpenForRA's result)."""
    ngenomes = int(argv[0])
    root = argv[1]
    pairs = argv[2:]
    if len(pairs) > 2 * ngenomes:
        print(f"Error: expected {ngenomes * 2 + 1} or fewer arguments")
        return 1
    ndbs = (ngenomes + 31) // 32
    ctx = engine.Context(device)
    k = None
    images = []
    for i in range(ndbs):
        p = os.path.join(root, "kmc", f"bitvec{i}")
        with open(p + ".kmc_pre", "rb") as f:
            pre = f.read()
        with open(p + ".kmc_suf", "rb") as f:
            suf = f.read()
        hoff = struct.unpack("<I", pre[-8:-4])[0]
        k = struct.unpack("<I", pre[len(pre) - 8 - hoff:len(pre) - 4 - hoff])[0]
        images.append((pre, suf))
    tbl = engine.PanTable(ctx, k, ngenomes)
    for i, (pre, suf) in enumerate(images):
        tbl.load_kmc1(i, pre, suf)
    nbytes = (ngenomes + 7) // 8
    for name, fasta in zip(pairs[0::2], pairs[1::2]):
        print(f"Anchoring {name} {fasta}")
        adir = os.path.join(root, "anchor", name)
        os.makedirs(adir, exist_ok=True)
        recs = list(read_fasta(fasta))
        ss = engine.SeqSet.from_host(ctx, [s for _, s in recs])
        res = engine.AnchorResult(tbl, ss, colsums=False)
        res.run()
        nb_row = (ngenomes + 7) // 8
        level = 6 | (engine.BgzfWriter.RLE if nb_row == 1 else engine.BgzfWriter.ROWS(nb_row) if nb_row < 256 else 0)
        w1 = engine.BgzfWriter(os.path.join(adir, "bitmap.1.gz"), level=level, threads=engine.usable_cpus())
        w100 = engine.BgzfWriter(os.path.join(adir, "bitmap.100.gz"), level=level, threads=2)
        with open(os.path.join(adir, "bitsum.bins.tsv"), "w") as fb, open(os.path.join(adir, "chrs.tsv"), "w") as fc:
            fb.write("chr\tstart" + "".join(f"\t{i}" for i in range(ngenomes + 1)) + "\n")
            fc.write("name\tid\tsize\tgene_count\n")
            for ci, (chrom, _) in enumerate(recs):
                rows, rows100, bins, info = res.download(ci)
                w1.write(rows)
                w100.write(rows100)
                for b in range(info["nbins"]):
                    fb.write(f"{ci}\t{b * info['binlen']}" + "".join(f"\t{int(c)}" for c in bins[b]) + "\n")
                fc.write(f"{chrom}\t{ci}\t{info['nkmers']}\t0\n")
        w1.close(os.path.join(adir, "bitmap.1.gzi"))
        w100.close(os.path.join(adir, "bitmap.100.gzi"))
        res.close()
        ss.close()
    tbl.close()
    ctx.close()
    return 0
