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


class _PinnedPool:
    """up to ``budget`` page-locked buffers of ``cap`` bytes (engine.HostBuffer) that go round between the reader threads and
    the thread that uploads their contents; a reader locks the memory of a new one itself (in parallel with the others' reads)
    as long as the budget is not spent.  ``get()`` returns None once locking memory has failed: read into pageable memory."""

    def __init__(self, make, cap: int, budget: int):
        import queue
        import threading
        self.make, self.cap, self.budget = make, cap, budget
        self.free, self.lock, self.made, self.failed, self.all = queue.Queue(), threading.Lock(), 0, False, []

    def get(self):
        import queue
        try:
            return self.free.get_nowait()
        except queue.Empty:
            pass
        with self.lock:
            create = not self.failed and self.made < self.budget
            if create:
                self.made += 1
        if create:
            try:
                buf = self.make(self.cap)
            except Exception:  # noqa: BLE001 — no memory to lock
                with self.lock:
                    self.made -= 1
                    self.failed = True
                return None
            with self.lock:
                self.all.append(buf)
            return buf
        if self.failed:  # (fewer buffers than files in flight: waiting for one could wait for a LATER file's, which is handed on only after this one)
            return None
        return self.free.get()

    def put(self, buf) -> None:
        self.free.put(buf)

    def close(self) -> None:
        for b in self.all:
            b.close()
        self.all = []


def _read_fasta_pinned(path: str, pool: _PinnedPool):
    """a plain FASTA file read into a page-locked buffer of ``pool``: ``(uint8 view of the text, the buffer)`` — the caller puts
    the buffer back once the text is on the device; compressed files, and any file when no memory can be locked, as
    ``_read_fasta_image`` reads them: ``(image, None)``"""
    if path.endswith((".gz", ".bgz")):
        return _read_fasta_image(path), None
    buf = pool.get()
    if buf is None:
        return _read_fasta_image(path), None
    try:
        n = 0
        with open(path, "rb", buffering=0) as f:
            view = memoryview(buf.array)
            while n < len(view):
                got = f.readinto(view[n:])
                if not got:
                    break
                n += got
            if n == len(view) and f.read(1):  # (the file grew since it was sized)
                raise OSError(f"{path} is larger than when the index was prepared")
        return buf.array[:n], buf
    except BaseException:
        pool.put(buf)
        raise


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

    # how a multi-process run divides the work (see run()): None = decide from the table's size
    shard: Optional[str] = dataclasses.field(default_factory=lambda: os.environ.get("PG_SHARD") or None)
    genome_blocks: int = dataclasses.field(default_factory=lambda: int(os.environ.get("PG_GENOME_BLOCKS", "0")))
    # build the replicated table from the genomes this process anchors, the other samples only set bits (PG_FULL_TABLE=1: all k-mers)
    filtered_table: bool = dataclasses.field(default_factory=lambda: os.environ.get("PG_FULL_TABLE", "") in ("", "0"))

    _EXTRA = ("device", "export_kmc", "rank", "world", "shard", "genome_blocks", "filtered_table")
    # keys of config.yaml that describe one invocation, not the index: not taken over when a directory is re-opened
    # (`prepare` is written for schema compatibility, but a later `index <dir>` run must not stop at "Prepared")
    _NOT_IN_CONFIG = ("input", "mode", "prefix", "prepare")
    SAMPLE_COLUMNS = ("fasta", "gff", "id", "anchor")  # after the index column `name` (index.py:282-293)

    # ---- opening an index: three ways in (index.py:196-268), one handler each ----
    def __post_init__(self):
        if self.mode not in (None, "r", "w"):
            raise ValueError(f"Invalid mode '{self.mode}', must be 'r' or 'w'")
        kind = "dir" if os.path.isdir(self.input) else "file" if os.path.isfile(self.input) else None
        self.write_mode = kind == "file" if self.mode is None else self.mode == "w"
        opener = {(True, "dir"): self._reopen_prepared, (True, "file"): self._create_from_samples,
                  (False, "dir"): self._open_readonly}.get((self.write_mode, kind))
        if opener is None:
            raise ValueError("Index input must be sample TSV or initialized directory" if self.write_mode
                             else "Index input must be directory mode='r'")
        opener()
        self._check_geometry()  # (after the config file has had its say)
        self._load_samples()
        self._ctx = None
        self._table = None
        self._table_scope = "all"
        self.timings: Dict[str, float] = {}  # seconds spent in load_inputs / build_table (host wall clock), for reports
        self._seqsets: Dict[str, engine.SeqSet] = {}

    def _reopen_prepared(self):
        self.prefix = self.input
        if not all(os.path.isfile(f) for f in (self.config_fname, self.samples_fname)):
            raise ValueError("Index write directory not initialized")
        self.input = self.samples_fname
        self.load_config()

    def _create_from_samples(self):
        self.prefix = self.prefix if self.prefix is not None else os.path.dirname(self.input)
        self.prefix = self.prefix or "."
        os.makedirs(self.prefix, exist_ok=True)
        self.init_config()

    def _open_readonly(self):
        self.prefix = self.input
        self.load_config()

    def _check_geometry(self):
        for nm in ("lowres_step", "max_bin_kbp", "min_bin_count"):
            v = getattr(self, nm)
            if not isinstance(v, (int, np.integer)) or v < 1:
                raise ValueError(f"{nm} must be a positive integer, got {v!r}")
        if self.lowres_step == 1:
            raise ValueError("lowres_step=1 would name the low-resolution bitmap like the full one (bitmap.1)")

    @property
    def result_geometry(self) -> dict:
        """what every AnchorResult of this index is created with (index.py:101-106, 1169-1172)"""
        return dict(lowres_step=int(self.lowres_step), max_bin_len=int(self.max_bin_kbp) * 1000,
                    min_bin_count=int(self.min_bin_count))

    def _load_samples(self):
        self.samples = pd.read_table(self.samples_fname)
        missing = {"name", "id"} - set(self.samples.columns)
        if missing:
            raise ValueError(f"{self.samples_fname} is missing required column(s): {', '.join(sorted(missing))}. "
                             "This directory does not look like a prepared Panagram index.")
        self.samples = self.samples.set_index("name")
        self.ngenomes = len(self.samples)
        self.genomes: Dict[str, Genome] = {
            name: Genome(self, int(row["id"]), name, row["fasta"], row.get("gff"), bool(row["anchor"]), write=self.write_mode)
            for name, row in self.samples.iterrows()}
        if self.anchor_genomes is None:
            self.anchor_genomes = [n for n, g in self.genomes.items() if g.anchored]

    # ---- naming (index.py:155-165, 359-405) ----
    @property
    def params(self):
        return {k_: v for k_, v in dataclasses.asdict(self).items() if k_ not in self._EXTRA}

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
        return (len(self.samples) + 31) // 32

    @property
    def bitvec_prefixes(self):
        return [self.kmc_prefix(f"bitvec{i}") for i in range(self.kmc_bitvec_count)]

    @property
    def opdef_filenames(self):
        return [self.kmc_prefix(f"opdef{i}.txt") for i in range(self.kmc_bitvec_count)]

    def write_opdefs(self):
        """The `kmc_tools complex` operation files of the reference workflow (index.py:407-426, rule `opdefs`
        workflow/Snakefile:71-79): per 32-sample group, its samples' one-hot databases as inputs and
        ``bitvec{i} = s0 + s1 + ... -ocsum`` as output.  The GPU table build needs none of this; the files are written
        next to the exported ``bitvec{i}`` databases (``--export_kmc``) for a user who keeps `kmc_tools` in the loop."""
        names = [str(n) for n in self.samples.index]
        os.makedirs(self.get_subdir("kmc"), exist_ok=True)
        for i, fname in enumerate(self.opdef_filenames):
            group = names[32 * i:32 * (i + 1)]
            text = ["INPUT:"] + [f"{n} = {self.kmc_prefix(n, 'onehot')}" for n in group]
            text += ["OUTPUT:", f"{self.kmc_prefix(f'bitvec{i}')} = " + " + ".join(group), "-ocsum"]
            with open(fname, "w") as f:
                f.write("\n".join(text) + "\n")

    @property
    def steps(self):
        return (1, self.lowres_step)

    def __getitem__(self, genome):
        return self.genomes[genome]

    # ---- config files (index.py:269-295, 347-357) ----
    @staticmethod
    def _write_atomically(path: str, write) -> None:
        """several ranks of one multi-GPU job write the same bytes: write-then-rename keeps readers whole"""
        tmp = f"{path}.{os.getpid()}.tmp"
        write(tmp)
        os.replace(tmp, path)

    def _normalised_samples(self, table: pd.DataFrame) -> pd.DataFrame:
        """The user's table -> the schema samples.tsv is stored in: index ``name``; columns fasta, gff, id, anchor."""
        if not {"name", "fasta"} <= set(table.columns):
            raise ValueError("Input samples must contain 'name' and 'fasta' column headers")
        bad = [str(n) for n in table["name"] if not re.fullmatch(NAME_REGEX, str(n))]
        if bad:
            raise ValueError("Invalid genome names: '" + "', '".join(bad) + f"'\nMust match r'{NAME_REGEX}'.")
        out = table.reindex(columns=["name", "fasta", "gff"] + (["anchor"] if "anchor" in table else []))
        out = out.set_index("name").dropna(how="all")
        out["id"] = np.arange(len(out), dtype=int)
        if self.anchor_genomes is None:  # the table's own anchor column, else every sample with a FASTA
            chosen = out["anchor"].astype(bool) if "anchor" in out else out["fasta"].notna()
            self.anchor_genomes = list(out.index[chosen])
        out["anchor"] = out.index.isin(self.anchor_genomes)
        return out[list(self.SAMPLE_COLUMNS)]

    def init_config(self):
        samples = self._normalised_samples(pd.read_table(self.input))
        self._write_atomically(self.samples_fname, lambda tmp: samples.to_csv(tmp, sep="\t"))
        self.write_config()

    def write_config(self, exclude=("prefix",)):
        prms = {k_: v for k_, v in self.params.items() if k_ not in exclude}

        def dump(tmp):
            with open(tmp, "w") as f:
                yaml.dump(prms, f)
        self._write_atomically(self.config_fname, dump)

    def load_config(self):
        with open(self.config_fname) as f:
            stored = yaml.load(f, yaml.SafeLoader) or {}
        for key, val in stored.items():
            if key in self._NOT_IN_CONFIG:
                continue
            section = getattr(self, key, None)
            if dataclasses.is_dataclass(section) and isinstance(val, dict):
                for sub, v in val.items():  # nested sections (kmc, genome_umap, chrom_umap) are updated in place
                    setattr(section, sub, v)
            else:
                setattr(self, key, val)

    # ---- the table: replaces rules kmc_count / opdefs / kmc_bitvec (workflow/Snakefile:54-110) ----
    @property
    def context(self) -> engine.Context:
        if self._ctx is None:
            self._ctx = engine.Context(self.device)
        return self._ctx

    def _pinned_reader_pool(self, names, count: int):
        """a ``_PinnedPool`` for reading the plain FASTA files of ``names`` — buffers as large as the largest of them, at most
        ``count`` and 8 GB in all — or None: less than 256 MB to read that way, or an engine without page-locked buffers (the CPU
        tests' stand-in); the files are then read into pageable arrays"""
        sizes = [os.path.getsize(self.genomes[n].fasta) for n in names if not self.genomes[n].fasta.endswith((".gz", ".bgz"))]
        make = getattr(self.context, "host_buffer", None)
        if not sizes or make is None or os.environ.get("PG_PINNED_READS", "1") in ("0", ""):
            return None
        # (a small job gains nothing: locking and unlocking the pool's memory costs 40 ms per GB each way; PG_PINNED_READS=2: any size)
        if sum(sizes) < (256 << 20) and os.environ.get("PG_PINNED_READS", "1") != "2":
            return None
        cap = max(sizes) + 1  # (+ 1: a file that grew is noticed)
        if cap > (4 << 30):
            return None
        # (four buffers keep the upload busy — a reader fills one in 10-20 ms, the parser empties one in 5-7 — and cost half of what
        # eight cost to lock and unlock: 8 x 100 Mb 0.43 -> 0.39 s, 64 x 200 Mb 2.1-2.2 -> 1.9-2.0 s, tools/e2e_fresh.py)
        count = int(os.environ.get("PG_PINNED_BUFFERS", "0")) or min(count, 4)
        return _PinnedPool(make, cap, max(2, min(count, len(sizes), (8 << 30) // cap)))

    def load_inputs(self):
        """Every sample's sequence parsed and packed on the GPU (0.375 byte per base), with the HyperLogLog
        registers of its distinct canonical k-mers.  Returns ``[(name, genome, seqset, min_count, registers)]`` in
        sample order.  Sketches merge by register-wise maximum, so the distinct k-mers of any union of samples —
        the whole pangenome, or one block of genomes — can be estimated before a table is allocated."""
        if getattr(self, "_inputs", None) is not None:
            return self._inputs
        import time
        from concurrent.futures import ThreadPoolExecutor
        t_start = time.perf_counter()
        inputs = []
        sketch = engine.KmerSketch(self.context, self.k)
        # the FASTA files are read a few ahead by host threads while the GPU parses and sketches — plain files straight into a
        # small pool of page-locked buffers (engine.HostBuffer) that go round: the text then goes up by DMA instead of through
        # the runtime's staging copy of pageable memory, and a file costs neither 50 000 page faults nor their unmapping
        # (64 x 200 Mb: 22 -> 10 ms per file beside the readers, tools/attic/r6_load_inputs_breakdown.py)
        todo = [n for n, g in self.genomes.items() if not pd.isna(g.fasta) and not is_fastq(g.fasta) and n not in self._seqsets]
        nread = int(os.environ.get("PG_READERS", "0")) or max(2, min(6, engine.usable_cpus() // 2))
        reader = ThreadPoolExecutor(max_workers=nread)
        pool = self._pinned_reader_pool(todo, nread + 2)
        read = (lambda path: _read_fasta_pinned(path, pool)) if pool is not None else (lambda path: (_read_fasta_image(path), None))
        # (never more files in flight than buffers: a reader that waited for one held by a LATER file's finished read would wait for ever)
        window = pool.budget if pool is not None else nread + 2
        ahead = {n: reader.submit(read, self.genomes[n].fasta) for n in todo[:window]}
        nxt = window
        for name, g in self.genomes.items():
            if pd.isna(g.fasta):
                continue
            if name in ahead:
                t0 = time.perf_counter()
                image, buf = ahead.pop(name).result()
                t1 = time.perf_counter()
                self._seqsets[name] = engine.SeqSet.from_fasta(self.context, image)
                del image
                if buf is not None:
                    pool.put(buf)  # (the text is on the device: the buffer takes the next file)
                # (where a load's time goes: waiting for the readers / upload + parse + pack of the text)
                self.timings["load_wait_read_s"] = self.timings.get("load_wait_read_s", 0.0) + t1 - t0
                self.timings["load_parse_s"] = self.timings.get("load_parse_s", 0.0) + time.perf_counter() - t1
                if nxt < len(todo):
                    ahead[todo[nxt]] = reader.submit(read, self.genomes[todo[nxt]].fasta)
                    nxt += 1
            if is_fastq(g.fasta):
                # read sets: kmc -ci2 -fq (workflow/Snakefile:88-89) — k-mers seen once are dropped
                # (the sketch counts them too: the table is sized from above)
                if name in self.anchor_genomes:
                    raise ValueError(f"{name}: a FASTQ sample cannot be an anchor genome")
                ss, min_count = engine.SeqSet.from_host(self.context, [read_fastq_joined(g.fasta)]), 2
            else:
                ss, min_count = self.seqset_for(name), 1
            t0 = time.perf_counter()
            sketch.reset()
            sketch.add(ss)
            inputs.append((name, g, ss, min_count, sketch.registers()))
            self.timings["load_sketch_s"] = self.timings.get("load_sketch_s", 0.0) + time.perf_counter() - t0
        if pool is not None:
            reader.submit(pool.close)  # (unlocking the memory costs as much as locking it, 40 ms per GB: beside the table build)
        reader.shutdown(wait=False)
        sketch.close()
        self.context.trim()  # (the FASTA text buffer the parser kept for the next file)
        self._inputs = inputs
        self.timings["load_inputs_s"] = self.timings.get("load_inputs_s", 0.0) + time.perf_counter() - t_start
        return inputs

    @staticmethod
    def _expected_keys(inputs) -> int:
        """distinct canonical k-mers of the union of ``inputs`` (+3 %: the sketch's standard error is 0.4 %)"""
        if not inputs:
            return 1024
        regs = np.maximum.reduce([i[4] for i in inputs])
        est = engine.KmerSketch.estimate_registers(regs)
        return est + est // 32 + 1024

    def _replicated_keys(self, inputs) -> int:
        """distinct k-mers of the largest table a rank holds in the replicated mode: the union of the genomes it anchors
        (``build_table``'s filtered build) — or of all samples where that build does not apply.  The same answer in
        every process."""
        if not (self.filtered_table and not self.export_kmc and all(i[3] <= 1 for i in inputs)):
            return self._expected_keys(inputs)
        if self.world > 1 and os.environ.get("PG_PARTITION", "pieces") != "genomes":
            # pieces of homology classes: a rank's table is built from its 1 / world of the anchored sequence
            # (+30 %: the pieces are only balanced to a piece, and k-mers of repeats occur in several ranks' shares)
            return int(self._expected_keys([i for i in inputs if i[0] in self.anchor_genomes]) * 1.3 / self.world) + 1024
        writer = self.writer_of_anchor() if self.world > 1 else {a: 0 for a in self.anchor_genomes}
        worst = 0
        for r in range(max(1, self.world)):
            mine = [i for i in inputs if writer.get(i[0]) == r]
            worst = max(worst, self._expected_keys(mine) if mine else 0)
        return worst or self._expected_keys(inputs)

    def _roomy_density(self, expected_keys: int, anchors, inputs=None) -> float:
        """keys per 128-byte line for the table about to be built (0: the library's 3): sparser where HBM is plentiful — it has to
        leave room for two batches of rows (one being written, one being anchored: ``batch_bytes`` each at most) of the
        anchors given (names, or {name: SeqSet} of pieces) — and the genomes are not repeat-rich (the longest input's sketch:
        distinct k-mers per position).  engine.PanTable.roomy_density."""
        nb = (self.ngenomes + 7) // 8
        if isinstance(anchors, dict):
            rows = sum(int(ss.lens.sum()) for ss in anchors.values()) * nb
        else:
            rows = sum(os.path.getsize(self.genomes[n].fasta) for n in anchors if n in self.genomes) * nb  # (a byte of FASTA per position, about)
        distinct = None
        if inputs:
            big = max(inputs, key=lambda i: int(i[2].lens.sum()))
            npos = max(1, int(big[2].lens.sum()))
            distinct = min(1.0, engine.KmerSketch.estimate_registers(big[4]) / npos)
        return engine.PanTable.roomy_density(self.context, self.k, self.ngenomes, expected_keys, 2 * min(self.batch_bytes, rows) + rows // 50,
                                             distinct_fraction=distinct)

    def build_table(self, keep: Optional[Sequence[str]] = None, insert_sets: Optional[Dict[str, "engine.SeqSet"]] = None) -> engine.PanTable:
        """``keep``: the genomes whose packed sequences stay resident for the anchor step (default: all anchors) — and,
        in the filtered build, the genomes whose k-mers the table is built from.  ``insert_sets`` (the contig-sharded
        multi-GPU mode): ``{genome: SeqSet}`` of the PIECES this process anchors — the table is built from those pieces,
        every sample then only sets its bits."""
        keep = set(self.anchor_genomes if keep is None else keep)
        if self._table is not None:
            # a filtered table answers only for what it was built from: anything else would silently read as absent
            scope = self._table_scope
            if scope != "all" and (insert_sets is not None or scope == "pieces" or not keep <= scope):
                raise RuntimeError("the cached table was built for " + ("this process's pieces" if scope == "pieces" else f"the anchors {sorted(scope)}") +
                                   " only; close() the index before building a table for other anchors")
            return self._table
        have = all(os.path.exists(p + ".kmc_pre") and os.path.exists(p + ".kmc_suf") for p in self.bitvec_prefixes)
        scope = "all"
        import time
        t_start, t_inputs_before = time.perf_counter(), self.timings.get("load_inputs_s", 0.0)
        # how the table will be probed decides its minimizer window with the key count (pg_table_set_coscheduled): the anchor
        # genomes of a batch share ONE co-scheduled launch; a single anchor has no partner
        cosched = max(1, len(keep if insert_sets is None else insert_sets))
        if self.kmc.use_existing and have:
            tbl = engine.PanTable(self.context, self.k, self.ngenomes, coscheduled=cosched)
            for i, p in enumerate(self.bitvec_prefixes):
                tbl.load_kmc_files(i, p)
            logger.info("KMC Database Loaded")
        else:
            # every input is parsed and packed once on the GPU (0.375 byte per base) and stays
            # resident through the build; a sketch of the distinct k-mers over all of them sizes the
            # table (and settles its minimizer length) once: no re-hash while it grows, no second
            # copy of the table in HBM.  Anchors keep their sequences for the anchor step.
            inputs = self.load_inputs()
            # The anchor step only ever asks for k-mers of the sequence THIS process anchors (``keep`` genomes, or the
            # pieces in ``insert_sets``): the table is built from that and the other samples only set their bits in it
            # (pg_table_update_seqset) — the rows are the ones the table of all genomes gives, the table is as small as
            # the anchored sequence's own k-mer set (a rank of a multi-GPU run, or a pangenome in which only some genomes
            # are anchors).  Not when the merged databases are to be exported, nor with read-set samples (their -ci2
            # count tables merge through the inserting path).
            first = [i for i in inputs if i[0] in keep]
            rest = [i for i in inputs if i[0] not in keep]
            can_filter = self.filtered_table and not self.export_kmc and all(i[3] <= 1 for i in inputs)
            if insert_sets is not None and can_filter:
                sketch = engine.KmerSketch(self.context, self.k)
                for ss in insert_sets.values():
                    sketch.add(ss)
                est = sketch.estimate()
                sketch.close()
                expected = est + est // 32 + 1024
                tbl = engine.PanTable(self.context, self.k, self.ngenomes, expected_keys=expected, coscheduled=cosched,
                                      keys_per_line=self._roomy_density(expected, keep if insert_sets is None else insert_sets, inputs))
                for name, ss in insert_sets.items():
                    tbl.insert_seqset(self.genomes[name].id, ss)
                for name, g, ss, _, _ in inputs:
                    tbl.update_seqset(g.id, ss)
                    if name not in keep:
                        self.drop_seqset(name)
                scope, filtered = "pieces", True
            else:
                filtered = bool(can_filter and first and rest)
                expected = self._expected_keys(first if filtered else inputs)
                tbl = engine.PanTable(self.context, self.k, self.ngenomes, expected_keys=expected, coscheduled=cosched,
                                      keys_per_line=self._roomy_density(expected, keep, inputs))
                for name, g, ss, min_count, _ in (first + rest if filtered else inputs):
                    if filtered and name not in keep:
                        tbl.update_seqset(g.id, ss)
                    else:
                        tbl.insert_seqset(g.id, ss, min_count=min_count)
                    if min_count > 1:
                        ss.close()
                    elif name not in keep:
                        self.drop_seqset(name)
                if filtered:
                    scope = frozenset(i[0] for i in first)
            self._inputs = None
            logger.info("k-mer table built on GPU (sketch: %d distinct k-mers%s): %s", expected,
                        (" of this process's pieces" if scope == "pieces" else f" of the {len(first)} genomes anchored here") if filtered else "",
                        tbl.stats())
            if self.export_kmc:
                os.makedirs(self.get_subdir("kmc"), exist_ok=True)
                for i, p in enumerate(self.bitvec_prefixes):
                    keys, vals = tbl.export(i)
                    write_kmc1(p, keys, vals, self.k)
                self.write_opdefs()
        self.context.synchronize()
        # (the inserts alone: reading, parsing and sketching the inputs is load_inputs_s, also when it ran in here)
        self.timings["table_build_s"] = time.perf_counter() - t_start - (self.timings.get("load_inputs_s", 0.0) - t_inputs_before)
        self._table, self._table_scope = tbl, scope
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

    # ---- which multi-GPU mode ----
    HBM_RESERVE = 8 << 30  # left alone next to the table: packed sequences, descriptors, the writers' staging

    def plan_sharding(self):
        """("replicated", 1) — one table of all genomes on every GPU, anchor genomes dealt to the ranks, no
        collective — or ("genome", nblocks): the pangenome's tables do not fit one GPU next to a working set of
        rows, so the genomes are cut into ``nblocks`` contiguous blocks, a GPU holds the table of ONE block at a
        time, every GPU probes every anchor position against its block and the blocks' bit columns are
        all-gathered (``distributed.run_genome_sharded``).  ``nblocks`` is the smallest multiple of the world
        size whose largest block fits; more blocks than GPUs run as passes.  ``shard`` / ``genome_blocks``
        (``PG_SHARD``, ``PG_GENOME_BLOCKS``) override the choice."""
        if self.shard not in (None, "replicated", "genome"):
            raise ValueError(f"shard must be 'replicated' or 'genome', got {self.shard!r}")
        if self.shard == "replicated":
            return "replicated", 1
        have = all(os.path.exists(p + ".kmc_pre") and os.path.exists(p + ".kmc_suf") for p in self.bitvec_prefixes)
        if self.kmc.use_existing and have and self.shard is None:
            return "replicated", 1  # merged bitvec databases cannot be cut into genome blocks
        N = self.ngenomes
        if self.shard == "genome" and self.genome_blocks > 0:
            return "genome", min(N, self.genome_blocks)
        inputs = self.load_inputs()
        free = self._agreed_free_memory()
        nb = (N + 7) // 8
        longest = max([int(i[2].lens.sum()) for i in inputs if i[0] in self.anchor_genomes] or [0])
        # next to the table: two batches of rows (one being written, one being anchored), at least one anchor each
        budget = free - self.HBM_RESERVE - 2 * min(self.batch_bytes, max(longest * nb, 1 << 30))
        by_id = {i[1].id: i for i in inputs}
        if self.shard is None and engine.PanTable.bytes_for(self.k, N, self._replicated_keys(inputs)) <= budget:
            return "replicated", 1
        # Candidates: nblocks = world, 2 world, ... — FEWER blocks mean fewer passes over the anchors' positions, and a block
        # table may be created DENSER than the library's 3 keys per line to make its genomes' union fit (round 6,
        # pg_table_create_dense): a pass against a denser, wider block is slower (``_block_rate``, measured on BASELINE
        # configs[4]: 8 x 3 Gb at d = 0.05 on one GPU — two genomes per block at 3.4-3.6 keys per line take 4 passes of 0.27 s
        # where one genome per block took 8 of 0.21 s: 14.1 -> 23 G k-mers/s for the job), so the cheapest candidate by
        # passes / rate wins; the search ends at the first candidate that fits at the library's density — more blocks only
        # add passes from there.
        anchor_positions = sum(int(i[2].lens.sum()) for i in inputs if i[0] in self.anchor_genomes)
        best = None  # (cost, nblocks, keys_per_line)
        nblocks = max(1, self.world)
        while True:
            nblocks = min(nblocks, N)
            per = (N + nblocks - 1) // nblocks
            keys = 0
            for b0 in range(0, N, per):
                blk = [by_id[g] for g in range(b0, min(N, b0 + per)) if g in by_id]
                keys = max(keys, self._expected_keys(blk))
            worst = engine.PanTable.bytes_for(self.k, per, keys)
            # a block's table sits next to the anchors' rows of ITS genomes only (ceil(per/8) bytes per anchor position, every
            # anchor's: each rank probes them all — unless the probe emits the block's columns directly) and the full-width
            # rows this rank holds as a writer: with several passes those of ALL its anchors (they gather bits pass by
            # pass), with one pass those whose writer jobs are still in flight (run_genome_sharded bounds them)
            mine = -(-len(self.anchor_genomes) // max(1, self.world))
            passes = -(-((N + per - 1) // per) // max(1, self.world))
            resident = (mine if passes > 1 else min(mine, 2 * self.writer_jobs(longest * nb * mine) + 1)) * longest * nb
            avail = free - self.HBM_RESERVE - max(2 * longest * nb, resident)
            fits_sparse = worst <= avail
            if fits_sparse:
                cand = (passes / self._block_rate(per, 3.0), (N + per - 1) // per, 0.0, False)
            else:
                # (blocks of one or two genomes that only fit dense have the probe emit their bit columns itself — no narrow
                # rows, one byte per anchor position more for the table: configs[4] on one GPU 3.6 -> 3.2 keys per line, +5 %)
                from .distributed import ShardedAnchoring
                direct = per <= 2 or (engine.COLUMNS_DIRECT and per <= min(8, ShardedAnchoring.DIRECT_MAX_WIDTH))
                room = avail - (0 if direct else anchor_positions * ((per + 7) // 8))
                kpl = keys * 128.0 * 1.02 / room if room > 0 else float("inf")  # (+2 %: the line count is rounded up to a prime)
                kpl = max(kpl, 3.05)
                cand = (passes / self._block_rate(per, kpl), (N + per - 1) // per, round(kpl + 0.05, 1), direct and per <= 2) if (per <= 64 and kpl <= self.BLOCK_KPL_MAX) else None
            if cand is not None and (best is None or cand[0] < best[0] - 1e-9):
                best = cand
            if fits_sparse or nblocks >= N:
                if best is None:  # (nothing fits by this arithmetic: one genome per block at the library's density, as before)
                    best = (0.0, (N + per - 1) // per, 0.0, False)
                self._block_keys_per_line, self._block_direct = best[2], best[3]
                return "genome", best[1]
            nblocks += max(1, self.world)

    # relative probe rate of a pass against a block table of ``per`` genomes at ``kpl`` keys per 128-byte line (8 x 3 Gb, d = 0.05,
    # one MI355X; profiles/r6g*_config5_blocks.txt): G k-mers/s per pass 112 at (1, 3.0); (2, .): 94 at 3.2, 93 at 3.4, 90 at 3.6,
    # 86-89 at 4.0, 80 at 4.5, 61 at 5.5, 45 at 6.2; 45 at (4, 5.8)
    BLOCK_KPL = ((3.0, 1.0), (3.6, 0.955), (4.0, 0.92), (4.5, 0.845), (5.5, 0.65), (6.2, 0.476))
    BLOCK_KPL_MAX = 6.2
    _block_keys_per_line = 0.0  # what plan_sharding chose for the block tables (0: the library's density)
    _block_direct = False       # ... and whether the probe emits the blocks' bit columns itself (no narrow rows)

    @classmethod
    def _block_rate(cls, per: int, kpl: float) -> float:
        pts = cls.BLOCK_KPL
        if kpl <= pts[0][0]:
            r = pts[0][1]
        elif kpl >= pts[-1][0]:
            r = pts[-1][1]
        else:
            r = next(a[1] + (b[1] - a[1]) * (kpl - a[0]) / (b[0] - a[0]) for a, b in zip(pts, pts[1:]) if a[0] <= kpl <= b[0])
        return r * max(1, per) ** -0.27

    # ---- panagram index command (index.py:172-191) ----
    def run(self):
        """Table build, then the anchors in batches: the anchor genomes of a batch share ONE
        co-scheduled launch (homologous regions side by side, table lines shared in L2 — the
        reference runs one thread per anchor FASTA instead, cpp/anchor.cpp:217-223), and host threads
        stream each genome's rows out of HBM into its BGZF files while the next batch is anchored.
        A pangenome whose table does not fit one GPU goes through the genome-sharded mode instead
        (``plan_sharding``)."""
        print("Wrote config.yaml and samples.tsv")
        if self.prepare:
            print("Prepared. Run 'python -m panagram_amd index <dir>' to build the index")
            return
        from concurrent.futures import ThreadPoolExecutor
        import time
        self._run_t0, self._cpu_t0 = time.perf_counter(), time.process_time()
        os.makedirs(self.get_subdir("logs"), exist_ok=True)
        own_group = self._ensure_process_group()
        try:
            self._run_planned(ThreadPoolExecutor)
        finally:
            if own_group:
                self._dist().destroy_process_group()

    def _ensure_process_group(self) -> bool:
        """Under a launcher that set up a rendezvous (torchrun: MASTER_ADDR / MASTER_PORT, RANK, WORLD_SIZE) a run with
        several ranks joins the process group itself — backend "nccl" (= RCCL over xGMI), this rank's GPU — so that the
        ranks can agree on the plan and, in the genome-sharded mode, exchange their bit columns.  Returns whether this
        call created the group (the caller then destroys it).  Without a rendezvous in the environment (ranks run by
        hand, one after the other: the tests) nothing is initialised — the contig-sharded mode needs no group."""
        if self.world <= 1 or "MASTER_ADDR" not in os.environ or int(os.environ.get("WORLD_SIZE", "1")) != self.world:
            return False
        import torch
        import torch.distributed as dist
        if dist.is_initialized():
            return False
        backend = os.environ.get("PG_DIST_BACKEND", "nccl")
        if backend == "nccl":
            torch.cuda.set_device(self.device)
            dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=torch.device("cuda", self.device))
        else:
            dist.init_process_group(backend, rank=self.rank, world_size=self.world)
        return True

    def _run_planned(self, ThreadPoolExecutor):
        import time
        mode, nblocks = self.plan_sharding()
        self._check_plan_agreed((mode, nblocks, self._block_keys_per_line if mode == "genome" else 0.0))
        if mode == "genome":
            from .distributed import run_genome_sharded
            logger.info("genome-sharded mode: %d genome blocks over %d GPU(s)%s", nblocks, self.world,
                        f", block tables at {self._block_keys_per_line:g} keys per line" if self._block_keys_per_line else "")
            self.exchange_stats = {}  # what this rank's exchange moved (bytes_received, passes, chunks): distributed.run_genome_sharded
            run_genome_sharded(self, nblocks, exchange_stats=self.exchange_stats)
            self.close()
            return
        if self.world > 1 and os.environ.get("PG_PARTITION", "pieces") != "genomes":
            # several GPUs, a table that fits one: pieces of homology classes are dealt to the ranks, so that every
            # rank's launch still co-schedules ALL anchor genomes (distributed.run_index_sharded); dealing whole
            # genomes to ranks (PG_PARTITION=genomes, below) leaves a rank's launch without co-scheduling partners —
            # and GPUs idle when there are fewer anchors than GPUs
            from .distributed import run_index_sharded
            run_index_sharded(self, self.rank, self.world, self._barrier())
            self.close()
            return
        mine = self.my_anchor_genomes()
        if not mine:  # more ranks than anchor genomes
            logger.info("rank %d of %d: no anchor genome to write", self.rank, self.world)
            self.close()
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
        with ThreadPoolExecutor(max_workers=self.writer_jobs(payload)) as pool:
            previous = None
            for batch in batches:
                t0 = time.perf_counter()
                job = self._anchor_batch(tbl, batch)
                t1 = time.perf_counter()
                futs = [pool.submit(self.genomes[name].write_from_result, job, gi) for gi, name in enumerate(batch)]
                if previous is not None:  # at most two batches of rows resident
                    self._finish_batch(*previous)
                previous = (job, batch, futs)
                # (where the time of a multi-batch run goes: rows allocated + anchored + tabulated / waiting for the writers of the batch before)
                self.timings["anchor_batches_s"] = self.timings.get("anchor_batches_s", 0.0) + t1 - t0
                self.timings["writers_wait_s"] = self.timings.get("writers_wait_s", 0.0) + time.perf_counter() - t1
            t1 = time.perf_counter()
            if previous is not None:
                self._finish_batch(*previous)
            self.timings["writers_wait_s"] = self.timings.get("writers_wait_s", 0.0) + time.perf_counter() - t1
            self.timings["batches"] = len(batches)
        self.close()

    @staticmethod
    def _dist():
        """torch.distributed when this process is part of an initialised process group, else None (torch is not
        imported for a single process)"""
        import sys
        td = sys.modules.get("torch.distributed")
        return td if td is not None and td.is_available() and td.is_initialized() else None

    def _barrier(self):
        d = self._dist()
        return d.barrier if d is not None and self.world > 1 else None

    def _agreed_free_memory(self) -> int:
        """HBM this process may plan with — THE SAME NUMBER ON EVERY RANK, since ranks that decide differently (one
        replicated, one genome-sharded; or different block counts) would meet in mismatched collectives: the minimum of
        the ranks' free memory when a process group exists, else (several ranks without one) the device's total minus
        a fixed allowance for other tenants, else this device's free memory."""
        free, total = self.context.mem_info()
        if self.world <= 1:
            return free
        d = self._dist()
        if d is None:
            return total - (8 << 30)
        import torch
        dev = self.context.torch_device() if d.get_backend() == "nccl" else torch.device("cpu")
        t = torch.tensor([free], dtype=torch.int64, device=dev)
        d.all_reduce(t, op=d.ReduceOp.MIN)
        return int(t.item())

    def _check_plan_agreed(self, plan) -> None:
        """every rank must have reached the same (mode, nblocks) before any of them enters a collective"""
        d = self._dist()
        if d is None or self.world <= 1:
            return
        got = [None] * self.world
        d.all_gather_object(got, (plan[0], int(plan[1])) + tuple(plan[2:]))
        if any(g != got[0] for g in got):
            raise RuntimeError(f"the ranks planned different sharding modes {got}: pin one with PG_SHARD / PG_GENOME_BLOCKS")

    @staticmethod
    def writer_jobs(payload_bytes: int) -> int:
        return int(os.environ.get("PG_WRITERS", "4" if payload_bytes > (4 << 30) else "2"))

    def writer_of_anchor(self) -> Dict[str, int]:
        """anchor genome -> the rank that writes its directory: dealt longest-first (FASTA size), the same answer
        in every process"""
        from .distributed import plan_shards
        units = [(n, 0, os.path.getsize(self.genomes[n].fasta)) for n in self.anchor_genomes]
        return {u[0]: r for r, sh in enumerate(plan_shards(units, max(1, self.world))) for u in sh}

    def my_anchor_genomes(self) -> List[str]:
        """Multi-GPU (one process per GPU, e.g. under torchrun: RANK / WORLD_SIZE): the table is
        replicated — every rank builds it from all inputs — and the anchor GENOMES are dealt to the
        ranks longest-first (FASTA size); each rank writes the directories of its genomes.  Anchors are
        independent (cpp/anchor.cpp:217-223 runs them as OpenMP iterations), so there is no collective
        and the files do not depend on the GPU count."""
        if self.world <= 1:
            return list(self.anchor_genomes)
        w = self.writer_of_anchor()
        return [n for n in self.anchor_genomes if w[n] == self.rank]

    batch_bytes = 32 << 30  # rows of one batch of anchor genomes held in HBM (bitmap.1 payload bytes)
    # BGZF compression of the bitmaps: a zlib level (host threads), or -2 = on the GPU (k_row_deflate)
    bgzf_level = int(os.environ.get("PG_BGZF_LEVEL", "-2"))

    def _anchor_batch(self, tbl, batch):
        sets = [self.seqset_for(name) for name in batch]
        merged = engine.SeqSet.concat(self.context, sets) if len(sets) > 1 else sets[0]
        for name, ss in zip(batch, sets):
            g = self.genomes[name]
            g.ensure_log()
            for nm, ln in zip(ss.names, ss.lens):
                if int(ln) < tbl.k:
                    g.log.warning(f"Contig {nm} is shorter than k={tbl.k}: 0 k-mers (the reference underflows here)")
            g.log.info("Anchoring Started")
        res = engine.AnchorResult(tbl, merged, colsums=True, **self.result_geometry)
        first = np.cumsum([0] + [len(s.names) for s in sets])
        if len(sets) > 1:  # homologous chromosomes side by side, matched by record id whatever order a FASTA lists them in
            res.coschedule(np.repeat(np.arange(len(sets)), [len(s.names) for s in sets]),
                           contig_class=engine.homology_classes([s.names for s in sets]))
        res.run()
        return dict(res=res, merged=merged if len(sets) > 1 else None,
                    genomes=[self.genomes[name].tabulate(res, int(first[gi]), int(first[gi + 1]), list(merged.names[first[gi]:first[gi + 1]]))
                             for gi, name in enumerate(batch)])

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
        # this genome's own log (logs/anchor.<name>.log.txt when set up): the reference ran one process — and so
        # one basicConfig — per genome; here many genomes share a process, each with a handler of its own
        self.log = logging.getLogger(f"{__name__}.genome.{name}")
        self._log_handler = None
        self._bench_t0 = None  # when this genome's anchoring began (write_benchmark)
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
        """route this genome's messages to ``logfile`` (same line format as the reference's anchor logs)"""
        self.close_log()
        if logfile:
            h = logging.FileHandler(logfile)
            h.setFormatter(logging.Formatter("[ %(asctime)s %(levelname)7s ] %(message)s", datefmt="%Y-%m-%d %H:%M:%S"))
            self.log.addHandler(h)
            self.log.setLevel(logging.INFO)
            self._log_handler = h

    def write_benchmark(self) -> None:
        """logs/anchor.<name>.benchmark.txt in the format of Snakemake's ``benchmark:`` directive — the reference's own timing
        artefact for this step (panagram/workflow/Snakefile:43-44; cpp/Snakefile:43-44): one header line, one row, tab-separated
        ``s  h:m:s  max_rss  max_vms  max_uss  max_pss  io_in  io_out  mean_load  cpu_time`` (seconds; MB; percent).  The
        reference runs one process per anchor; here every anchor of a run shares one process per GPU, so ``s`` is the wall
        time from the moment this genome's anchoring began (its log's "Anchoring Started": the launch it shares with the
        other anchors of its batch) until its files were complete, and the memory / IO / CPU columns are the process's."""
        import resource
        import time
        if not self.index.write_mode:
            return
        t0 = self._bench_t0 if self._bench_t0 is not None else getattr(self.index, "_run_t0", None)
        if t0 is None:
            return
        secs = max(0.0, time.perf_counter() - t0)
        cpu = max(0.0, time.process_time() - getattr(self.index, "_cpu_t0", 0.0))
        rss = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0
        vms = uss = pss = io_in = io_out = float("nan")
        try:
            import psutil
            pr = psutil.Process()
            mi = pr.memory_full_info()
            vms, uss, pss = mi.vms / 2 ** 20, mi.uss / 2 ** 20, getattr(mi, "pss", float("nan")) / 2 ** 20
            io = pr.io_counters()
            io_in, io_out = io.read_chars / 2 ** 20, io.write_chars / 2 ** 20
        except Exception:  # noqa: BLE001 — psutil missing or /proc restricted: the columns stay NaN, as Snakemake writes them then
            pass
        hms = f"{int(secs // 3600)}:{int(secs % 3600 // 60):02d}:{int(secs % 60):02d}"
        os.makedirs(self.index.get_subdir("logs"), exist_ok=True)
        path = os.path.join(self.index.get_subdir("logs"), f"anchor.{self.name}.benchmark.txt")
        fields = [f"{secs:.4f}", hms] + [f"{x:.2f}" for x in (rss, vms, uss, pss, io_in, io_out, 100.0 * cpu / secs if secs > 0 else 0.0, cpu)]
        with open(path + ".tmp", "w") as f:
            f.write("s\th:m:s\tmax_rss\tmax_vms\tmax_uss\tmax_pss\tio_in\tio_out\tmean_load\tcpu_time\n" + "\t".join(fields) + "\n")
        os.replace(path + ".tmp", path)
        self._bench_t0 = None  # a second run of the same object counts from its own start

    def ensure_log(self, started: Optional[float] = None):
        """logs/anchor.<name>.log.txt, unless a log has been set up already.  ``started``: when this genome's anchoring
        began, if that was before this call (a genome assembled from pieces: its launches ran long before its log opens)"""
        if self._bench_t0 is None:
            import time
            self._bench_t0 = time.perf_counter() if started is None else started
        if self._log_handler is None and self.index.write_mode:
            os.makedirs(self.index.get_subdir("logs"), exist_ok=True)
            self.setup_log(os.path.join(self.index.get_subdir("logs"), f"anchor.{self.name}.log.txt"))

    def close_log(self):
        if self._log_handler is not None:
            self.log.removeHandler(self._log_handler)
            self._log_handler.close()
            self._log_handler = None

    def anchor_on_gpu(self, table: engine.PanTable, ss: engine.SeqSet):
        """Enqueue the anchor kernels for a packed FASTA and fetch the small outputs (bins, column
        sums); the bitmap rows stay in HBM for ``write_from_result``."""
        self.ensure_log()
        for nm, ln in zip(ss.names, ss.lens):
            if int(ln) < table.k:
                self.log.warning(f"Contig {nm} is shorter than k={table.k}: 0 k-mers (the reference underflows here)")
        self.log.info("Anchoring Started")
        res = engine.AnchorResult(table, ss, colsums=True, **self.index.result_geometry)
        res.run()
        return dict(res=res, merged=None, genomes=[self.tabulate(res, 0, len(ss.names), list(ss.names))])

    def tabulate(self, res, lo: int, hi: int, names: List[str]):
        """The small outputs of this genome's contigs ``lo..hi-1`` of a finished result — per-contig bins and
        geometry, column sums, per-gene occupancy — as the tuple ``write_from_result`` takes; the bitmap rows
        stay in HBM."""
        self.ensure_log()
        small = res.contigs_small(lo, hi - lo)  # (one call for all contigs; small[i] = what res.download(lo + i, False, False) gives)
        cs = res.contig_colsums(lo, hi - lo).astype(np.int64).sum(axis=0)
        gene_hists = self._tabulate_genes(res, names, small, lo) if self.annotated else None
        return (lo, hi, names, small, cs, gene_hists)

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
                self.log.warning(f"Skipping gene at {chrom}:{s_}-{e_}, coordinates out-of-bounds")
            hist = np.zeros(self.ngenomes + 1, np.int64)
            if ok.any():
                h, _ = res.window_stats(first_contig + ci, st[ok], en[ok], step=1, colsums=False)
                hist = h.sum(axis=0).astype(np.int64)
            out[chrom] = (len(grp), hist)
            self.log.info(f"Annotated {chrom}")
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
        self._write_tables(names, small if isinstance(small, engine.SmallOutputs) else [(b, info) for _, _, b, info in small], cs, gene_hists)
        self.close_log()

    def _bgzf_threads(self) -> int:
        return max(1, min(64, self.index.cores if self.index.cores > 1 else engine.usable_cpus()))

    def _write_tables(self, names, bins_infos, paircount_sums, gene_hists=None):
        N = self.ngenomes
        chr_rows: List[Tuple[str, int, int, int]] = []
        if isinstance(bins_infos, engine.SmallOutputs):
            # the contigs' bins in one array: the text is formatted by the library (two million rows per anchor genome for
            # an assembly of 20 000 contigs; row by row in the interpreter that was most of such a run)
            bins_infos.write_bins_tsv(self.bins_fname, N)
            for ci, chrom in enumerate(names):
                gene_count = gene_hists[chrom][0] if gene_hists and chrom in gene_hists else 0
                chr_rows.append((chrom, ci, int(bins_infos.nkmers[ci]), gene_count))
            if len(names) <= 1000:
                for chrom in names:
                    self.log.info(f"Anchored {chrom}")
            else:  # (a log line per contig of a fragmented assembly is a second of logging per genome)
                self.log.info(f"Anchored {len(names)} contigs ({names[0]} ... {names[-1]})")
        else:
            bins_rows: List[str] = ["chr\tstart" + "".join(f"\t{i}" for i in range(N + 1)) + "\n"]
            for ci, (chrom, (bins, info)) in enumerate(zip(names, bins_infos)):
                starts = np.arange(info["nbins"], dtype=np.int64) * info["binlen"]
                body = np.column_stack([np.full(info["nbins"], ci, np.int64), starts, bins.astype(np.int64)])
                bins_rows.extend("\t".join(map(str, row)) + "\n" for row in body.tolist())
                gene_count = gene_hists[chrom][0] if gene_hists and chrom in gene_hists else 0
                chr_rows.append((chrom, ci, info["nkmers"], gene_count))
                self.log.info(f"Anchored {chrom}")
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
        self.write_benchmark()
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
    ``root/kmc/bitvec{i}`` (KMC1 or KMC2 layout, memory-mapped and imported on the GPU), writes
    ``root/anchor/<name>/{bitmap.1,bitmap.100}.gz(.gzi)``, ``bitsum.bins.tsv``, ``chrs.tsv``.  The reference runs
    its anchors as concurrent OpenMP iterations (cpp/anchor.cpp:217-223); here they share co-scheduled launches
    (batches bounded by the HBM their rows take) and their bitmaps are compressed on the GPU and streamed to the
    files by writer threads — the rows never visit the host uncompressed.  Unlike the reference, a missing or
    ill-formed DB is an error (cpp/anchor.cpp:29 ignores OpenForRA's result)."""
    from concurrent.futures import ThreadPoolExecutor
    ngenomes = int(argv[0])
    root = argv[1]
    pairs = argv[2:]
    if len(pairs) > 2 * ngenomes:
        print(f"Error: expected {ngenomes * 2 + 1} or fewer arguments")
        return 1
    ndbs = (ngenomes + 31) // 32
    ctx = engine.Context(device)
    prefixes = [os.path.join(root, "kmc", f"bitvec{i}") for i in range(ndbs)]
    k = engine.kmc_kmer_length(np.memmap(prefixes[0] + ".kmc_pre", dtype=np.uint8, mode="r"))
    anchors = list(zip(pairs[0::2], pairs[1::2]))
    # (how the table will be probed decides its minimizer window with the key count: one FASTA has no co-scheduling partner)
    tbl = engine.PanTable(ctx, k, ngenomes, coscheduled=max(1, len(anchors)))
    for i, p in enumerate(prefixes):
        tbl.load_kmc_files(i, p)
    nb_row = (ngenomes + 7) // 8
    bgzf_level = int(os.environ.get("PG_BGZF_LEVEL", "-2"))

    def write_anchor(res, name, lo, names):
        adir = os.path.join(root, "anchor", name)
        os.makedirs(adir, exist_ok=True)
        for step in (1, 100):
            res.write_bgzf(step, os.path.join(adir, f"bitmap.{step}.gz"), os.path.join(adir, f"bitmap.{step}.gzi"),
                           level=bgzf_level, threads=max(1, engine.usable_cpus() // 2), first_contig=lo, ncontigs=len(names))
        with open(os.path.join(adir, "bitsum.bins.tsv"), "w") as fb, open(os.path.join(adir, "chrs.tsv"), "w") as fc:
            fb.write("chr\tstart" + "".join(f"\t{i}" for i in range(ngenomes + 1)) + "\n")
            fc.write("name\tid\tsize\tgene_count\n")
            for ci, chrom in enumerate(names):
                _, _, bins, info = res.download(lo + ci, want_bitmap1=False, want_bitmap100=False)
                for b in range(info["nbins"]):
                    fb.write(f"{ci}\t{b * info['binlen']}" + "".join(f"\t{int(c)}" for c in bins[b]) + "\n")
                fc.write(f"{chrom}\t{ci}\t{info['nkmers']}\t0\n")

    free = ctx.mem_info()[0]
    batch_bytes = int(min(Index.batch_bytes, max(1 << 30, (free - (6 << 30)) / 2.3)))
    with ThreadPoolExecutor(max_workers=2) as pool:
        previous = None  # (result, merged seqset, seqsets, futures) of the batch being written
        i = 0
        while i < len(anchors):
            batch, sets, rows = [], [], 0
            while i < len(anchors):
                name, fasta = anchors[i]
                ss = engine.SeqSet.from_fasta(ctx, fasta)
                need = int(ss.lens.sum()) * nb_row
                if batch and rows + need > batch_bytes:
                    ss.close()
                    break
                print(f"Anchoring {name} {fasta}")
                batch.append(name)
                sets.append(ss)
                rows += need
                i += 1
            merged = engine.SeqSet.concat(ctx, sets) if len(sets) > 1 else sets[0]
            res = engine.AnchorResult(tbl, merged, colsums=False)
            if len(sets) > 1:
                res.coschedule(np.repeat(np.arange(len(sets)), [len(x.names) for x in sets]),
                               contig_class=engine.homology_classes([x.names for x in sets]))
            res.run()
            first = np.cumsum([0] + [len(x.names) for x in sets])
            futs = [pool.submit(write_anchor, res, nm, int(first[j]), list(sets[j].names)) for j, nm in enumerate(batch)]
            if previous is not None:
                _join_cli_batch(*previous)
            previous = (res, merged if len(sets) > 1 else None, sets, futs)
        if previous is not None:
            _join_cli_batch(*previous)
    tbl.close()
    ctx.close()
    return 0


def _join_cli_batch(res, merged, sets, futs):
    try:
        for f in futs:
            f.result()
    finally:
        res.close()
        if merged is not None:
            merged.close()
        for x in sets:
            x.close()
