"""Thin object layer over the C-ABI: Context, PanTable, SeqSet, AnchorResult, BgzfWriter.

All compute happens in libpanagram_hip.so on the GPU; this file moves pointers.
numpy arrays are the host buffers the ABI asks the caller to own.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
import weakref
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import PG_ANCHOR_COLSUMS, PG_ANCHOR_COLUMNS_ONLY, PG_ANCHOR_ROWS_ONLY, PanagramHipError, check  # noqa: F401


def usable_cpus() -> int:
    """Host cores this process may actually use: the scheduler affinity, capped by the cgroup CPU
    quota (a container can see 256 hardware threads and be allowed 16 cores' worth of time)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def kmc_kmer_length(pre) -> int:
    """k of a KMC database from the image of its .kmc_pre file (either layout)"""
    v = _bytes_view(pre)
    k = C.c_uint32()
    check(_lib.load().pg_kmc_kmer_length(_ptr(v), v.size, C.byref(k)))
    return int(k.value)


# k_probe can emit a one-genome block's bit columns itself (ROWMODE 3: no row buffer at all).  Off by default: since the
# probe's round-2 diet the rows + k_cols_extract_b1 route is the faster one (one GPU as rank 0 of 8, 8 x 10^8 positions:
# 6.50 ms per step against 7.15, tools/ab_sharded.sh); PG_COLUMNS_DIRECT=1 turns it on where the row buffer's HBM matters.
COLUMNS_DIRECT = os.environ.get("PG_COLUMNS_DIRECT", "0") not in ("", "0")


_ADOPT_LOCK = threading.Lock()


def tile_positions() -> int:
    """k-mer positions per tile (launch unit; a bit-column block holds tile_positions() // 8 bytes per tile and genome:
    size buffers with AnchorResult.columns_bytes, never with a constant)"""
    return int(_lib.load().pg_tile_positions())


class _Owner:
    """Handles are destroyed children first (the C-ABI's rule): a parent remembers its live
    children weakly and closes them before itself, so that garbage collection — which finalises
    a dropped object graph in no particular order — can never free a table under a result."""

    def _adopt(self, child) -> None:
        with _ADOPT_LOCK:  # (sequence sets are parsed by several host threads at once: Index.load_inputs)
            if not hasattr(self, "_children"):
                self._children = []
            self._children = [w for w in self._children if w() is not None]
            self._children.append(weakref.ref(child))

    def _close_children(self) -> None:
        for w in getattr(self, "_children", []):
            c = w()
            if c is not None:
                c.close()
        self._children = []


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _bytes_view(seq) -> np.ndarray:
    """bytes / bytearray / str / uint8 ndarray -> contiguous uint8 ndarray (no copy when possible)."""
    if isinstance(seq, str):
        seq = seq.encode("latin-1")
    if isinstance(seq, np.ndarray):
        return np.ascontiguousarray(seq, dtype=np.uint8)
    return np.frombuffer(seq, dtype=np.uint8)


class Context(_Owner):
    """One per GPU / per rank."""

    def __init__(self, device: int = 0):
        self._lib = _lib.load()
        h = C.c_void_p()
        check(self._lib.pg_ctx_create(device, C.byref(h)))
        self._h = h
        self.device = device

    def set_stream(self, hip_stream: Optional[int]) -> None:
        """Adopt an external hipStream_t handle (0 = HIP's default stream, i.e. torch's
        default stream); ``None`` goes back to the context's own stream."""
        if hip_stream is None:
            check(self._lib.pg_ctx_set_stream(self._h, None, 1))
        else:
            check(self._lib.pg_ctx_set_stream(self._h, C.c_void_p(hip_stream), 0))

    def synchronize(self) -> None:
        check(self._lib.pg_ctx_synchronize(self._h))

    def mem_info(self) -> Tuple[int, int]:
        """(free, total) device memory in bytes"""
        f, t = C.c_uint64(), C.c_uint64()
        check(self._lib.pg_ctx_mem_info(self._h, C.byref(f), C.byref(t)))
        return int(f.value), int(t.value)

    def trim(self) -> None:
        """give back device memory kept for reuse (row buffers of closed results)"""
        check(self._lib.pg_ctx_trim(self._h))

    def host_buffer(self, nbytes: int) -> "HostBuffer":
        """page-locked host memory (pg_host_alloc): a numpy uint8 array that goes to the device by DMA"""
        return HostBuffer(self, nbytes)

    def torch_device(self):
        """the torch device the context's buffers live on (torch only carries the collectives' buffers)"""
        import torch
        return torch.device("cuda", self.device)

    def close(self) -> None:
        if self._h:
            self._close_children()
            self._lib.pg_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceBuffer:
    """a plain (zeroed) device buffer of the library's own — what the single-process genome-sharded pipeline exchanges
    its bit columns through (several ranks use torch tensors: the collective takes those)"""

    def __init__(self, ctx: Context, nbytes: int):
        self.ctx, self._lib, self.nbytes = ctx, ctx._lib, int(nbytes)
        p = C.c_void_p()
        check(self._lib.pg_device_alloc(ctx._h, self.nbytes, C.byref(p)))
        self._p = p

    def data_ptr(self) -> int:
        return int(self._p.value)

    def zero(self, nbytes: Optional[int] = None) -> None:
        check(self._lib.pg_device_memset(self.ctx._h, self._p, 0, self.nbytes if nbytes is None else nbytes))

    def close(self) -> None:
        if self._p:
            self._lib.pg_device_free(self.ctx._h, self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HostBuffer:
    """``nbytes`` of page-locked host memory as ``.array`` (numpy uint8); ``close()`` gives it back (not before every view of
    ``.array`` is out of use)."""

    def __init__(self, ctx: Context, nbytes: int):
        import threading
        self.ctx, self.nbytes = ctx, int(nbytes)
        self._lock = threading.Lock()  # (a pool may give its buffers back from a helper thread while the context is being closed)
        p = C.c_void_p()
        check(ctx._lib.pg_host_alloc(ctx._h, self.nbytes, C.byref(p)))
        self._p = p
        self.array = np.ctypeslib.as_array((C.c_uint8 * max(1, self.nbytes)).from_address(p.value))[:self.nbytes]
        ctx._adopt(self)  # (closed with the context at the latest)

    def close(self) -> None:
        with self._lock:
            p, self._p = self._p, None
            if p is not None and p.value and self.ctx._h:
                self.array = None
                check(self.ctx._lib.pg_host_free(self.ctx._h, p))

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 — interpreter shutdown / a context that is gone already
            pass


class SeqSet(_Owner):
    """The contigs of one FASTA, 2-bit packed in HBM."""

    def __init__(self, ctx: Context, lens: Sequence[int]):
        self.ctx = ctx
        self._lib = ctx._lib
        self.lens = np.asarray(lens, dtype=np.uint64)
        self.names = [""] * len(self.lens)
        h = C.c_void_p()
        check(self._lib.pg_seqset_create(ctx._h, len(self.lens), _ptr(self.lens), C.byref(h)))
        self._h = h
        ctx._adopt(self)

    @classmethod
    def from_host(cls, ctx: Context, seqs: Sequence) -> "SeqSet":
        views = [_bytes_view(s) for s in seqs]
        ss = cls(ctx, [len(v) for v in views])
        for i, v in enumerate(views):
            ss.load_host(i, v)
        return ss

    @classmethod
    def from_fasta(cls, ctx: Context, source) -> "SeqSet":
        """FASTA file (path; ``.gz``/``.bgz`` through gzip) or FASTA text (bytes / uint8 array):
        the GPU strips the line breaks and packs; ``names`` / ``lens`` describe the records."""
        if isinstance(source, (str, os.PathLike)):
            path = os.fspath(source)
            if path.endswith((".gz", ".bgz")):
                import gzip
                with gzip.open(path, "rb") as f:
                    text = np.frombuffer(f.read(), dtype=np.uint8)
            else:
                text = np.fromfile(path, dtype=np.uint8)
        else:
            text = _bytes_view(source)
        ss = cls.__new__(cls)
        ss.ctx, ss._lib = ctx, ctx._lib
        h = C.c_void_p()
        check(ss._lib.pg_seqset_from_fasta(ctx._h, _ptr(text), len(text), C.byref(h)))
        ss._h = h
        ctx._adopt(ss)
        n = int(ss._lib.pg_seqset_ncontigs(h))
        lens, need = np.zeros(n, np.uint64), C.c_uint64()
        check(ss._lib.pg_seqset_describe(h, _ptr(lens), None, 0, C.byref(need)))  # (all contigs in two calls, not one each)
        buf = np.zeros(max(1, need.value), np.uint8)
        check(ss._lib.pg_seqset_describe(h, None, _ptr(buf), buf.size, None))
        names = buf[:need.value].tobytes().decode("latin-1").split("\0")[:n] if n else []
        ss.names, ss.lens = names, lens
        return ss

    @classmethod
    def concat(cls, ctx: Context, sets: Sequence["SeqSet"]) -> "SeqSet":
        """One seqset with the contigs of all ``sets`` in order (device copy), for a co-scheduled
        result over several anchor genomes."""
        ss = cls.__new__(cls)
        ss.ctx, ss._lib = ctx, ctx._lib
        arr = (C.c_void_p * len(sets))(*[x._h for x in sets])
        h = C.c_void_p()
        check(ss._lib.pg_seqset_concat(ctx._h, arr, len(sets), C.byref(h)))
        ss._h = h
        ctx._adopt(ss)
        ss.names = [n for x in sets for n in x.names]
        ss.lens = np.concatenate([x.lens for x in sets]) if sets else np.zeros(0, np.uint64)
        return ss

    @classmethod
    def concat_ranges(cls, ctx: Context, parts: Sequence[Tuple["SeqSet", int, int]]) -> "SeqSet":
        """One seqset from contig ranges ``(set, first contig, count)``, in order (a set may appear several times)."""
        ss = cls.__new__(cls)
        ss.ctx, ss._lib = ctx, ctx._lib
        n = len(parts)
        arr = (C.c_void_p * n)(*[p[0]._h for p in parts])
        first = np.array([p[1] for p in parts], np.uint32)
        cnt = np.array([p[2] for p in parts], np.uint32)
        h = C.c_void_p()
        check(ss._lib.pg_seqset_concat_ranges(ctx._h, arr, _ptr(first), _ptr(cnt), n, C.byref(h)))
        ss._h = h
        ctx._adopt(ss)
        ss.names = [nm for s_, f, c in parts for nm in s_.names[f:f + c]]
        ss.lens = (np.concatenate([np.asarray(s_.lens[f:f + c], np.uint64) for s_, f, c in parts])
                   if parts else np.zeros(0, np.uint64))
        return ss

    def slice(self, pieces: Sequence[Tuple[int, int, int]]) -> "SeqSet":
        """A seqset of pieces ``(contig, start base, bases)`` of this one's contigs (starts: multiples of 32), packed
        planes copied on the device: the contig-sharded multi-GPU mode's chunks of long chromosomes.  A piece that
        carries ``k - 1`` bases of overlap holds exactly the k-mer positions ``[start, start + bases - k + 1)`` of its
        contig (cpp/anchor.cpp:127 cuts its chunks the same way)."""
        ss = SeqSet.__new__(SeqSet)
        ss.ctx, ss._lib = self.ctx, self._lib
        n = len(pieces)
        contig = np.array([p[0] for p in pieces], np.uint32)
        start = np.array([p[1] for p in pieces], np.uint64)
        lens = np.array([p[2] for p in pieces], np.uint64)
        h = C.c_void_p()
        check(self._lib.pg_seqset_slice(self.ctx._h, self._h, n, _ptr(contig), _ptr(start), _ptr(lens), C.byref(h)))
        ss._h = h
        self.ctx._adopt(ss)
        ss.names = [f"{self.names[c]}:{s0}" for c, s0, _ in pieces]
        ss.lens = lens
        return ss

    def load_host(self, idx: int, seq) -> None:
        v = _bytes_view(seq)
        check(self._lib.pg_seqset_load_host(self._h, idx, _ptr(v), len(v)))

    def unpack(self, idx: int) -> bytes:
        """contig idx as the kernels see it: ACGT upper case, N for every non-ACGT byte"""
        out = np.empty(int(self.lens[idx]), np.uint8)
        check(self._lib.pg_seqset_unpack(self._h, idx, _ptr(out)))
        return out.tobytes()

    def load_dev(self, idx: int, dev_ptr: int, length: int) -> None:
        check(self._lib.pg_seqset_load_dev(self._h, idx, C.c_void_p(dev_ptr), length))

    def total_kmers(self, k: int) -> int:
        return int(self._lib.pg_seqset_total_kmers(self._h, k))

    def close(self) -> None:
        if self._h:
            self._close_children()
            self._lib.pg_seqset_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class KmerSketch:
    """Distinct canonical k-mers of a set of inputs before any table exists (HyperLogLog on the GPU,
    standard error 0.4 %): ``PanTable(..., expected_keys=sketch.estimate())`` is then sized once."""

    def __init__(self, ctx: Context, k: int):
        self.ctx, self._lib, self.k = ctx, ctx._lib, k
        h = C.c_void_p()
        check(self._lib.pg_sketch_create(ctx._h, k, C.byref(h)))
        self._h = h
        ctx._adopt(self)

    def add(self, seqs: SeqSet) -> None:
        check(self._lib.pg_sketch_add_seqset(self._h, seqs._h))

    def estimate(self) -> int:
        n = C.c_uint64()
        check(self._lib.pg_sketch_estimate(self._h, C.byref(n)))
        return int(n.value)

    def registers(self) -> np.ndarray:
        out = np.empty(65536, np.uint8)
        check(self._lib.pg_sketch_registers(self._h, _ptr(out)))
        return out

    def reset(self) -> None:
        check(self._lib.pg_sketch_reset(self._h))

    @staticmethod
    def estimate_registers(regs: np.ndarray) -> int:
        """estimate for register values held on the host — e.g. the register-wise maximum of several sketches
        (= the sketch of the union of their inputs)"""
        regs = np.ascontiguousarray(regs, np.uint8)
        if regs.size != 65536:
            raise ValueError("a sketch has 65536 registers")
        n = C.c_uint64()
        check(_lib.load().pg_sketch_estimate_registers(_ptr(regs), C.byref(n)))
        return int(n.value)

    def close(self) -> None:
        if self._h:
            self._lib.pg_sketch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PanTable(_Owner):
    """GPU-resident k-mer -> genome-mask table (replaces kmc/bitvec{i})."""

    def __init__(self, ctx: Context, k: int, ngenomes: int, expected_keys: int = 0, coscheduled: int = 0, keys_per_line: float = 0.0):
        """``coscheduled``: how many anchor genomes one probe launch will anchor side by side against this table (0: not
        known = several; 1: one genome per launch, no co-scheduling partner) — it decides the minimizer window with the
        key count (pg_table_set_coscheduled, include/panagram_hip.h).  ``keys_per_line`` (0: the library's 3): a denser
        table for a known ``expected_keys`` — pg_table_create_dense: the genome-sharded mode's block tables"""
        self.ctx = ctx
        self._lib = ctx._lib
        self.k, self.ngenomes = k, ngenomes
        self.nbytes = (ngenomes + 7) // 8
        self.ndbs = (ngenomes + 31) // 32
        h = C.c_void_p()
        if keys_per_line:
            check(self._lib.pg_table_create_dense(ctx._h, k, ngenomes, expected_keys, float(keys_per_line), C.byref(h)))
        else:
            check(self._lib.pg_table_create(ctx._h, k, ngenomes, expected_keys, C.byref(h)))
        self._h = h
        ctx._adopt(self)
        if coscheduled:
            self.set_coscheduled(coscheduled)

    # Densities a table may be created at when HBM is plentiful (round 6): the library's own 3 keys per 128-byte line date from
    # 80-GB parts; at 1.5 keys per line fewer keys sit outside their home line (8 x 100 Mb: 15.5 % -> 9.5 %), the overflow drain
    # of a tile is shorter and k_probe gains 3 % at 8 genomes, 5-6 % at 27-64 (profiles/r6k2_density_sweep.txt) for twice the table
    # bytes — 21 GB instead of 10 of an MI355X's 288.
    # Not for repeat-rich genomes: half of every genome in transposable-element families (bench.py's plant-like leg) runs 5 %
    # SLOWER against the sparser table (profiles/r6m_roomy_robustness.txt) — the lines of a repeat family are the ones that are
    # used again and again, and at half the density they are twice as many for the caches to hold.  ``distinct_fraction`` =
    # distinct k-mers / k-mer positions of ONE genome (its sketch) tells the two apart.
    ROOMY_DENSITIES = (1.25, 1.5, 2.0, 2.5)  # (1.25 against 1.5 at configs[1]: 3.154 against 3.206 ms, 1.0 no better: profiles/r6s_density_piece.txt)
    ROOMY_SHARE = 0.6       # of what is free beside ``other_bytes`` and the reserve
    ROOMY_RESERVE = 8 << 30
    ROOMY_MIN_DISTINCT = 0.85

    @classmethod
    def roomy_density(cls, ctx: Context, k: int, ngenomes: int, expected_keys: int, other_bytes: int = 0,
                      distinct_fraction: Optional[float] = None) -> float:
        """keys per line (0: the library's default) for a table of ``expected_keys`` that has ``other_bytes`` of HBM to leave
        alone (the rows it will be anchored into, buffers still to come): the sparsest of ROOMY_DENSITIES that fits into
        ROOMY_SHARE of the rest (more than 64 genomes, inline / split lines: the same load, 65 / 96 x 10 Mb -4 % / -20 % probe time,
        profiles/r6q_wide_density.txt); tables of unknown size and of repeat-rich genomes (``distinct_fraction`` of one genome
        below ROOMY_MIN_DISTINCT) keep the default"""
        if not expected_keys or os.environ.get("PG_TABLE_ROOMY", "1") in ("0", "") or os.environ.get("PG_TABLE_KEYS_PER_LINE"):
            return 0.0
        if distinct_fraction is not None and distinct_fraction < cls.ROOMY_MIN_DISTINCT:
            return 0.0
        room = (ctx.mem_info()[0] - int(other_bytes) - cls.ROOMY_RESERVE) * cls.ROOMY_SHARE
        for kpl in cls.ROOMY_DENSITIES:
            if cls.bytes_for(k, ngenomes, expected_keys, kpl) <= room:
                return kpl
        return 0.0

    def set_coscheduled(self, anchors: int) -> None:
        """tell an EMPTY table how it will be probed (see __init__)"""
        check(self._lib.pg_table_set_coscheduled(self._h, int(anchors)))

    @staticmethod
    def bytes_for(k: int, ngenomes: int, expected_keys: int, keys_per_line: float = 0.0) -> int:
        """device bytes a table created for that many keys (at that density; 0: the library's) takes"""
        b = C.c_uint64()
        if keys_per_line:
            check(_lib.load().pg_table_bytes_for_dense(k, ngenomes, expected_keys, float(keys_per_line), C.byref(b)))
        else:
            check(_lib.load().pg_table_bytes_for(k, ngenomes, expected_keys, C.byref(b)))
        return int(b.value)

    def insert_seqset(self, genome_idx: int, seqs: SeqSet, min_count: int = 1) -> None:
        """OR genome ``genome_idx``'s bit into every canonical k-mer of ``seqs`` that occurs at least
        ``min_count`` times in it (kmc -ci<min_count>; 2 for read sets)"""
        if min_count > 1:
            check(self._lib.pg_table_insert_seqset_min(self._h, genome_idx, seqs._h, min_count))
            return
        check(self._lib.pg_table_insert_seqset(self._h, genome_idx, seqs._h))

    def update_seqset(self, genome_idx: int, seqs: SeqSet) -> None:
        """OR genome ``genome_idx``'s bit into the k-mers of ``seqs`` that the table ALREADY holds; no key is added
        (a table of the anchors' k-mers only answers the anchor step like the table of all genomes)"""
        check(self._lib.pg_table_update_seqset(self._h, genome_idx, seqs._h))

    def clear(self) -> None:
        """empty the table, keeping its allocation (the next genome block is built in the same memory)"""
        check(self._lib.pg_table_clear(self._h))

    def insert_keys(self, db_idx: int, keys: np.ndarray, counters: np.ndarray) -> None:
        keys = np.ascontiguousarray(keys, np.uint64)
        counters = np.ascontiguousarray(counters, np.uint32)
        if len(keys) != len(counters):
            raise ValueError("keys and counters differ in length")
        check(self._lib.pg_table_insert_keys(self._h, db_idx, _ptr(keys), _ptr(counters), len(keys)))

    def load_kmc1(self, db_idx: int, pre: bytes, suf: bytes) -> None:
        """images of X.kmc_pre / X.kmc_suf (KMC1 or KMC2 layout) as 32-genome group ``db_idx``"""
        check(self._lib.pg_table_load_kmc(self._h, db_idx, pre, len(pre), suf, len(suf)))

    load_kmc = load_kmc1

    def load_kmc_files(self, db_idx: int, prefix: str) -> None:
        """``prefix.kmc_pre`` / ``prefix.kmc_suf`` as 32-genome group ``db_idx`` (CKMCFile::OpenForRA,
        cpp/anchor.cpp:29): the files are mapped, not read into Python"""
        pre = np.memmap(prefix + ".kmc_pre", dtype=np.uint8, mode="r")
        suf = np.memmap(prefix + ".kmc_suf", dtype=np.uint8, mode="r")
        check(self._lib.pg_table_load_kmc(self._h, db_idx, _ptr(pre), pre.size, _ptr(suf), suf.size))

    def stats(self) -> dict:
        v = [C.c_uint64() for _ in range(4)]
        check(self._lib.pg_table_stats(self._h, *[C.byref(x) for x in v]))
        return dict(nkeys=v[0].value, nslots=v[1].value, nbuckets=v[2].value, bytes=v[3].value)

    def rehash(self, keys_per_bucket: float) -> None:
        check(self._lib.pg_table_rehash(self._h, keys_per_bucket))

    def spill(self):
        """(fraction of keys outside their home line, slots per line) as of the last rehash()"""
        f, n = C.c_double(), C.c_uint32()
        check(self._lib.pg_table_spill(self._h, C.byref(f), C.byref(n)))
        return f.value, n.value

    def measure_spill(self) -> float:
        """fraction of keys outside their minimizer's home line, measured now (one pass over the table)"""
        f = C.c_double()
        check(self._lib.pg_table_measure_spill(self._h, C.byref(f)))
        return f.value

    @property
    def minimizer(self) -> int:
        """minimizer length m the table places k-mers by (0 = the k-mer itself)"""
        return int(self._lib.pg_table_minimizer(self._h))

    def set_minimizer(self, m: int) -> None:
        """pin m on an empty table (tuning knob; see include/panagram_hip.h)"""
        check(self._lib.pg_table_set_minimizer(self._h, int(m)))

    def export(self, db_idx: int) -> Tuple[np.ndarray, np.ndarray]:
        n = C.c_uint64()
        check(self._lib.pg_table_export(self._h, db_idx, None, None, 0, C.byref(n)))
        keys = np.empty(n.value, np.uint64)
        vals = np.empty(n.value, np.uint32)
        if n.value:
            check(self._lib.pg_table_export(self._h, db_idx, _ptr(keys), _ptr(vals), n.value, C.byref(n)))
        return keys, vals

    def counters_for_read(self, db_idx: int, seq) -> np.ndarray:
        """GetCountersForRead equivalent (cpp/anchor.cpp:148, index.py:934-935)."""
        v = _bytes_view(seq)
        n = max(0, len(v) - self.k + 1)
        out = np.zeros(n, np.uint32)
        check(self._lib.pg_counters_for_read(self._h, db_idx, _ptr(v), len(v), _ptr(out)))
        return out

    def anchor_contig(self, seq, colsums: bool = True):
        """One-shot: rows[nkmers,nbytes], rows100, bins[nbins,N+1], colsums[N] or None."""
        v = _bytes_view(seq)
        nk = max(0, len(v) - self.k + 1)
        binlen = 200000
        if nk // binlen < 100:
            binlen = nk // 100
        binlen = max(binlen, 1)
        nbins = (nk + binlen - 1) // binlen
        rows = np.zeros((nk, self.nbytes), np.uint8)
        rows100 = np.zeros(((nk + 99) // 100, self.nbytes), np.uint8)
        bins = np.zeros((nbins, self.ngenomes + 1), np.uint32)
        cs = np.zeros(self.ngenomes, np.uint64) if colsums else None
        nkm = C.c_uint64()
        check(self._lib.pg_anchor_contig(self._h, _ptr(v), len(v), _ptr(rows), _ptr(rows100), _ptr(bins),
                                         _ptr(cs), C.byref(nkm)))
        assert nkm.value == nk
        return rows, rows100, bins, cs

    def close(self) -> None:
        if self._h:
            self._close_children()
            self._lib.pg_table_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def homology_classes(name_lists: Sequence[Sequence[str]]) -> np.ndarray:
    """``contig_class`` for ``AnchorResult.coschedule`` from the genomes' record ids: contigs with the same id (the same
    chromosome name in different FASTAs) form one class, classes numbered in order of first appearance.  A genome
    fewer than half of whose ids occur in another genome — assemblies usually carry per-assembly accessions (CM0xxxx.1,
    another chr naming) — is paired BY POSITION instead: its i-th contig joins the class of the first genome's i-th
    contig (chromosome-level assemblies of one species list their chromosomes in the same order), so that the
    co-schedule still interleaves the genomes instead of running them one after the other.  Within an id-matched
    genome a contig whose id nobody shares (an extra scaffold) is a class of its own."""
    from collections import Counter
    occ = Counter(nm for names in name_lists for nm in set(names) if nm)
    by_id = [len(name_lists) > 1 and len(names) > 0 and 2 * sum(1 for nm in names if nm and occ[nm] > 1) >= len(names)
             for names in name_lists]
    seen, out, first = {}, [], None  # first: classes of the first genome's contigs, by position
    for g, names in enumerate(name_lists):
        mine = []
        for i, nm in enumerate(names):
            if by_id[g]:
                key = ("id", nm) if (nm and occ[nm] > 1) else ("own", g, i)
            elif first is not None and i < len(first):
                mine.append(first[i])
                continue
            else:
                key = ("own", g, i)
            mine.append(seen.setdefault(key, len(seen)))
        if first is None:
            first = mine
        out += mine
    return np.asarray(out, np.uint32)


class AnchorResult:
    """Device-resident outputs of anchoring one SeqSet against one PanTable."""

    def __init__(self, table: PanTable, seqs: SeqSet, colsums: bool = True, rows_only: bool = False,
                 lowres_step: int = 100, max_bin_len: int = 200000, min_bin_count: int = 100, columns_only: bool = False):
        """``lowres_step`` / ``max_bin_len`` / ``min_bin_count``: the Python path's parameters
        (index.py:101-106, 1169-1172); the defaults are what cpp/anchor.cpp hard-codes."""
        self.table, self.seqs, self.ctx = table, seqs, table.ctx
        self.ngenomes, self.nbytes, self.lowres_step = table.ngenomes, table.nbytes, lowres_step
        self._lib = table._lib
        h = C.c_void_p()
        self.flags = (PG_ANCHOR_COLSUMS if colsums else 0) | (PG_ANCHOR_ROWS_ONLY if rows_only or columns_only else 0) | \
                     (PG_ANCHOR_COLUMNS_ONLY if columns_only else 0)
        check(self._lib.pg_result_create_ex(table._h, seqs._h, self.flags, lowres_step, max_bin_len, min_bin_count,
                                            C.byref(h)))
        self._h = h
        table._adopt(self)   # a result dies before its table and before its sequences
        seqs._adopt(self)

    @classmethod
    def rows_container(cls, ctx: Context, k: int, ngenomes: int, seqs: SeqSet, colsums: bool = True,
                       lowres_step: int = 100, max_bin_len: int = 200000, min_bin_count: int = 100) -> "AnchorResult":
        """A result without a table: ``ngenomes``-wide rows over ``seqs`` (zeroed), filled by
        ``merge_columns_range`` and finished by ``rows_epilogue`` — the writer's side of the genome-sharded mode."""
        r = cls.__new__(cls)
        r.table, r.seqs, r.ctx = None, seqs, ctx
        r.ngenomes, r.nbytes, r.lowres_step = ngenomes, (ngenomes + 7) // 8, lowres_step
        r._lib = ctx._lib
        r.flags = (PG_ANCHOR_COLSUMS if colsums else 0) | PG_ANCHOR_ROWS_ONLY
        h = C.c_void_p()
        check(r._lib.pg_result_create_rows(ctx._h, k, ngenomes, seqs._h, r.flags, lowres_step, max_bin_len,
                                           min_bin_count, C.byref(h)))
        r._h = h
        seqs._adopt(r)
        return r

    def coschedule(self, contig_group, piece_tiles: int = 0, contig_class=None) -> None:
        """Interleave the tiles of several anchor genomes (``contig_group[c]`` = genome of contig c;
        None: launch order) so that homologous regions share their table lines in L2.  ``contig_class[c]``:
        homology class of contig c (e.g. the chromosome) when the genomes list their contigs in different orders."""
        if contig_group is None:
            check(self._lib.pg_result_coschedule(self._h, None, 0))
            return
        grp = np.ascontiguousarray(contig_group, np.uint32)
        if len(grp) != len(self.seqs.lens):
            raise ValueError("contig_group needs one entry per contig")
        if contig_class is None:
            check(self._lib.pg_result_coschedule(self._h, _ptr(grp), piece_tiles))
            return
        cls = np.ascontiguousarray(contig_class, np.uint32)
        if len(cls) != len(grp):
            raise ValueError("contig_class needs one entry per contig")
        check(self._lib.pg_result_coschedule_classes(self._h, _ptr(grp), _ptr(cls), piece_tiles))


    def coschedule_ranges(self, contig_group, range_first_contig, piece_tiles: int = 0) -> None:
        """``coschedule`` with the contigs cut into consecutive ranges (starting at the given contigs, the first at 0)
        that are scheduled independently, so that ``run_range`` over whole ranges follows the schedule"""
        grp = np.ascontiguousarray(contig_group, np.uint32)
        rf = np.ascontiguousarray(range_first_contig, np.uint32)
        if len(grp) != len(self.seqs.lens):
            raise ValueError("contig_group needs one entry per contig")
        check(self._lib.pg_result_coschedule_ranges(self._h, _ptr(grp), piece_tiles, _ptr(rf), len(rf)))

    def contig_colsums(self, idx: int = 0, ncontigs: Optional[int] = None) -> np.ndarray:
        n = len(self.seqs.lens) - idx if ncontigs is None else ncontigs
        out = np.zeros((n, self.ngenomes), np.uint64)
        check(self._lib.pg_result_contig_colsums(self._h, idx, n, _ptr(out)))
        return out

    def run(self) -> None:
        """Enqueue the anchor kernels (asynchronous)."""
        check(self._lib.pg_anchor_run(self._h))

    def run_range(self, first_contig: int, ncontigs: int) -> None:
        """rows-only results: probe a contig range only (asynchronous)"""
        check(self._lib.pg_anchor_run_range(self._h, first_contig, ncontigs))

    def columns_direct(self, width: int) -> bool:
        """can ``run_columns_range`` emit ``width``-genome columns straight from the probe for this table?"""
        return bool(self._lib.pg_result_columns_direct(self._h, width))

    def run_columns_range(self, first_contig: int, ncontigs: int, width: int, dev_ptr: int) -> None:
        """probe a contig range and write its compact bit columns to ``dev_ptr`` — no rows in between (async)"""
        check(self._lib.pg_anchor_run_columns_range(self._h, first_contig, ncontigs, width, C.c_void_p(dev_ptr)))

    def timing(self):
        """(probe_ms, epilogue_ms) of the last run(), from HIP events on the context's stream."""
        a, b = C.c_float(), C.c_float()
        check(self._lib.pg_result_timing(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def timing_reset(self) -> None:
        check(self._lib.pg_result_timing_reset(self._h))

    def timing_mean(self):
        """(mean probe_ms, mean epilogue_ms, runs) over every run() since ``timing_reset()``"""
        a, b, n = C.c_double(), C.c_double(), C.c_uint32()
        check(self._lib.pg_result_timing_mean(self._h, C.byref(a), C.byref(b), C.byref(n)))
        return a.value, b.value, int(n.value)

    def fused_runs(self) -> int:
        """whole runs of this result whose statistics were computed inside k_probe (tile by tile, from rows still in the
        cache) instead of by the pass over every row — pg_result_fused_runs; only with PG_FUSE_STATS=1 (an experiment: slower)"""
        n = C.c_uint32()
        check(self._lib.pg_result_fused_runs(self._h, C.byref(n)))
        return int(n.value)

    def rows_epilogue(self) -> None:
        """bitmap.100 / bins / column sums from the (combined) rows in the device buffer (async)."""
        check(self._lib.pg_rows_epilogue(self._h))

    def columns_bytes(self, width: int) -> int:
        return int(self._lib.pg_result_columns_bytes(self._h, width))

    def extract_columns(self, g0: int, width: int, dev_ptr: int) -> None:
        """compact bit columns of genomes [g0, g0+width) of the rows -> device buffer (async)"""
        check(self._lib.pg_result_extract_columns(self._h, g0, width, C.c_void_p(dev_ptr)))

    def merge_columns(self, dev_ptr: int, nparts: int, per: int) -> None:
        """all ranks' column blocks (block i = genomes from i*per) -> full rows (async)"""
        check(self._lib.pg_result_merge_columns(self._h, C.c_void_p(dev_ptr), nparts, per))

    def columns_bytes_range(self, width: int, first_contig: int, ncontigs: int) -> int:
        return int(self._lib.pg_result_columns_bytes_range(self._h, width, first_contig, ncontigs))

    def extract_columns_range(self, g0: int, width: int, first_contig: int, ncontigs: int, dev_ptr: int) -> None:
        check(self._lib.pg_result_extract_columns_range(self._h, g0, width, first_contig, ncontigs, C.c_void_p(dev_ptr)))

    def merge_columns_range(self, dev_ptr: int, part0: int, nparts: int, per: int, first_contig: int, ncontigs: int,
                            accumulate: bool = False, part_stride_bytes: int = 0) -> None:
        """genome blocks part0 .. part0+nparts-1 (``per`` genomes each) of a contig range -> rows (async);
        ``accumulate`` ORs them into the rows instead of writing the rows whole; ``part_stride_bytes``: distance
        between the blocks at ``dev_ptr`` (0: the range's own block size)"""
        check(self._lib.pg_result_merge_columns_range(self._h, C.c_void_p(dev_ptr), part0, nparts, per, first_contig,
                                                      ncontigs, 1 if accumulate else 0, part_stride_bytes))

    def rows_tensor(self):
        """Zero-copy torch uint8 view of the device bitmap.1 buffer (for RCCL collectives)."""
        import torch
        (ptr, nbytes), _ = self.device_ptrs()

        class _Wrap:
            __cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 3}

        return torch.as_tensor(_Wrap(), device=torch.device("cuda", self.ctx.device))

    def contig_info(self, idx: int) -> dict:
        nk, n100 = C.c_uint64(), C.c_uint64()
        nb, bl = C.c_uint32(), C.c_uint32()
        check(self._lib.pg_result_contig_info(self._h, idx, C.byref(nk), C.byref(n100), C.byref(nb), C.byref(bl)))
        return dict(nkmers=nk.value, nrows100=n100.value, nbins=nb.value, binlen=bl.value)

    def window_stats(self, idx: int, starts, ends, step: int = 1, colsums: bool = True):
        """(hist [nwin, N+1], colsums [nwin, N] or None) of row windows [start, end) of contig idx"""
        starts = np.ascontiguousarray(starts, np.uint64)
        ends = np.ascontiguousarray(ends, np.uint64)
        n, N = len(starts), self.ngenomes
        hist = np.zeros((n, N + 1), np.uint64)
        cs = np.zeros((n, N), np.uint64) if colsums else None
        check(self._lib.pg_result_window_stats(self._h, idx, step, n, _ptr(starts), _ptr(ends), _ptr(hist), _ptr(cs)))
        return hist, cs

    def write_bgzf(self, step: int, gz_path: str, gzi_path: Optional[str] = None, level: int = 6,
                   threads: int = 1, first_contig: int = 0, ncontigs: Optional[int] = None) -> None:
        """Stream the bitmap.<step> payload of contigs [first_contig, first_contig + ncontigs) (default:
        all) from HBM into a BGZF file + .gzi; releases the GIL, so a worker thread can write while
        the main thread anchors on."""
        n = len(self.seqs.lens) - first_contig if ncontigs is None else ncontigs
        check(self._lib.pg_result_write_bgzf_range(self._h, step, first_contig, n, os.fsencode(gz_path),
                                                   os.fsencode(gzi_path) if gzi_path else None, level, threads))

    def download(self, idx: int, want_bitmap1: bool = True, want_bitmap100: bool = True):
        info = self.contig_info(idx)
        nb = self.nbytes
        rows = np.empty((info["nkmers"], nb), np.uint8) if want_bitmap1 else None
        rows100 = np.empty((info["nrows100"], nb), np.uint8) if want_bitmap100 else None
        bins = np.empty((info["nbins"], self.ngenomes + 1), np.uint32)
        check(self._lib.pg_result_download(self._h, idx, _ptr(rows), _ptr(rows100), _ptr(bins)))
        return rows, rows100, bins, info

    def contigs_small(self, first: int = 0, ncontigs: Optional[int] = None) -> "SmallOutputs":
        """geometry and bin histograms of contigs ``first .. first+ncontigs-1`` in two library calls (one device-to-host
        copy) whatever their number — per-contig ``download`` calls cost a fragmented assembly 26 us per contig"""
        n = len(self.seqs.lens) - first if ncontigs is None else ncontigs
        nk, n100 = np.zeros(n, np.uint64), np.zeros(n, np.uint64)
        nb, bl = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
        check(self._lib.pg_result_contigs_small(self._h, first, n, _ptr(nk), _ptr(n100), _ptr(nb), _ptr(bl), None, 0))
        bins = np.zeros((int(nb.sum(dtype=np.uint64)), self.ngenomes + 1), np.uint32)
        if bins.size:
            check(self._lib.pg_result_contigs_small(self._h, first, n, None, None, None, None, _ptr(bins), bins.size))
        return SmallOutputs(nk, n100, nb, bl, bins)

    def colsums(self) -> np.ndarray:
        cs = np.zeros(self.ngenomes, np.uint64)
        check(self._lib.pg_result_colsums(self._h, _ptr(cs)))
        return cs

    def device_ptrs(self):
        d1, d100 = C.c_void_p(), C.c_void_p()
        b1, b100 = C.c_uint64(), C.c_uint64()
        check(self._lib.pg_result_device_ptrs(self._h, C.byref(d1), C.byref(b1), C.byref(d100), C.byref(b100)))
        return (d1.value, b1.value), (d100.value, b100.value)

    def close(self) -> None:
        if self._h:
            self._lib.pg_result_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SmallOutputs:
    """The small per-contig outputs of a run of contigs: geometry arrays and the contigs' bin rows back to back
    (``bins[nbins_total, N + 1]``).  ``out[i]`` is the tuple ``AnchorResult.download(i, False, False)`` returns."""

    def __init__(self, nkmers, nrows100, nbins, binlen, bins):
        self.nkmers, self.nrows100, self.nbins, self.binlen, self.bins = nkmers, nrows100, nbins, binlen, bins
        self.bin_off = np.concatenate([[0], np.cumsum(nbins, dtype=np.int64)])

    def __len__(self):
        return len(self.nkmers)

    def info(self, i: int) -> dict:
        return dict(nkmers=int(self.nkmers[i]), nrows100=int(self.nrows100[i]), nbins=int(self.nbins[i]), binlen=int(self.binlen[i]))

    def __getitem__(self, i: int):
        return None, None, self.bins[self.bin_off[i]:self.bin_off[i + 1]], self.info(i)

    def __iter__(self):
        return (self[i] for i in range(len(self)))

    def write_bins_tsv(self, path: str, ngenomes: int) -> None:
        """bitsum.bins.tsv of these contigs (numbered from 0), formatted by the library"""
        check(_lib.load().pg_write_bins_tsv(os.fsencode(path), ngenomes, len(self), _ptr(np.ascontiguousarray(self.nbins, np.uint32)),
                                            _ptr(np.ascontiguousarray(self.binlen, np.uint32)), _ptr(np.ascontiguousarray(self.bins, np.uint32))))


class BgzfWriter:
    """BGZF + .gzi writer (replaces htslib bgzf_* / bgzip.BGZipWriter + `bgzip -rI`)."""

    RLE = 0x100  # PG_BGZF_RLE: zlib's run-length strategy, OR-ed into level (one-byte rows)

    @staticmethod
    def ROWS(width: int) -> int:
        """PG_BGZF_ROWS(width): row-aware deflate for rows of 2..255 bytes, OR-ed into level"""
        return (width & 0xFF) << 16

    def __init__(self, path: str, level: int = 6, threads: int = 1):
        self._lib = _lib.load()
        h = C.c_void_p()
        check(self._lib.pg_bgzf_open(path.encode(), level, threads, C.byref(h)))
        self._h = h

    def write(self, data) -> None:
        v = _bytes_view(data)
        check(self._lib.pg_bgzf_write(self._h, _ptr(v), v.size))

    def close(self, gzi_path: Optional[str] = None) -> None:
        if self._h:
            h, self._h = self._h, None
            check(self._lib.pg_bgzf_close(h, gzi_path.encode() if gzi_path else None))
