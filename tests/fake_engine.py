"""An ORACLE-BACKED stand-in for ``panagram_amd.engine`` — TEST INFRASTRUCTURE ONLY.

The product has no CPU fallback (``panagram_amd.engine`` needs an MI355X); the multi-process host
logic around it — ``Index.run()``'s mode choice, the genome-sharded chunk pipeline with its
all-gather, the dealing of anchor genomes to writers, the file assembly — still has to be exercised
in the CPU suite with world_size 2 over gloo.  The tests swap this module in for the engine
(``monkeypatch.setattr(panagram_amd.index, "engine", fake_engine)``): same classes and methods,
the compute done by ``oracle/pyoracle.py``, device buffers = CPU torch tensors / numpy arrays
addressed through the same raw pointers the C-ABI takes.  Never imported by the product.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence

import numpy as np

from oracle import pyoracle as po
from panagram_amd.engine import BgzfWriter, SmallOutputs  # the host BGZF writer and the bins' text writer are real (no GPU involved)  # noqa: F401

HBM_FREE = 64 << 30        # what Context.mem_info reports; tests shrink it to force the genome-sharded mode
BYTES_PER_KEY = 43         # PanTable.bytes_for: 16-byte slots at 0.375 load
COLUMNS_DIRECT = False     # (engine.COLUMNS_DIRECT: PG_COLUMNS_DIRECT)
CREATED_DENSITIES = []     # keys_per_line of every PanTable created (0: the library's density): what the planner asked for


def usable_cpus() -> int:
    return 2


def tile_positions() -> int:
    return 1024  # (as the HIP engine's pg_tile_positions())


from panagram_amd.engine import homology_classes  # noqa: E402,F401  (pure host logic)


def _view(ptr: int, nbytes: int) -> np.ndarray:
    return np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(ptr))


class Context:
    def __init__(self, device: int = 0):
        self.device = device

    def mem_info(self):
        return HBM_FREE, HBM_FREE

    def trim(self):
        pass

    def synchronize(self):
        pass

    def set_stream(self, s):
        pass

    def torch_device(self):
        import torch
        return torch.device("cpu")

    def close(self):
        pass


class DeviceBuffer:
    def __init__(self, ctx, nbytes):
        self._a = np.zeros(max(int(nbytes), 16), np.uint8)

    def data_ptr(self):
        return self._a.ctypes.data

    def zero(self, nbytes=None):
        self._a[:len(self._a) if nbytes is None else nbytes] = 0

    def close(self):
        pass


class SeqSet:
    def __init__(self, ctx, names: Sequence[str], seqs: Sequence[bytes]):
        self.ctx, self.names, self.seqs = ctx, list(names), [bytes(s) for s in seqs]
        self.lens = np.array([len(s) for s in self.seqs], np.uint64)

    @classmethod
    def from_fasta(cls, ctx, source):
        if isinstance(source, (str, os.PathLike)):
            from panagram_amd.index import read_fasta
            recs = list(read_fasta(os.fspath(source)))
        else:
            recs = po.parse_fasta_cpp(bytes(np.asarray(source, np.uint8).tobytes() if isinstance(source, np.ndarray) else source))
            recs = [(n, s.translate(None, b" \t\r\n\x0b\x0c")) for n, s in recs]
        return cls(ctx, [n for n, _ in recs], [s for _, s in recs])

    @classmethod
    def from_host(cls, ctx, seqs):
        return cls(ctx, [""] * len(seqs), [bytes(s) for s in seqs])

    @classmethod
    def concat(cls, ctx, sets):
        return cls(ctx, [n for s in sets for n in s.names], [q for s in sets for q in s.seqs])

    @classmethod
    def concat_ranges(cls, ctx, parts):
        return cls(ctx, [n for s, f, c in parts for n in s.names[f:f + c]], [q for s, f, c in parts for q in s.seqs[f:f + c]])

    def slice(self, pieces):
        assert all(s0 % 32 == 0 and s0 + n <= len(self.seqs[c]) for c, s0, n in pieces)
        return SeqSet(self.ctx, [f"{self.names[c]}:{s0}" for c, s0, _ in pieces], [self.seqs[c][s0:s0 + n] for c, s0, n in pieces])

    def total_kmers(self, k):
        return int(sum(max(0, len(s) - k + 1) for s in self.seqs))

    def close(self):
        pass


class KmerSketch:
    def __init__(self, ctx, k):
        self.k, self._regs = k, np.zeros(65536, np.uint8)

    def reset(self):
        self._regs = np.zeros(65536, np.uint8)

    def add(self, ss: SeqSet):
        self._regs = np.maximum(self._regs, po.sketch_registers(ss.seqs, self.k))

    def registers(self):
        return self._regs.copy()

    def estimate(self):
        return po.sketch_estimate(self._regs)

    @staticmethod
    def estimate_registers(regs):
        return po.sketch_estimate(np.asarray(regs, np.uint8))

    def close(self):
        pass


class PanTable:
    def __init__(self, ctx, k, ngenomes, expected_keys=0, coscheduled=0, keys_per_line=0.0):
        self.ctx, self.k, self.ngenomes = ctx, k, ngenomes
        self.coscheduled = coscheduled  # (what the product told the table about how it will be probed)
        self.keys_per_line = keys_per_line  # (... and the density the planner chose for a block table; 0: the library's)
        CREATED_DENSITIES.append(keys_per_line)
        self.nbytes, self.ndbs = (ngenomes + 7) // 8, (ngenomes + 31) // 32
        self._genomes: List[List[bytes]] = [[] for _ in range(ngenomes)]
        self._min = [1] * ngenomes
        self._dbs = None

    @classmethod
    def roomy_density(cls, ctx, k, ngenomes, expected_keys, other_bytes=0, distinct_fraction=None):
        return 0.0  # (the stand-in has no HBM to be generous with)

    @staticmethod
    def bytes_for(k, ngenomes, expected_keys, keys_per_line=0.0):
        return int(int(expected_keys) * BYTES_PER_KEY * ((((ngenomes + 31) // 32) + 1) // 2) * (3.0 / keys_per_line if keys_per_line else 1.0))

    def clear(self):
        self._genomes = [[] for _ in range(self.ngenomes)]
        self._min = [1] * self.ngenomes
        self._dbs = None

    def spill(self):
        return 0.0, 8

    def insert_seqset(self, g, ss: SeqSet, min_count=1):
        self._genomes[g] = list(ss.seqs)
        self._min[g] = min_count
        self._dbs = None

    def update_seqset(self, g, ss: SeqSet):
        # (bits for keys already present only; the stand-in keeps whole k-mer sets — rows of anchored positions are the same)
        self.insert_seqset(g, ss)

    def dbs(self):
        if self._dbs is None:
            self._dbs = po.build_bitvec_dbs(self._genomes, self.k, self._min)
        return self._dbs

    def stats(self):
        n = sum(len(kk) for kk, _ in self.dbs())
        return dict(nkeys=n, nslots=n * 3, nbuckets=n, bytes=n * BYTES_PER_KEY)

    def export(self, i):
        return self.dbs()[i]

    def close(self):
        pass


class AnchorResult:
    def __init__(self, table, seqs, colsums=True, rows_only=False, lowres_step=100, max_bin_len=200000, min_bin_count=100):
        self._init(table.ctx, table, table.k, table.ngenomes, seqs, lowres_step, max_bin_len, min_bin_count)

    def _init(self, ctx, table, k, n, seqs, lowres_step, max_bin_len, min_bin_count):
        self.ctx, self.table, self.k, self.ngenomes, self.seqs = ctx, table, k, n, seqs
        self.nbytes, self.lowres_step = (n + 7) // 8, lowres_step
        self._geo = (max_bin_len, min_bin_count)
        self._nk = [max(0, len(s) - k + 1) for s in seqs.seqs]
        self._rows = [np.zeros((nk, self.nbytes), np.uint8) for nk in self._nk]
        self._done = False

    @classmethod
    def rows_container(cls, ctx, k, ngenomes, seqs, colsums=True, lowres_step=100, max_bin_len=200000, min_bin_count=100):
        r = cls.__new__(cls)
        r._init(ctx, None, k, ngenomes, seqs, lowres_step, max_bin_len, min_bin_count)
        return r

    # ---- probing ----
    def coschedule(self, groups, piece_tiles=0, contig_class=None):
        assert contig_class is None or len(contig_class) == len(groups)

    def coschedule_ranges(self, groups, range_first, piece_tiles=0):
        assert len(groups) == len(self._nk) and list(range_first) == sorted(set(range_first)) and range_first[0] == 0

    def run_range(self, c0, nc):
        for ci in range(c0, c0 + nc):
            if self._nk[ci]:
                self._rows[ci] = po.anchor_contig(self.table.dbs(), self.seqs.seqs[ci], self.k, self.ngenomes)[0]

    def run(self):
        self.run_range(0, len(self._nk))
        self._done = True

    def rows_epilogue(self):
        self._done = True

    # ---- the exchange ----
    def columns_bytes_range(self, width, c0, nc):
        tile = tile_positions()
        return sum((nk + tile - 1) // tile for nk in self._nk[c0:c0 + nc]) * (tile // 8) * width

    def extract_columns_range(self, g0, width, c0, nc, ptr):
        blk = po.extract_columns(self._rows[c0:c0 + nc], self.ngenomes, g0, width, tile=tile_positions())
        _view(ptr, len(blk))[:] = blk

    def merge_columns_range(self, ptr, part0, nparts, per, c0, nc, accumulate=False, part_stride_bytes=0):
        nb = self.columns_bytes_range(per, c0, nc)
        stride = part_stride_bytes or nb
        src = _view(ptr, stride * (nparts - 1) + nb).copy()
        blocks = [np.zeros(nb, np.uint8)] * part0 + [src[i * stride:i * stride + nb] for i in range(nparts)]
        merged = po.merge_columns(blocks, self._nk[c0:c0 + nc], self.ngenomes, per, tile=tile_positions())
        for ci, m in zip(range(c0, c0 + nc), merged):
            self._rows[ci] = (self._rows[ci] | m) if accumulate else m

    # ---- outputs ----
    def _binlen(self, nk):
        binlen, minb = self._geo
        if nk // binlen < minb:
            binlen = nk // minb
        return max(1, binlen)

    def contig_info(self, ci):
        nk = self._nk[ci]
        bl = self._binlen(nk)
        return dict(nkmers=nk, nrows100=(nk + self.lowres_step - 1) // self.lowres_step, nbins=(nk + bl - 1) // bl, binlen=bl)

    def _popc(self, ci):
        return np.unpackbits(self._rows[ci], axis=1, bitorder="little")[:, :self.ngenomes].sum(axis=1)

    def download(self, ci, want_bitmap1=True, want_bitmap100=True):
        info = self.contig_info(ci)
        rows = self._rows[ci]
        popc = self._popc(ci)
        bins = np.zeros((info["nbins"], self.ngenomes + 1), np.uint32)
        for b in range(info["nbins"]):
            bins[b] = np.bincount(popc[b * info["binlen"]:(b + 1) * info["binlen"]], minlength=self.ngenomes + 1)
        return (rows if want_bitmap1 else None, rows[::self.lowres_step] if want_bitmap100 else None, bins, info)

    def contigs_small(self, first=0, ncontigs=None):
        n = len(self._nk) - first if ncontigs is None else ncontigs
        items = [self.download(first + i, False, False) for i in range(n)]
        infos = [it[3] for it in items]
        bins = np.vstack([it[2] for it in items]) if items else np.zeros((0, self.ngenomes + 1), np.uint32)
        return SmallOutputs(np.array([i["nkmers"] for i in infos], np.uint64), np.array([i["nrows100"] for i in infos], np.uint64),
                            np.array([i["nbins"] for i in infos], np.uint32), np.array([i["binlen"] for i in infos], np.uint32),
                            np.ascontiguousarray(bins, np.uint32))

    def contig_colsums(self, idx=0, ncontigs=None):
        n = len(self._nk) - idx if ncontigs is None else ncontigs
        out = np.zeros((n, self.ngenomes), np.uint64)
        for i in range(n):
            out[i] = np.unpackbits(self._rows[idx + i], axis=1, bitorder="little")[:, :self.ngenomes].sum(axis=0)
        return out

    def colsums(self):
        return self.contig_colsums().sum(axis=0)

    def window_stats(self, idx, starts, ends, step=1, colsums=True):
        rows = self._rows[idx] if step == 1 else self._rows[idx][::self.lowres_step]
        h, c = po.window_stats(rows, self.ngenomes, starts, ends)
        return h.astype(np.uint64), (c.astype(np.uint64) if colsums else None)

    def write_bgzf(self, step, gz_path, gzi_path=None, level=6, threads=1, first_contig=0, ncontigs=None):
        n = len(self._nk) - first_contig if ncontigs is None else ncontigs
        w = BgzfWriter(gz_path, level=6, threads=1)
        for ci in range(first_contig, first_contig + n):
            w.write(self._rows[ci] if step == 1 else self._rows[ci][::self.lowres_step])
        w.close(gzi_path)

    def close(self):
        pass
