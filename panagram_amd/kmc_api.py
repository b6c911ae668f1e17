"""Drop-in for the Python-side FFI seam ``panagram.extra.py_kmc_api`` (built from the KMC
submodule; used at ``panagram/index.py:849-860,934-935``), backed by the GPU table.

    from panagram_amd import kmc_api as py_kmc_api     # instead of: from .extra import py_kmc_api
    db = py_kmc_api.KMCFile(); db.OpenForRA(prefix)
    vec = py_kmc_api.CountVec(); db.GetCountersForRead(seq, vec)
    np.array(vec, dtype="uint32")

Same names, argument meaning and (bool) return convention; one deliberate difference: an
ill-formed database raises instead of silently answering zeros.

What this seam costs: it keeps the reference's data flow — the sequence goes up as text and ONE
u32 PER POSITION AND DATABASE comes back to the host (``GetCountersForRead``'s contract), i.e.
4 bytes per position over PCIe around about a millisecond of kernels per 10^8 positions: 6-16 G
k-mers/s by box against 200 G for ``Index.run()`` / ``run_anchor``, which keep the
rows in HBM until they leave as BGZF blocks.  It exists for parity tests and for callers that
cannot change; the fast path is the one INTEGRATION.md patches in.
"""
from __future__ import annotations

import struct
from typing import Optional

import numpy as np

from . import engine

_CTX: Optional[engine.Context] = None


def _context(device: int = 0) -> engine.Context:
    global _CTX
    if _CTX is None:
        _CTX = engine.Context(device)
    return _CTX


class CountVec(list):
    """Stands in for the pybind-bound ``std::vector<uint32>``: ``np.array(vec, dtype='uint32')``
    works; ``vec.array`` is the zero-copy result of the last query."""
    array: Optional[np.ndarray] = None

    def __array__(self, dtype=None, copy=None):
        a = self.array if self.array is not None else np.asarray(list(self), dtype=np.uint32)
        return a.astype(dtype, copy=False) if dtype is not None else a


class KMCFile:
    def __init__(self, device: int = 0):
        self._device = device
        self._tbl: Optional[engine.PanTable] = None
        self._k = 0

    def OpenForRA(self, prefix: str) -> bool:
        try:
            with open(prefix + ".kmc_pre", "rb") as f:
                pre = f.read()
            with open(prefix + ".kmc_suf", "rb") as f:
                suf = f.read()
        except OSError:
            return False
        hoff = struct.unpack("<I", pre[-8:-4])[0]
        self._k = struct.unpack("<I", pre[len(pre) - 8 - hoff:len(pre) - 4 - hoff])[0]
        # (coscheduled=1: GetCountersForRead answers one sequence at a time — no co-scheduling partner)
        self._tbl = engine.PanTable(_context(self._device), self._k, 32, coscheduled=1)
        self._tbl.load_kmc1(0, pre, suf)  # raises PanagramHipError on an ill-formed DB
        return True

    def KmerLength(self) -> int:
        return self._k

    def KmerCount(self) -> int:
        return self._tbl.stats()["nkeys"] if self._tbl else 0

    def GetCountersForRead(self, seq, vec: CountVec) -> bool:
        if self._tbl is None:
            return False
        out = self._tbl.counters_for_read(0, seq)
        vec.array = out
        vec[:] = []  # materialised lazily through __array__
        return True

    def Close(self) -> bool:
        if self._tbl is not None:
            self._tbl.close()
            self._tbl = None
        return True
