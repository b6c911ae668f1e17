"""ctypes wrapper of oracle/anchor_oracle.c (TEST INFRASTRUCTURE / CPU baseline only)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        src = os.path.join(_HERE, "anchor_oracle.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.run(["make", "-s", "-C", _HERE], check=True)
        L = C.CDLL(so)
        L.odb_from_arrays.restype = C.c_void_p
        L.odb_from_arrays.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_uint32, C.c_uint32]
        L.odb_from_kmc1.restype = C.c_void_p
        L.odb_from_kmc1.argtypes = [C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64]
        L.odb_free.argtypes = [C.c_void_p]
        L.odb_counters_for_read.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
        L.oracle_write_bits.restype = C.c_int64
        L.oracle_write_bits.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p]
        _LIB = L
    return _LIB


class OracleDB:
    """KMC1-style DB in host memory: prefix LUT + sorted suffix records, binary-search lookup."""

    def __init__(self, handle):
        if not handle:
            raise ValueError("oracle DB construction failed")
        self._h = C.c_void_p(handle)

    @classmethod
    def from_arrays(cls, keys, counters, k, lut_p=None, min_count=1, max_count=0xFFFFFFFF):
        from . import pyoracle as po
        keys = np.ascontiguousarray(keys, np.uint64)
        counters = np.ascontiguousarray(counters, np.uint32)
        p = po.pick_lut_prefix_len(k, len(keys)) if lut_p is None else lut_p
        return cls(lib().odb_from_arrays(keys.ctypes.data, counters.ctypes.data, len(keys), k, p,
                                         min_count, max_count))

    @classmethod
    def from_files(cls, prefix):
        pre = open(prefix + ".kmc_pre", "rb").read()
        suf = open(prefix + ".kmc_suf", "rb").read()
        return cls(lib().odb_from_kmc1(pre, len(pre), suf, len(suf)))

    def counters_for_read(self, seq: bytes, k: int) -> np.ndarray:
        s = np.frombuffer(seq, np.uint8)
        out = np.zeros(max(0, len(s) - k + 1), np.uint32)
        lib().odb_counters_for_read(self._h, s.ctypes.data, len(s), out.ctypes.data)
        return out

    def close(self):
        if self._h:
            lib().odb_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def write_bits(dbs, ngenomes: int, seq, k: int):
    """cpp/anchor.cpp:112-195 for one contig -> rows, rows100, bins, bin_starts."""
    s = np.frombuffer(seq, np.uint8) if not isinstance(seq, np.ndarray) else seq
    nbytes = (ngenomes + 7) // 8
    nk = max(0, len(s) - k + 1)
    binlen = 200000
    if nk // binlen < 100:
        binlen = nk // 100
    nbins = (nk + binlen - 1) // binlen if binlen else 0
    rows = np.zeros((nk, nbytes), np.uint8)
    rows100 = np.zeros(((nk + 99) // 100, nbytes), np.uint8)
    bins = np.zeros((max(nbins, 1), ngenomes + 1), np.uint64)
    starts = np.zeros(max(nbins, 1), np.uint64)
    arr = (C.c_void_p * len(dbs))(*[d._h for d in dbs])
    n = lib().oracle_write_bits(arr, len(dbs), ngenomes, s.ctypes.data, len(s), rows.ctypes.data,
                                rows100.ctypes.data, bins.ctypes.data, starts.ctypes.data)
    if n < 0:
        raise ValueError("contig with fewer than 100 k-mers: the reference divides by zero")
    return rows, rows100, bins[:n], starts[:n]
