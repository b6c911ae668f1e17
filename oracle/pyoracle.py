"""CPU oracle (numpy) for the `panagram index` anchor hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is product code: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import it, and only as the checker.  The product path (``panagram_amd``)
never imports this module and fails loudly when the HIP library is missing.

This is a restatement (not a copy) of the reference algorithm:

* canonical k-mer walk + lookup      = KMC ``CKMCFile::GetCountersForRead``
  (third-party, not vendored; call sites ``cpp/anchor.cpp:148``,
  ``panagram/index.py:935``; semantics pinned by running the reference's own
  ``cpp/run_anchor`` binary, see ``tests/golden/make_golden.py``)
* byte scatter / popcount / 1-in-100 decimation / per-bin histogram
                                      = ``KMCdb::write_bits`` ``cpp/anchor.cpp:112-195``
* FASTA line parse, chrs.tsv rows    = ``KMCdb::anchor_fasta`` ``cpp/anchor.cpp:37-109``
* k-mer set construction              = ``kmc -ci1 -fm`` + ``kmc_tools transform
  set_counts`` + ``kmc_tools complex -ocsum`` (``panagram/workflow/Snakefile:54-110``,
  ``panagram/index.py:407-426``): the counter of a k-mer in ``bitvec{i}`` is the OR
  of ``1 << (g % 32)`` over the genomes g of group i that contain it.
* KMC1 on-disk layout                 = SURVEY.md Appendix A (validated by having the
  reference binary read files written by :func:`write_kmc1`).

Parity status: PINNED against golden vectors produced by the reference binary
(``tests/golden/*.npz``; ``tests/test_oracle_golden.py``).
"""
from __future__ import annotations

import io
import os
import struct
from typing import Dict, Iterable, List, Sequence, Tuple

import numpy as np

# ---------------------------------------------------------------------------
# nucleotide codes: A=0 C=1 G=2 T=3, case-insensitive, anything else invalid
# ---------------------------------------------------------------------------
_CODE = np.full(256, 255, dtype=np.uint8)
for _c, _v in zip(b"ACGT", range(4)):
    _CODE[_c] = _v
    _CODE[_c + 32] = _v  # lower case


def encode(seq: bytes) -> np.ndarray:
    """ASCII -> codes 0..3, 255 for any byte outside ``ACGTacgt``."""
    return _CODE[np.frombuffer(seq, dtype=np.uint8)]


def canonical_kmers(seq: bytes, k: int) -> Tuple[np.ndarray, np.ndarray]:
    """All k-mer windows of ``seq``.

    Returns ``(keys, valid)``: ``keys[i]`` = min(value(fwd), value(revcomp)) of
    ``seq[i:i+k]`` as a 2k-bit integer, first symbol most significant
    (SURVEY Appendix A); ``valid[i]`` False when the window holds a non-ACGT byte.
    """
    if not (1 <= k <= 32):
        raise ValueError("k must be in 1..32")
    codes = encode(seq)
    n = len(codes) - k + 1
    if n <= 0:
        return np.zeros(0, np.uint64), np.zeros(0, bool)
    bad = (codes == 255)
    c = np.where(bad, 0, codes).astype(np.uint64)
    fwd = np.zeros(n, np.uint64)
    rc = np.zeros(n, np.uint64)
    for j in range(k):
        w = c[j:j + n]
        fwd |= w << np.uint64(2 * (k - 1 - j))
        rc |= (np.uint64(3) - w) << np.uint64(2 * j)
    badc = np.concatenate([[0], np.cumsum(bad, dtype=np.int64)])
    valid = (badc[k:k + n] - badc[:n]) == 0
    return np.minimum(fwd, rc), valid


# ---------------------------------------------------------------------------
# FASTA parse exactly as cpp/anchor.cpp:77-100 does it (getline; '>' starts a
# record; name = header up to first space; other lines concatenated verbatim)
# ---------------------------------------------------------------------------
def parse_fasta_cpp(data: bytes) -> List[Tuple[str, bytes]]:
    recs: List[Tuple[str, bytes]] = []
    name = None
    parts: List[bytes] = []
    lines = data.split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()  # getline does not yield a trailing empty line
    for line in lines:
        if line[:1] == b">":
            if name:  # `if (!name.empty())`: an empty header never flushes
                recs.append((name, b"".join(parts)))
                parts = []
            name = line[1:].decode("latin-1")
        else:
            parts.append(line)
    if name is not None:
        recs.append((name, b"".join(parts)))
    return [(nm.split(" ")[0], s) for nm, s in recs]


# ---------------------------------------------------------------------------
# k-mer set construction (restates kmc -ci1 + set_counts + complex -ocsum)
# ---------------------------------------------------------------------------
def build_bitvec_dbs(genomes: Sequence[Sequence[bytes]], k: int,
                     min_counts: Sequence[int] = ()) -> List[Tuple[np.ndarray, np.ndarray]]:
    """``genomes[g]`` = list of contig sequences of sample g (sample order =
    samples.tsv order).  Returns one ``(sorted_keys u64, masks u32)`` per group of
    32 samples (``index.py:391-393``); mask bit ``g % 32`` set iff sample g holds
    the canonical k-mer (``workflow/Snakefile:26-28,106-108``).  ``min_counts[g]`` is
    kmc's ``-ci`` for sample g (1 for assemblies, 2 for read sets: ``workflow/Snakefile:88-89``):
    canonical k-mers occurring fewer times in the sample are dropped."""
    ndbs = (len(genomes) + 31) // 32
    dbs = []
    for d in range(ndbs):
        all_keys = []
        all_bits = []
        for g in range(32 * d, min(32 * d + 32, len(genomes))):
            ks = []
            for contig in genomes[g]:
                keys, valid = canonical_kmers(contig, k)
                ks.append(keys[valid])
            if ks:
                u, cnt = np.unique(np.concatenate(ks), return_counts=True)
                ci = min_counts[g] if g < len(min_counts) else 1
                u = u[cnt >= ci]
            else:
                u = np.zeros(0, np.uint64)
            all_keys.append(u)
            all_bits.append(np.full(len(u), 1 << (g % 32), np.uint32))
        keys = np.concatenate(all_keys) if all_keys else np.zeros(0, np.uint64)
        bits = np.concatenate(all_bits) if all_bits else np.zeros(0, np.uint32)
        order = np.argsort(keys, kind="stable")
        keys, bits = keys[order], bits[order]
        if len(keys):
            starts = np.flatnonzero(np.concatenate([[True], keys[1:] != keys[:-1]]))
            masks = np.bitwise_or.reduceat(bits, starts)
            keys = keys[starts]
        else:
            masks = bits
        dbs.append((keys, masks.astype(np.uint32)))
    return dbs


# ---------------------------------------------------------------------------
# KMC1 database files
# ---------------------------------------------------------------------------
SKETCH_BITS = 16


def sketch_registers(seqs: Sequence[bytes], k: int) -> np.ndarray:
    """HyperLogLog registers (2^16, uint8) over the canonical k-mers of ``seqs`` — the CPU
    restatement of the build's own table-sizing sketch (include/panagram_hip.h pg_sketch_*; the
    reference has no counterpart: KMC is only given a memory cap, panagram/workflow/Snakefile:101).
    hash = splitmix64 finalizer of the canonical k-mer value; register = top 16 bits; value = 1 +
    leading zeros of the remaining 48 bits (49 when they are all zero)."""
    regs = np.zeros(1 << SKETCH_BITS, np.uint8)
    for seq in seqs:
        vals, valid = canonical_kmers(seq, k)
        x = vals[valid].astype(np.uint64)
        with np.errstate(over="ignore"):
            x ^= x >> np.uint64(30)
            x *= np.uint64(0xbf58476d1ce4e5b9)
            x ^= x >> np.uint64(27)
            x *= np.uint64(0x94d049bb133111eb)
            x ^= x >> np.uint64(31)
        idx = (x >> np.uint64(64 - SKETCH_BITS)).astype(np.int64)
        rest = x << np.uint64(SKETCH_BITS)
        # leading zeros of a 64-bit word: 63 - floor(log2) through the bit length of the top set bit
        rho = np.full(len(x), 65 - SKETCH_BITS, np.uint8)
        nz = rest != 0
        r = rest[nz]
        lz = np.zeros(len(r), np.uint8)
        for sh in (32, 16, 8, 4, 2, 1):
            hi_clear = (r >> np.uint64(64 - sh)) == 0
            lz[hi_clear] += sh
            r[hi_clear] <<= np.uint64(sh)
        rho[nz] = lz + 1
        np.maximum.at(regs, idx, rho)
    return regs


def sketch_estimate(regs: np.ndarray) -> int:
    """HyperLogLog estimate with the small-range (linear counting) correction."""
    m = float(len(regs))
    est = (0.7213 / (1.0 + 1.079 / m)) * m * m / float(np.sum(np.ldexp(1.0, -regs.astype(np.int64))))
    zeros = int(np.count_nonzero(regs == 0))
    if est <= 2.5 * m and zeros:
        est = m * np.log(m / zeros)
    return int(est + 0.5)


def pick_lut_prefix_len(k: int, nkeys: int) -> int:
    """(k - lut_prefix_len) % 4 == 0, sized so buckets stay small."""
    cands = [p for p in range(1, min(k, 13)) if (k - p) % 4 == 0]
    best = cands[0]
    for p in cands:
        if 4 ** p <= max(nkeys, 1) * 4:
            best = p
    return best


def write_kmc1(prefix: str, keys: np.ndarray, counters: np.ndarray, k: int,
               lut_prefix_len: int | None = None, counter_size: int = 4,
               min_count: int = 1, max_count: int = 0xFFFFFFFF) -> None:
    """Write ``prefix.kmc_pre`` / ``prefix.kmc_suf`` in the KMC1 layout."""
    keys = np.asarray(keys, np.uint64)
    counters = np.asarray(counters, np.uint32)
    assert np.all(keys[1:] > keys[:-1]), "keys must be sorted and unique"
    p = pick_lut_prefix_len(k, len(keys)) if lut_prefix_len is None else lut_prefix_len
    assert (k - p) % 4 == 0
    suf_syms = k - p
    suf_bytes = suf_syms // 4
    pref = (keys >> np.uint64(2 * suf_syms)).astype(np.int64)
    lut = np.searchsorted(pref, np.arange(4 ** p, dtype=np.int64), side="left").astype(np.uint64)
    with open(prefix + ".kmc_pre", "wb") as f:
        f.write(b"KMCP")
        f.write(lut.tobytes())
        hdr = struct.pack("<IIIIIIQB3x24xI", k, 0, counter_size, p, min_count,
                          max_count & 0xFFFFFFFF, len(keys), 0, 0)
        assert len(hdr) == 64
        f.write(hdr)
        f.write(struct.pack("<I", 64))
        f.write(b"KMCP")
    rec = np.zeros((len(keys), suf_bytes + counter_size), np.uint8)
    for b in range(suf_bytes):  # most-significant suffix byte first
        rec[:, b] = ((keys >> np.uint64(8 * (suf_bytes - 1 - b))) & np.uint64(0xFF)).astype(np.uint8)
    for b in range(counter_size):
        rec[:, suf_bytes + b] = ((counters >> np.uint32(8 * b)) & np.uint32(0xFF)).astype(np.uint8)
    with open(prefix + ".kmc_suf", "wb") as f:
        f.write(b"KMCS")
        f.write(rec.tobytes())
        f.write(b"KMCS")


def read_kmc1(prefix: str):
    """Read a KMC1 database -> dict(k, keys u64 sorted, counters u32, min_count, max_count)."""
    with open(prefix + ".kmc_pre", "rb") as f:
        pre = f.read()
    with open(prefix + ".kmc_suf", "rb") as f:
        suf = f.read()
    return parse_kmc1(pre, suf)


def parse_kmc1(pre: bytes, suf: bytes):
    if pre[:4] != b"KMCP" or pre[-4:] != b"KMCP" or suf[:4] != b"KMCS" or suf[-4:] != b"KMCS":
        raise ValueError("not a KMC database (bad markers)")
    (hoff,) = struct.unpack("<I", pre[-8:-4])
    hdr = pre[len(pre) - 8 - hoff:len(pre) - 8]
    k, mode, csz, p, minc, maxc, total = struct.unpack("<IIIIIIQ", hdr[:32])
    both = hdr[32]
    (ver,) = struct.unpack("<I", hdr[60:64])
    if ver != 0:
        raise ValueError("only the KMC1 layout (kmc_version=0) is supported")
    if mode != 0:
        raise ValueError("quality-mode KMC databases are not supported")
    lut = np.frombuffer(pre, np.uint64, 4 ** p, 4)
    suf_bytes = (k - p) // 4
    rec = np.frombuffer(suf, np.uint8, total * (suf_bytes + csz), 4).reshape(total, suf_bytes + csz)
    keys = np.zeros(total, np.uint64)
    for b in range(suf_bytes):
        keys |= rec[:, b].astype(np.uint64) << np.uint64(8 * (suf_bytes - 1 - b))
    # prefix of record i = the LUT bucket it falls in
    bounds = np.concatenate([lut, [np.uint64(total)]]).astype(np.int64)
    pref = np.repeat(np.arange(4 ** p, dtype=np.uint64), np.diff(bounds))
    keys |= pref << np.uint64(2 * (k - p))
    counters = np.zeros(total, np.uint32)
    for b in range(csz):
        counters |= rec[:, suf_bytes + b].astype(np.uint32) << np.uint32(8 * b)
    return dict(k=k, keys=keys, counters=counters, min_count=minc, max_count=maxc,
                lut_prefix_len=p, counter_size=csz, both_strands=both)


# ---------------------------------------------------------------------------
# KMC2 layout (kmc_version 0x200: what `kmc` itself writes, workflow/Snakefile:101-104).  KMC is an un-vendored
# third-party dependency (refresh-bio/KMC v3.2.1, setup.py:28-31); this restates its published file description
# (kmc_api: [marker][one prefix LUT per bin][signature map][header][header position][marker]) and its signature
# rule (kmc_api/mmer.h).  PINNED the only way available here: databases written by write_kmc2 are READ BY THE
# REFERENCE BINARY (cpp/run_anchor links KMC's reader), whose outputs must equal the ones it gives for the same
# k-mers in KMC1 files (tests/golden/make_golden.py: kmc2_* fixtures).
# ---------------------------------------------------------------------------
def _mmer_allowed(mmer: int, length: int) -> bool:
    """KMC's restriction on signatures: no m-mer starting with AAA or ACA, none with AA anywhere but at its start"""
    if (mmer & 0x3F) == 0x3F:   # ends with TTT
        return False
    if (mmer & 0x3F) == 0x3B:   # ends with TGT
        return False
    if (mmer & 0x3C) == 0x3C:   # TT before the last symbol
        return False
    for _ in range(length - 3):
        if (mmer & 0xF) == 0:   # AA inside
            return False
        mmer >>= 2
    if mmer == 0 or mmer == 0x04 or (mmer & 0xF) == 0:  # AAA / ACA / *AA at the start
        return False
    return True


def kmc_signature_norm(length: int) -> np.ndarray:
    """norm[m-mer] = min(allowed(m-mer) ? m-mer : special, allowed(revcomp) ? revcomp : special), special = 4^len"""
    special = 1 << (2 * length)
    allowed = np.array([_mmer_allowed(i, length) for i in range(special)])
    idx = np.arange(special, dtype=np.int64)
    rev = np.zeros(special, np.int64)
    for j in range(length):
        rev |= (3 - ((idx >> (2 * j)) & 3)) << (2 * (length - 1 - j))
    a = np.where(allowed, idx, special)
    b = np.where(allowed[rev], rev, special)
    return np.minimum(a, b)


def kmc_signatures(keys: np.ndarray, k: int, sig_len: int) -> np.ndarray:
    """signature of every k-mer value = the smallest norm over its k - sig_len + 1 m-mers"""
    norm = kmc_signature_norm(sig_len)
    keys = np.asarray(keys, np.uint64)
    best = np.full(len(keys), 1 << (2 * sig_len), np.int64)
    mask = np.uint64((1 << (2 * sig_len)) - 1)
    for j in range(k - sig_len + 1):
        mm = ((keys >> np.uint64(2 * j)) & mask).astype(np.int64)
        best = np.minimum(best, norm[mm])
    return best


def write_kmc2(prefix: str, keys: np.ndarray, counters: np.ndarray, k: int, lut_prefix_len: int,
               sig_len: int = 7, nbins: int = 13, counter_size: int = 4, min_count: int = 1,
               max_count: int = 0xFFFFFFFF, guard: bool = False) -> None:
    """``prefix.kmc_pre`` / ``.kmc_suf`` in the KMC2 layout: signature -> bin through an (arbitrary but
    recorded) map, every bin sorted and given its own prefix LUT, the suffix file = the bins one after another."""
    keys = np.asarray(keys, np.uint64)
    counters = np.asarray(counters, np.uint32)
    p = lut_prefix_len
    assert (k - p) % 4 == 0
    sig = kmc_signatures(keys, k, sig_len)
    sig_map = (np.arange((1 << (2 * sig_len)) + 1, dtype=np.int64) * 2654435761 % nbins).astype(np.uint32)
    bins = sig_map[sig].astype(np.int64)
    order = np.lexsort((keys, bins))
    keys, counters, bins = keys[order], counters[order], bins[order]
    suf_syms = k - p
    suf_bytes = suf_syms // 4
    pref = (keys >> np.uint64(2 * suf_syms)).astype(np.int64)
    # record number of the first record of (bin b, prefix x) = records with (bin, prefix) < (b, x)
    combined = bins * (4 ** p) + pref
    lut = np.searchsorted(combined, np.arange(nbins * 4 ** p, dtype=np.int64), side="left").astype(np.uint64)
    with open(prefix + ".kmc_pre", "wb") as f:
        f.write(b"KMCP")
        f.write(lut.tobytes())
        if guard:
            f.write(struct.pack("<Q", len(keys)))
        f.write(sig_map.tobytes())
        hdr = struct.pack("<IIIIIIIQB3x24xI", k, 0, counter_size, p, sig_len, min_count, max_count & 0xFFFFFFFF,
                          len(keys), 0, 0x200)
        assert len(hdr) == 68
        f.write(hdr)
        f.write(struct.pack("<I", 68))
        f.write(b"KMCP")
    rec = np.zeros((len(keys), suf_bytes + counter_size), np.uint8)
    for b in range(suf_bytes):
        rec[:, b] = ((keys >> np.uint64(8 * (suf_bytes - 1 - b))) & np.uint64(0xFF)).astype(np.uint8)
    for b in range(counter_size):
        rec[:, suf_bytes + b] = ((counters >> np.uint32(8 * b)) & np.uint32(0xFF)).astype(np.uint8)
    with open(prefix + ".kmc_suf", "wb") as f:
        f.write(b"KMCS")
        f.write(rec.tobytes())
        f.write(b"KMCS")


def parse_kmc2(pre: bytes, suf: bytes):
    """KMC2 images -> dict(k, keys sorted, counters, ...): every record's prefix from its (bin, prefix) LUT slot"""
    if pre[:4] != b"KMCP" or pre[-4:] != b"KMCP" or suf[:4] != b"KMCS" or suf[-4:] != b"KMCS":
        raise ValueError("not a KMC database (bad markers)")
    (hoff,) = struct.unpack("<I", pre[-8:-4])
    hdr = pre[len(pre) - 8 - hoff:len(pre) - 8]
    k, mode, csz, p, sig_len, minc, maxc, total = struct.unpack("<IIIIIIIQ", hdr[:36])
    (ver,) = struct.unpack("<I", hdr[-4:])
    if ver != 0x200:
        raise ValueError("not the KMC2 layout")
    map_bytes = ((1 << (2 * sig_len)) + 1) * 4
    lut_bytes = len(pre) - 4 - map_bytes - hoff - 8
    nlut = lut_bytes // 8
    if nlut % (4 ** p):
        nlut -= 1  # a guard entry
    lut = np.frombuffer(pre, np.uint64, nlut, 4).astype(np.int64)
    suf_bytes = (k - p) // 4
    rec = np.frombuffer(suf, np.uint8, total * (suf_bytes + csz), 4).reshape(total, suf_bytes + csz)
    keys = np.zeros(total, np.uint64)
    for b in range(suf_bytes):
        keys |= rec[:, b].astype(np.uint64) << np.uint64(8 * (suf_bytes - 1 - b))
    bounds = np.concatenate([lut, [total]])
    slot = np.repeat(np.arange(nlut, dtype=np.int64), np.diff(bounds))
    keys |= (slot % (4 ** p)).astype(np.uint64) << np.uint64(2 * (k - p))
    counters = np.zeros(total, np.uint32)
    for b in range(csz):
        counters |= rec[:, suf_bytes + b].astype(np.uint32) << np.uint32(8 * b)
    order = np.argsort(keys, kind="stable")
    return dict(k=k, keys=keys[order], counters=counters[order], min_count=minc, max_count=maxc,
                lut_prefix_len=p, counter_size=csz, signature_len=sig_len, nbins=nlut // (4 ** p))


# ---------------------------------------------------------------------------
# lookup = GetCountersForRead
# ---------------------------------------------------------------------------
def counters_for_read(db: Tuple[np.ndarray, np.ndarray], seq: bytes, k: int,
                      min_count: int = 1, max_count: int = 0xFFFFFFFF) -> np.ndarray:
    keys, masks = db
    q, valid = canonical_kmers(seq, k)
    out = np.zeros(len(q), np.uint32)
    if len(keys) == 0 or len(q) == 0:
        return out
    idx = np.searchsorted(keys, q)
    idx[idx >= len(keys)] = len(keys) - 1
    hit = valid & (keys[idx] == q)
    c = masks[idx]
    hit &= (c >= min_count) & (c <= max_count)
    out[hit] = c[hit]
    return out


# ---------------------------------------------------------------------------
# write_bits / anchor_fasta restatement
# ---------------------------------------------------------------------------
def row_byte_counts(ngenomes: int) -> List[int]:
    """bytes taken from each DB's u32 (cpp/anchor.cpp:139-145; index.py:940-945)."""
    nbytes = (ngenomes + 7) // 8
    ndbs = (ngenomes + 31) // 32
    ns = []
    for d in range(ndbs):
        if nbytes <= 4:
            ns.append(nbytes)
        elif d == ndbs - 1 and nbytes % 4 > 0:
            ns.append(nbytes % 4)
        else:
            ns.append(4)
    return ns


def bin_length(nkmers: int) -> int:
    """cpp/anchor.cpp:114-118 (== index.py:1170-1172 with the default params)."""
    binlen = 200000
    if nkmers // binlen < 100:
        binlen = nkmers // 100
    return binlen


def anchor_contig(dbs, seq: bytes, k: int, ngenomes: int, min_count: int = 1,
                  max_count: int = 0xFFFFFFFF):
    """One contig -> (rows[nkmers, nbytes] u8, rows100, bins[nbins, N+1] i64,
    bin_starts, colsums[N] i64).  Bit g of a row (byte g//8, bit g%8) = genome g."""
    nbytes = (ngenomes + 7) // 8
    nkmers = len(seq) - k + 1
    ns = row_byte_counts(ngenomes)
    rows = np.zeros((nkmers, nbytes), np.uint8)
    popc = np.zeros(nkmers, np.int64)
    off = 0
    for d, n in enumerate(ns):
        ints = counters_for_read(dbs[d], seq, k, min_count, max_count)
        b4 = ints.view(np.uint8).reshape(nkmers, 4)
        rows[:, off:off + n] = b4[:, :n]
        popc += np.unpackbits(b4, axis=1).sum(axis=1, dtype=np.int64)
        off += n
    rows100 = rows[::100]
    binlen = bin_length(nkmers)
    starts = np.arange(0, nkmers, binlen, dtype=np.int64)
    bins = np.zeros((len(starts), ngenomes + 1), np.int64)
    for i, s in enumerate(starts):
        bins[i] = np.bincount(popc[s:s + binlen], minlength=ngenomes + 1)[:ngenomes + 1]
    bits = np.unpackbits(rows, axis=1, bitorder="little")[:, :ngenomes]
    colsums = bits.sum(axis=0, dtype=np.int64)
    return rows, rows100, bins, starts, colsums


def anchor_fasta(dbs, fasta_bytes: bytes, k: int, ngenomes: int, min_count: int = 1,
                 max_count: int = 0xFFFFFFFF):
    """Whole anchor FASTA -> dict of payloads/texts as ``run_anchor`` writes them
    (decompressed), plus ``colsums`` for total_paircounts (index.py:1051)."""
    recs = parse_fasta_cpp(fasta_bytes)
    b1, b100 = [], []
    bins_txt = io.StringIO()
    bins_txt.write("chr\tstart" + "".join(f"\t{i}" for i in range(ngenomes + 1)) + "\n")
    chrs_txt = io.StringIO()
    chrs_txt.write("name\tid\tsize\tgene_count\n")
    colsums = np.zeros(ngenomes, np.int64)
    for ci, (name, seq) in enumerate(recs):
        rows, rows100, bins, starts, cs = anchor_contig(dbs, seq, k, ngenomes, min_count, max_count)
        b1.append(rows.tobytes())
        b100.append(rows100.tobytes())
        for s, r in zip(starts, bins):
            bins_txt.write(f"{ci}\t{s}" + "".join(f"\t{c}" for c in r) + "\n")
        chrs_txt.write(f"{name}\t{ci}\t{len(rows)}\t0\n")
        colsums += cs
    return dict(bitmap1=b"".join(b1), bitmap100=b"".join(b100),
                bins_tsv=bins_txt.getvalue(), chrs_tsv=chrs_txt.getvalue(), colsums=colsums)


# ---------------------------------------------------------------------------
# deterministic synthetic genomes
# ---------------------------------------------------------------------------
_ACGT = np.frombuffer(b"ACGT", np.uint8)


def synth_genomes(ngenomes: int, contig_lens: Sequence[int], d: float, seed: int) -> List[List[np.ndarray]]:
    """Base genome i.i.d. uniform; genome g>0 = base with per-base substitution at
    rate d (new base != old), rng seed ``seed+g``.  Returns code arrays (0..3)."""
    rng = np.random.default_rng(seed)
    base = [rng.integers(0, 4, L, dtype=np.uint8) for L in contig_lens]
    out = [base]
    for g in range(1, ngenomes):
        r = np.random.default_rng(seed + g)
        contigs = []
        for b in base:
            mut = r.random(len(b)) < d
            shift = r.integers(1, 4, len(b), dtype=np.uint8)
            contigs.append(np.where(mut, (b + shift) & 3, b).astype(np.uint8))
        out.append(contigs)
    return out


def codes_to_ascii(codes: np.ndarray) -> bytes:
    return _ACGT[codes].tobytes()


def fasta_text(names: Sequence[str], seqs: Sequence[bytes], width: int = 80) -> bytes:
    out = []
    for nm, s in zip(names, seqs):
        out.append(b">" + nm.encode() + b"\n")
        for i in range(0, len(s), width):
            out.append(s[i:i + width] + b"\n")
    return b"".join(out)


# ---------------------------------------------------------------------------
# window statistics over bitmap rows: the per-gene occupancy tabulation of
# Genome.run_anchor (index.py:1055-1064: np.unique(bitsum[start:end])) and the
# per-bin pancount / paircount tables of Index.bitmap_to_bins (index.py:438-449)
# ---------------------------------------------------------------------------
def window_stats(rows: np.ndarray, ngenomes: int, starts, ends) -> Tuple[np.ndarray, np.ndarray]:
    """``rows`` = (n, nbytes) u8 bitmap payload.  Returns ``hist[w, c]`` = positions of window
    ``[starts[w], ends[w])`` whose row has c bits set, ``colsums[w, g]`` = positions with bit g."""
    bits = np.unpackbits(rows, axis=1, bitorder="little")[:, :ngenomes]
    popc = bits.sum(axis=1)
    hist = np.zeros((len(starts), ngenomes + 1), np.int64)
    cs = np.zeros((len(starts), ngenomes), np.int64)
    for w, (s, e) in enumerate(zip(starts, ends)):
        s, e = int(s), min(int(e), len(rows))
        if s < e:
            hist[w] = np.bincount(popc[s:e], minlength=ngenomes + 1)
            cs[w] = bits[s:e].sum(axis=0)
    return hist, cs


# ---------------------------------------------------------------------------
# genome-sharded exchange: compact bit columns <-> rows.  Layout restated from
# include/panagram_hip.h: tile t (`tile` = pg_tile_positions() positions of one contig: 1024 today, 512 until round 4)
# owns ns = tile / 64 slots of `width` u64 words; word (ns t + s) * width + j holds genome g0 + j at positions
# 64 s .. 64 s + 63 of the tile (bit l).  `tile` has no default: a caller that guesses the engine's tile gets blocks of
# another geometry without any error.
# ---------------------------------------------------------------------------
def extract_columns(contig_rows: Sequence[np.ndarray], ngenomes: int, g0: int, width: int, *, tile: int) -> np.ndarray:
    """(``tile``: positions per tile of the engine whose blocks these are — pg_tile_positions(); a tile owns tile / 64 slots)"""
    tiles, ns = [], tile // 64
    for rows in contig_rows:
        bits = np.unpackbits(rows, axis=1, bitorder="little")[:, :ngenomes]
        nt = (len(rows) + tile - 1) // tile
        pad = np.zeros((nt * tile, width), np.uint8)
        hi = min(ngenomes, g0 + width)
        if hi > g0:
            pad[:len(rows), :hi - g0] = bits[:, g0:hi]
        # [tile, slot, lane, j] -> [tile, slot, j, lane] -> 64 lane bits -> one u64, little bit order
        v = pad.reshape(nt, ns, 64, width).transpose(0, 1, 3, 2)
        tiles.append(np.packbits(v, axis=3, bitorder="little").reshape(nt * ns * width * 8))
    return np.concatenate(tiles) if tiles else np.zeros(0, np.uint8)


def merge_columns(blocks: Sequence[np.ndarray], contig_nkmers: Sequence[int], ngenomes: int, per: int, *, tile: int) -> List[np.ndarray]:
    nbytes = (ngenomes + 7) // 8
    out, off, ns = [], 0, tile // 64
    for nk in contig_nkmers:
        nt = (nk + tile - 1) // tile
        bits = np.zeros((nt * tile, nbytes * 8), np.uint8)
        for i, blk in enumerate(blocks):
            words = blk[off * per:(off + nt * ns * 8) * per]
            v = np.unpackbits(words.reshape(nt, ns, per, 8), axis=3, bitorder="little")  # [tile, slot, j, lane]
            v = v.transpose(0, 1, 3, 2).reshape(nt * tile, per)
            hi = min(ngenomes, (i + 1) * per)
            if hi > i * per:
                bits[:, i * per:hi] = v[:, :hi - i * per]
        out.append(np.packbits(bits[:nk], axis=1, bitorder="little"))
        off += nt * ns * 8
    return out
