"""Counterpart of the reference's ``scripts/make_bins_bits.py``: per window of
``bin_size`` positions of a chromosome, how many of the sampled (every ``step``-th) k-mer positions are
*unique* (present in exactly 1 genome) and *universal* (present in all), from the low-resolution
bitmap.  Same arithmetic as ``make_bars`` (``scripts/make_bins_bits.py:34-59``): position of sample j is
``j*step``, its window ``j*step // bin_size``; x = ``bin_size, 2*bin_size, ... < nsamples*step`` and
only the first ``len(x)`` windows are reported (``:97-114``).  Reads what ``Genome.run_anchor`` wrote
(``bitmap.100.gz``/``.gzi``) through the reference's addressing rule — no GPU involved, 1 % of the
positions."""
from __future__ import annotations

import sys
from typing import List, Tuple

import numpy as np


def make_bars(occ: np.ndarray, num_samples: int, bin_size: int = 200000, step: int = 100) -> Tuple[List[int], List[int], List[int]]:
    """occ[j] = number of genomes holding sampled position j.  Returns (x, z_univ, z_unique)."""
    occ = np.asarray(occ)
    x = list(range(bin_size, len(occ) * step, bin_size))
    win = (np.arange(len(occ), dtype=np.int64) * step) // bin_size
    nwin = len(x) + 1
    z_unique = np.bincount(win[occ == 1], minlength=nwin)[:nwin]
    z_univ = np.bincount(win[(occ == num_samples) & (occ != 1)], minlength=nwin)[:nwin]
    return x, [int(v) for v in z_univ[:len(x)]], [int(v) for v in z_unique[:len(x)]]


def chromosome_bars(index, genome: str, chrom: str, bin_size: int = 200000, step: int = 100):
    g = index.genomes[genome]
    if g.blocks is None:
        g.init_read()
    occ = index.query_bitmap(genome, chrom, 0, g.seq_len(chrom), step).to_numpy().sum(axis=1)
    return make_bars(occ, index.ngenomes, bin_size, step)


def main(argv=None) -> int:
    from .index import Index
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) < 2:
        print("usage: python -m panagram_amd.bins_bits <index_dir> <anchor_genome> [chrom ...]")
        return 2
    idx = Index(argv[0], mode="r")
    g = idx.genomes[argv[1]]
    g.init_read()
    for chrom in (argv[2:] or list(g.chrs.index)):
        x, z_univ, z_unique = chromosome_bars(idx, argv[1], chrom)
        for arr in (x, z_univ, z_unique):  # the reference prints comma-terminated lines
            print("".join(f"{v}," for v in arr))
    return 0


if __name__ == "__main__":
    sys.exit(main())
