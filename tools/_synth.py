"""Synthetic pangenomes for the tools in this directory (SURVEY §8d generator: i.i.d. base genome,
derived genomes by per-base substitution).  The oracle package is off limits here: only tests, smoke() and
bench.py's cpu_baseline leg may use it."""
from typing import List, Sequence

import numpy as np

_ACGT = np.frombuffer(b"ACGT", np.uint8)


def synth_genomes(ngenomes: int, contig_lens: Sequence[int], d: float, seed: int) -> List[List[np.ndarray]]:
    rng = np.random.default_rng(seed)
    base = [rng.integers(0, 4, n, dtype=np.uint8) for n in contig_lens]
    out = [base]
    for g in range(1, ngenomes):
        r = np.random.default_rng(seed + g)
        out.append([np.where(r.random(len(b)) < d, (b + r.integers(1, 4, len(b), dtype=np.uint8)) & 3, b).astype(np.uint8)
                    for b in base])
    return out


def codes_to_ascii(codes: np.ndarray) -> bytes:
    return _ACGT[codes].tobytes()


def fasta_text(names: Sequence[str], seqs: Sequence[bytes], width: int = 80) -> bytes:
    parts = []
    for nm, s in zip(names, seqs):
        parts.append(b">" + nm.encode() + b"\n")
        parts.extend(s[i:i + width] + b"\n" for i in range(0, len(s), width))
    return b"".join(parts)
