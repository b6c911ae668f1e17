"""panagram_amd — MI355X-native pan-kmer anchoring engine (the `panagram index` anchor hot path).

Product code lives in ``csrc/`` (HIP kernels + C-ABI, built into
``libpanagram_hip.so``); the Python modules mirror the reference's host-side
interface for this path (``panagram/index.py``, ``py_kmc_api``).
"""
__version__ = "0.1.0"
