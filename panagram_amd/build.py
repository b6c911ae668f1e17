"""Build libpanagram_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libpanagram_hip.so")
SOURCES = ["pg_kernels.hip", "pg_anchor.hip", "pg_deflate.hip", "pg_api.hip", "pg_bgzf.cpp"]
HEADERS = ["pg_device.h", "pg_kernels.h", "pg_guard.h", os.path.join("..", "..", "include", "panagram_hip.h")]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = True, defines=()) -> str:
    """Compile the HIP extension; returns the path of the shared library.
    ``defines`` (e.g. ["PG_ANCHOR_TILE=1024"]) override kernel tuning constants."""
    if not force and not defines and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libpanagram_hip.so")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result", "-o", LIB] + [f"-D{d}" for d in defines] + [os.path.join(CSRC, s) for s in SOURCES] + ["-lz", "-lpthread"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, defines=[a[2:] for a in sys.argv[1:] if a.startswith("-D")])
