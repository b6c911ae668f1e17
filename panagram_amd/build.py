"""Build libpanagram_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

The sources are compiled to objects IN PARALLEL and linked: pg_anchor.hip — 300 instantiations of k_probe / k_insert_tile, three
minutes of hipcc as one unit — goes in as three units (-DPG_ANCHOR_PART=0/1/2: the minimizer windows each unit instantiates; see
the top of that file), so a build takes about a minute of wall time on four cores."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libpanagram_hip.so")
OBJ = os.path.join(HERE, "build")
SOURCES = ["pg_kernels.hip", "pg_anchor.hip", "pg_deflate.hip", "pg_api.hip", "pg_bgzf.cpp"]
HEADERS = ["pg_device.h", "pg_kernels.h", "pg_guard.h", os.path.join("..", "..", "include", "panagram_hip.h")]
# (source, extra defines, object name): the units of one build
UNITS = [("pg_anchor.hip", ["PG_ANCHOR_PART=2"], "pg_anchor_p2.o"), ("pg_anchor.hip", ["PG_ANCHOR_PART=1"], "pg_anchor_p1.o"),
         ("pg_anchor.hip", ["PG_ANCHOR_PART=0"], "pg_anchor_p0.o"), ("pg_api.hip", [], "pg_api.o"), ("pg_kernels.hip", [], "pg_kernels.o"),
         ("pg_deflate.hip", [], "pg_deflate.o"), ("pg_bgzf.cpp", [], "pg_bgzf.o")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result"]


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
    os.makedirs(OBJ, exist_ok=True)
    defines = list(defines)
    units = UNITS
    if any(d.split("=")[0] in ("PG_PHASE_TIMING", "PG_ANCHOR_PART") for d in defines):  # (one device variable / the caller's own split)
        units = [u for u in UNITS if u[0] != "pg_anchor.hip"] + [("pg_anchor.hip", [], "pg_anchor.o")]

    def compile_unit(u):
        src, extra, obj = u
        cmd = [hipcc] + FLAGS + ["-c", "-o", os.path.join(OBJ, obj)] + [f"-D{d}" for d in defines + extra] + [os.path.join(CSRC, src)]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)
        return os.path.join(OBJ, obj)

    jobs = max(1, min(len(units), int(os.environ.get("PG_BUILD_JOBS", "0")) or (os.cpu_count() or 2)))
    with ThreadPoolExecutor(max_workers=jobs) as pool:
        objs = list(pool.map(compile_unit, units))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB + ".tmp"] + objs + ["-lz", "-lpthread"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, defines=[a[2:] for a in sys.argv[1:] if a.startswith("-D")])
