"""CPU: the C-ABI library loads and exports every symbol include/panagram_hip.h declares."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "panagram_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(pg_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from panagram_amd import _lib, build
    build.build(verbose=False)
    lib = _lib.load()
    syms = header_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/panagram_hip.h but not exported"
    # and the ctypes prototype table covers the header exactly
    assert sorted(_lib.PROTOTYPES) == syms


def test_no_gpu_means_loud_failure():
    """Without a device the product refuses to run (no CPU fallback)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from panagram_amd import engine
    with pytest.raises(engine.PanagramHipError) as ei:
        engine.Context(0)
    assert "no CPU fallback" in str(ei.value) or "HIP" in str(ei.value)


def test_product_never_imports_oracle():
    """Nothing under panagram_amd/ may reference the oracle."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "panagram_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle" not in src.replace("no oracle", ""), f"{f} mentions the oracle"
    # tools/ are measurement scripts around the product: they must not lean on the oracle either;
    # bench.py may, in its cpu_baseline leg only
    for f in os.listdir(os.path.join(ROOT, "tools")):
        if f.endswith((".py", ".sh")):
            src = open(os.path.join(ROOT, "tools", f), errors="replace").read()
            assert "import oracle" not in src and "from oracle" not in src, f"tools/{f} imports the oracle"
    bench = open(os.path.join(ROOT, "bench.py")).read()
    assert bench.count("from oracle") == 1 and bench.index("from oracle") > bench.index("def cpu_baseline(")
    assert bench.index("from oracle") < bench.index("def main(")


def test_bgzf_writer_roundtrip(tmp_path):
    """BGZF writer is host-only code: readable by gzip, .gzi offsets self-consistent."""
    import gzip
    import numpy as np
    from panagram_amd import engine
    rng = np.random.default_rng(5)
    data = rng.integers(0, 4, 300000, dtype=np.uint8).tobytes() + bytes(200000)
    p = str(tmp_path / "x.gz")
    w = engine.BgzfWriter(p, threads=3)
    for i in range(0, len(data), 77777):  # ragged writes
        w.write(data[i:i + 77777])
    w.close(p + "i")
    assert gzip.open(p, "rb").read() == data
    gzi = np.fromfile(p + "i", np.uint64)
    nblocks = (len(data) + 65279) // 65280
    assert gzi[0] == nblocks - 1
    ent = gzi[1:].reshape(-1, 2)
    assert list(ent[:, 1]) == [65280 * i for i in range(1, nblocks)]
    raw = open(p, "rb").read()
    for coff in ent[:, 0]:
        assert raw[int(coff):int(coff) + 4] == b"\x1f\x8b\x08\x04"  # a BGZF block starts there
    assert raw[-28:] == bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0])


def test_every_entry_point_sits_behind_the_exception_firewall():
    """no C++ exception crosses the ABI.  Every extern "C" int function of the host translation units
    whose body is more than a one-line expression opens with PG_API_BEGIN (pg_guard.h) and closes with PG_API_END."""
    for name in ("pg_api.hip", "pg_bgzf.cpp"):
        lines = open(os.path.join(ROOT, "panagram_amd", "csrc", name)).read().split("\n")
        n = 0
        for i, line in enumerate(lines):
            if not line.startswith('extern "C" int pg_') or line.rstrip().endswith(("}", ";")):
                continue
            j = i
            while not lines[j].rstrip().endswith("{"):
                j += 1
            end = lines.index("}", j)
            assert lines[j + 1].strip() == "PG_API_BEGIN", f"{name}:{i + 1} {line[:60]} is not guarded"
            assert lines[end - 1].strip() == "PG_API_END", f"{name}:{i + 1} {line[:60]}: guard not closed"
            n += 1
        assert n >= 3


_FIREWALL_CHILD = r"""
import ctypes, os, resource, sys
sys.path.insert(0, %(root)r)
from panagram_amd import _lib
lib = _lib.load()
path = os.path.join(%(tmp)r, "f.gz")
# (a) host allocation failure inside an entry point: with the address space capped, the compressed-block buffer of a
#     256 MiB write cannot be allocated -> std::bad_alloc inside pg_bgzf_write
w = ctypes.c_void_p()
assert lib.pg_bgzf_open(path.encode(), 1, 1, ctypes.byref(w)) == 0
data = bytes(256 << 20)
soft, hard = resource.getrlimit(resource.RLIMIT_AS)
import re
vm = int(re.search(r"VmSize:\s+(\d+) kB", open("/proc/self/status").read()).group(1)) << 10
resource.setrlimit(resource.RLIMIT_AS, (vm + (64 << 20), hard))
rc = lib.pg_bgzf_write(w, data, len(data))
print("write rc", rc, lib.pg_last_error().decode())
# (b) thread creation refused (each worker wants an 8 MiB stack): std::system_error inside pg_bgzf_open's pool
w2 = ctypes.c_void_p()
rc2 = lib.pg_bgzf_open((path + "2").encode(), 1, 512, ctypes.byref(w2))
print("open rc", rc2, lib.pg_last_error().decode())
resource.setrlimit(resource.RLIMIT_AS, (soft, hard))
print("alive")
"""


def test_host_allocation_failure_is_an_error_code_not_an_abort(tmp_path):
    """A std::bad_alloc / std::system_error raised inside the library comes back as a negative return code with a
    message in pg_last_error() — the interpreter survives (run in a child: the address-space limit is process-wide)."""
    import subprocess
    import sys
    from panagram_amd import build
    build.build(verbose=False)
    code = _FIREWALL_CHILD % {"root": ROOT, "tmp": str(tmp_path)}
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    out = p.stdout
    assert p.returncode == 0, (p.returncode, out, p.stderr[-2000:])
    assert "alive" in out
    m = re.search(r"write rc (-?\d+) (.*)", out)
    assert m and int(m.group(1)) == -4 and "memory" in m.group(2), out  # PG_E_CAPACITY
    m = re.search(r"open rc (-?\d+) (.*)", out)
    assert m and int(m.group(1)) < 0 and m.group(2), out
