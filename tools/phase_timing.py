"""Where a wave of k_probe spends its time: a library built with -DPG_PHASE_TIMING (bash tools/build_variant.sh ph
-DPG_PHASE_TIMING; cp build_variants/lib_ph.so panagram_amd/libpanagram_hip.so) stamps s_memtime at the phase boundaries
of every batch; this runs bench.py's default workload once and prints the phases' shares.
    python tools/phase_timing.py [bench args]"""
import ctypes
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
NAMES = ["prologue", "front end", "wait for lines", "staging", "issue next fetch", "slot scan", "row store",
         "loop bookkeeping", "drain + tail", "waves", "overflow entries"]


def main():
    import bench  # noqa: F401  (same process: the counters live in the library's device memory)
    from panagram_amd import _lib
    lib = _lib.load()
    fn = lib.pg_debug_phase_cycles
    fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    out = (ctypes.c_ulonglong * 16)()
    sys.argv = ["bench.py", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-compare", "--no-other-shapes", "--no-sharded-leg",
                "--no-e2e", "--no-robustness"] + sys.argv[1:]
    import io
    import contextlib
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.main()
    line = [l for l in buf.getvalue().splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert fn(out, 0) == 0
    v = list(out)
    waves = v[9]
    tot = sum(v[:9]) + v[10]
    print(f"value {d['value'] / 1e9:.1f} G k-mers/s, k_probe {d['roofline']['avg_launch_ms']:.3f} ms; waves timed {waves}, {tot / waves:.0f} cycles per wave (tile)")
    for n, c in list(zip(NAMES[:9], v[:9])) + [(NAMES[10], v[10])]:
        print(f"  {n:22s} {c / waves:9.0f} cycles per tile  {100.0 * c / tot:5.1f} %")
    if any(v[11:16]):  # (round 6: the fused statistics at the tile's end, inside "drain + tail")
        for n, c in zip(["fused: wait for the row stores", "fused: zero LDS, 1-in-100 request", "fused: rows, first pass (+ flush)",
                         "fused: rows, second pass, 1-in-100 stores", "fused: histogram out"], v[11:16]):
            print(f"    {n:48s} {c / waves:9.0f} cycles per tile  {100.0 * c / tot:5.1f} %")


if __name__ == "__main__":
    main()
