#!/usr/bin/env python3
"""Where a workgroup of k_row_deflate spends its time: a library built with -DPG_DF_PHASE (bash tools/build_deflate_variant.sh dfph
-DPG_DF_PHASE; cp build_variants/lib_dfph.so panagram_amd/libpanagram_hip.so) stamps the cycle counter at the kernel's phase
boundaries.   python tools/deflate_phases.py [N ...]   (N genomes = rows of ceil(N / 8) bytes; DR_MB: genome length, default 40)"""
import ctypes as C
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from panagram_amd import _lib, engine  # noqa: E402

NAMES = ["stage", "pass A (CRC, first / last unequal)", "run boundaries (scans)", "CRC combine", "bit counts", "offsets (scan)", "emission", "trailer"]
lib = _lib.load()
lib.pg_debug_df_phase_cycles.restype = C.c_int
lib.pg_debug_df_phase_cycles.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
MB = float(os.environ.get("DR_MB", "40"))
dev = torch.device("cuda", 0)
ctx = engine.Context(0)
for N in [int(x) for x in sys.argv[1:]] or [8, 27, 64]:
    L = int(MB * 1e6)
    lens = [L // 5] * 5
    genomes = bench.synth_genomes_device(N, lens, float(os.environ.get("DR_D", "0.01")), 1234, dev)
    tbl = engine.PanTable(ctx, 21, N, expected_keys=int(L * (1 + (N - 1) * 0.19) * 1.05))
    first = None
    for g in range(N):
        ss = engine.SeqSet(ctx, lens)
        for c, t in enumerate(genomes[g]):
            ss.load_dev(c, t.data_ptr(), t.numel())
        tbl.insert_seqset(g, ss)
        if g == 0:
            first = ss
        else:
            ss.close()
    res = engine.AnchorResult(tbl, first, colsums=True)
    res.run()
    ctx.synchronize()
    nbytes = (N + 7) // 8
    rows = first.total_kmers(21) * nbytes
    out = (C.c_ulonglong * 16)()
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "x.gz")
        res.write_bgzf(1, p, p + "i", level=-2, threads=16)
        lib.pg_debug_df_phase_cycles(out, 1)
        t0 = time.perf_counter()
        res.write_bgzf(1, p, p + "i", level=-2, threads=16)
        dt = time.perf_counter() - t0
        lib.pg_debug_df_phase_cycles(out, 1)
        size = os.path.getsize(p)
    wg = max(1, out[15])
    tot = sum(out[i] for i in range(len(NAMES)))
    print(f"N={N} rows of {nbytes} B: {rows / 1e9:.2f} GB in {dt * 1e3:.1f} ms = {rows / dt / 1e9:.1f} GB/s, ratio {rows / size:.1f}; {wg} workgroups, {tot / wg:.0f} cycles each")
    for i, nm in enumerate(NAMES):
        print(f"    {nm:38s} {out[i] / wg:10.0f} cycles  {100.0 * out[i] / max(1, tot):5.1f} %")
    sys.stdout.flush()
    res.close(); first.close(); tbl.close()
    del genomes
    torch.cuda.empty_cache()
