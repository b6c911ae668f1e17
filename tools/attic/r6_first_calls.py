"""what the first calls of a fresh process cost (no torch): python tools/attic/r6_first_calls.py"""
import os, sys, time
t00 = time.perf_counter()
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import numpy as np
t0 = time.perf_counter()
from panagram_amd import engine
t1 = time.perf_counter()
ctx = engine.Context(0)
t2 = time.perf_counter()
sk = engine.KmerSketch(ctx, 21)
t3 = time.perf_counter()
hb = ctx.host_buffer(100 << 20)
t4 = time.perf_counter()
txt = b">c\n" + b"ACGT" * 300000 + b"\n"
ss = engine.SeqSet.from_fasta(ctx, txt)
t5 = time.perf_counter()
sk.add(ss); sk.registers()
t6 = time.perf_counter()
tbl = engine.PanTable(ctx, 21, 8, expected_keys=1 << 20)
tbl.insert_seqset(0, ss); ctx.synchronize()
t7 = time.perf_counter()
res = engine.AnchorResult(tbl, ss, colsums=True); res.run(); ctx.synchronize()
t8 = time.perf_counter()
print(f"import numpy {t0 - t00:.3f}  import engine {t1 - t0:.3f}  Context {t2 - t1:.3f}  KmerSketch {t3 - t2:.3f}  host_buffer(100 MB) {t4 - t3:.3f}  "
      f"first from_fasta {t5 - t4:.3f}  first sketch {t6 - t5:.3f}  first table + insert {t7 - t6:.3f}  first anchor {t8 - t7:.3f}")
