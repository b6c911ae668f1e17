import os, sys, time, tempfile, subprocess, json
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
d = sys.argv[1]
from panagram_amd import engine, index as pidx
calls = []
def wrap(cls, name):
    f = getattr(cls, name)
    def g(self, *a, **k):
        t0 = time.perf_counter(); r = f(self, *a, **k)
        try: self.ctx.synchronize()
        except Exception: pass
        calls.append((name, time.perf_counter() - t0)); return r
    setattr(cls, name, g)
for n in ("insert_seqset", "update_seqset", "stats"):
    wrap(engine.PanTable, n)
oi = engine.PanTable.__init__
def init(self, *a, **k):
    t0 = time.perf_counter(); oi(self, *a, **k); self.ctx.synchronize(); calls.append(("init kpl=%s" % k.get("keys_per_line"), time.perf_counter() - t0))
engine.PanTable.__init__ = init
import contextlib
with contextlib.redirect_stdout(sys.stderr):
    idx = pidx.Index(os.path.join(d, "samples.tsv"), prefix=os.path.join(d, "idx"), k=31)
    if os.environ.get("PROBE_RUN"):
        t0 = time.perf_counter(); idx.run(); t1 = time.perf_counter()
    else:
        idx.load_inputs()
        t0 = time.perf_counter(); idx.build_table(); t1 = time.perf_counter()
from collections import defaultdict
agg = defaultdict(lambda: [0, 0.0])
for n, t in calls: agg[n][0] += 1; agg[n][1] += t
print("build_table %.3f s:" % (t1 - t0), {n: (c, round(t, 3)) for n, (c, t) in agg.items()}, "timings", {k: round(v, 3) for k, v in idx.timings.items() if isinstance(v, float)})
