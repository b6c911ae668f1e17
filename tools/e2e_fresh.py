#!/usr/bin/env python3
"""Files -> files as a user's process sees it: FASTA files written by ONE process (synthetic genomes of BASELINE configs[3] by
default), Index.run() timed in ANOTHER, fresh one per setting (nothing allocated or freed before it).

    python tools/e2e_fresh.py [--genomes 64 --mb 200 --k 31 --d 0.005] [--env "PG_TABLE_ROOMY=0;PG_WRITERS=8" ...]

Each --env is one run (NAME=VALUE pairs separated by ';', '' = defaults)."""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CHILD = r"""
import json, os, sys, time
sys.path.insert(0, %(root)r)
import contextlib
from panagram_amd import index as pidx
t0 = time.perf_counter()
with contextlib.redirect_stdout(sys.stderr):
    idx = pidx.Index(os.path.join(%(d)r, "samples.tsv"), prefix=os.path.join(%(d)r, "idx"), k=%(k)d)
    idx.run()
t1 = time.perf_counter()
out = dict(idx.timings)
out["seconds"] = t1 - t0
print(json.dumps(out))
"""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genomes", type=int, default=64)
    ap.add_argument("--mb", type=float, default=200.0)
    ap.add_argument("--contigs", type=int, default=10)
    ap.add_argument("--k", type=int, default=31)
    ap.add_argument("--d", type=float, default=0.005)
    ap.add_argument("--env", action="append", default=None)
    ap.add_argument("--pause", type=float, default=0.0, help="seconds between a process's end and the next one's start")
    ap.add_argument("--write-only", default="", help=argparse.SUPPRESS)
    a = ap.parse_args()
    if a.write_only:  # (the writer process)
        import torch
        import bench
        dev = torch.device("cuda:0")
        lens = [int(a.mb * 1e6) // a.contigs] * a.contigs
        genomes = bench.synth_genomes_device(a.genomes, lens, a.d, 1238, dev)
        rows = ["name\tfasta"]
        for g in range(a.genomes):
            fa = os.path.join(a.write_only, f"g{g}.fa")
            bench.write_fasta_from_device(fa, [f"chr{c + 1}" for c in range(a.contigs)], genomes[g])
            rows.append(f"g{g}\t{fa}")
        with open(os.path.join(a.write_only, "samples.tsv"), "w") as f:
            f.write("\n".join(rows) + "\n")
        return
    d = tempfile.mkdtemp(prefix="pg_e2e_fresh_")
    try:
        subprocess.run([sys.executable, os.path.abspath(__file__), "--write-only", d, "--genomes", str(a.genomes), "--mb", str(a.mb),
                        "--contigs", str(a.contigs), "--k", str(a.k), "--d", str(a.d)], check=True)
        for envs in (a.env or [""]):
            env = dict(os.environ)
            for kv in filter(None, envs.split(";")):
                n, v = kv.split("=", 1)
                env[n] = v
            shutil.rmtree(os.path.join(d, "idx"), ignore_errors=True)
            time.sleep(a.pause)
            t0 = time.perf_counter()
            p = subprocess.run([sys.executable, "-c", CHILD % dict(root=ROOT, d=d, k=a.k)], env=env, capture_output=True, text=True)
            wall = time.perf_counter() - t0
            lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
            if p.returncode != 0 or not lines:
                print(f"[{envs}] failed: {p.stderr[-600:]}")
                continue
            r = json.loads(lines[-1])
            for ln in p.stderr.splitlines():
                if "table built" in ln:
                    print("    " + ln[-260:])
            print(f"[{envs or 'defaults'}] Index.run() {r['seconds']:.2f} s (process wall {wall:.1f} s): " +
                  json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items() if k != "seconds"}), flush=True)
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
