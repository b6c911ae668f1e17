#!/usr/bin/env python3
"""BASELINE configs[3] files -> files (64 x 200 Mb, k = 31: 102 GB of rows through k_row_deflate and to disk) with
different numbers of writer jobs / other knobs of the write path:

    python tools/e2e_config4_writers.py --writers 2,4,8 [--genomes 64] [--mb 200]

Prints bench.e2e_leg()'s record per setting (seconds, anchor_and_write_s, GB/s of payload and of index bytes)."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--writers", default="4,8")
ap.add_argument("--genomes", type=int, default=64)
ap.add_argument("--mb", type=float, default=200.0)
ap.add_argument("--k", type=int, default=31)
ap.add_argument("--env", default="", help="NAME=VALUE[,NAME=VALUE...] set for every run")
a = ap.parse_args()
for kv in filter(None, a.env.split(",")):
    n, v = kv.split("=", 1)
    os.environ[n] = v
dev = torch.device("cuda:0")
args = argparse.Namespace(d=0.005, seed=1238)
lens = [int(a.mb * 1e6) // 10] * 10
for w in a.writers.split(","):
    os.environ["PG_WRITERS"] = w
    r = bench.e2e_leg(dev, args, a.genomes, lens, a.k)
    r["payload_gb_per_s"] = r["bitmap_payload_bytes"] / r["anchor_and_write_s"] / 1e9
    r["index_write_gb_per_s"] = r["index_bytes_out"] / r["anchor_and_write_s"] / 1e9
    keep = ("seconds", "read_parse_sketch_s", "table_insert_s", "anchor_and_write_s", "payload_gb_per_s", "index_write_gb_per_s", "row_batches", "anchor_batches_s", "writers_wait_s")
    print(f"PG_WRITERS={w}: " + json.dumps({k: (round(r[k], 3) if isinstance(r[k], float) else r[k]) for k in keep}), flush=True)
    torch.cuda.empty_cache()
