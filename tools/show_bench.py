#!/usr/bin/env python3
"""Condensed view of a bench.py line:  python tools/show_bench.py <file>"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("value G/s", round(d["value"] / 1e9, 2), "ms/step", round(d["ms_per_step"], 3), "frac", r and r["frac"] is not None and round(r["frac"], 3), r and r["bound"],
      "valu4", r and r.get("valu_issue_frac_4cycle"), "hbm", r and r.get("hbm_counter_frac"), "probe ms", r and round(r["avg_launch_ms"], 3),
      "stats ms", r and round(r["epilogue_kernel_ms"], 3))
c = d.get("cpu_baseline")
if c:
    print("cpu", round(c["value"] / 1e6, 2), "M/s", c["cores"], "cores", c["rows_equal_gpu"])
cfg = d["config"]
print("build s", cfg.get("table_build_s"), "per-genome G/s", cfg.get("per_genome_launches_value", 0) / 1e9)
if "genome_sharded_leg" in cfg:
    g = cfg["genome_sharded_leg"]
    print("sharded leg:", {k: g[k] for k in g if k in ("value", "ms_per_step", "error", "genome_blocks", "collective_bytes_received_per_rank_per_step")})
for o in cfg.get("other_shapes", []):
    print("other shape:", {k: o[k] for k in o if k in ("value", "ms_per_step", "error", "rows_equal_gpu", "table_build_s", "k_probe_ms", "k_epilogue_ms")})
if "e2e" in d:
    print("e2e:", {k: d["e2e"][k] for k in d["e2e"] if k in ("seconds", "value", "table_build_s", "anchor_and_write_s", "error")})
if "robustness" in d:
    for k, v in d["robustness"].items():
        print("robustness", k, {x: v[x] for x in v if x in ("value", "per_genome_launches_value", "error")} if isinstance(v, dict) else v)
