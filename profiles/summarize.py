#!/usr/bin/env python3
"""Condense a gpurun_out/prof_<tag>/ directory (written by profiles/run_prof.sh on the GPU box)
into profiles/<tag>_summary.md + the raw kernel-stats CSV, so the numbers bench.py reports
can be checked against rocprofv3.   python profiles/summarize.py <tag>"""
import json
import os
import sys

import pandas as pd

tag = sys.argv[1]
src = os.path.join("gpurun_out", f"prof_{tag}")
dst = os.path.dirname(os.path.abspath(__file__))
out = [f"# rocprofv3 summary — {tag}\n"]
bench = open(os.path.join(src, "trace_bench.json")).read().strip()
if bench:
    b = json.loads(bench.splitlines()[-1])
    out.append(f"bench.py line under the profiler: value={b['value']:.4g} {b['unit']}, "
               f"avg_launch_ms={b['roofline']['avg_launch_ms']:.4f}, frac={b['roofline']['frac']}\n")
    out.append(f"workload: {b['config']['workload']}\n")
ks = pd.read_csv(os.path.join(src, "trace", "trace_kernel_stats.csv"))
ks.to_csv(os.path.join(dst, f"{tag}_kernel_stats.csv"), index=False)
ks["Name"] = ks["Name"].str.slice(0, 60)
out.append("\n## `rocprofv3 --kernel-trace --stats` (top kernels)\n")
out.append("```\n" + ks.head(8).to_string(index=False) + "\n```\n")
out.append("\n## PMC passes (mean per dispatch of the dominant kernel, separate runs per counter set)\n")
for d in ("pmc_fetch", "pmc_write", "pmc_ea", "pmc_misc", "pmc_sq"):
    f = os.path.join(src, d, "pmc_counter_collection.csv")
    if not os.path.exists(f):
        continue
    df = pd.read_csv(f)
    name = df.Kernel_Name[df.Kernel_Name.str.contains("k_probe")]
    if name.empty:
        continue
    a = df[df.Kernel_Name == name.iloc[0]]
    out.append(f"### {d}\n```\n" + a.groupby("Counter_Name").Counter_Value.agg(["mean", "count"]).to_string() + "\n```\n")
    meta = a.iloc[0]
    out.append(f"grid={meta.Grid_Size} wg={meta.Workgroup_Size} vgpr={meta.VGPR_Count} sgpr={meta.SGPR_Count} lds={meta.LDS_Block_Size}\n")
open(os.path.join(dst, f"{tag}_summary.md"), "w").write("\n".join(out))
print("\n".join(out))
