#!/usr/bin/env python3
"""On the GPU box: condense gpurun_out/prof_<tag> (profiles/run_prof.sh) to one small text file and drop the raw CSVs
(they run to hundreds of MB for many-genome shapes):  python tools/pmc_summary.py <tag> <positions per launch>"""
import os, shutil, sys
import pandas as pd
tag, npos = sys.argv[1], float(sys.argv[2])
src = f"gpurun_out/prof_{tag}"
out = {}
meta = None
for d in ("pmc_fetch", "pmc_write", "pmc_ea", "pmc_misc", "pmc_sq"):
    f = f"{src}/{d}/pmc_counter_collection.csv"
    if not os.path.exists(f):
        continue
    df = pd.read_csv(f)
    a = df[df.Kernel_Name.str.contains("k_probe")]
    if len(a):
        out.update(a.groupby("Counter_Name").Counter_Value.mean().to_dict())
        meta = a.iloc[0]
lines = [f"# {tag}: k_probe counters per launch and per position ({npos:.4g} positions per launch)"]
for k_, v in sorted(out.items()):
    lines.append(f"{k_:28s} {v:14.6g}  {v / npos:10.4f} per position")
if meta is not None:
    lines.append(f"vgpr={meta.VGPR_Count} sgpr={meta.SGPR_Count} lds={meta.LDS_Block_Size} grid={meta.Grid_Size}")
ks = pd.read_csv(f"{src}/trace/trace_kernel_stats.csv")
ks = ks[ks.Name.str.contains("k_probe|k_epilogue|k_insert")]
ks["Name"] = ks.Name.str.slice(0, 70)
lines.append(ks[["Name", "Calls", "AverageNs", "MinNs", "MaxNs"]].to_string(index=False))
open(f"gpurun_out/{tag}_pmc_summary.txt", "w").write("\n".join(lines) + "\n")
shutil.rmtree(src)
print("\n".join(lines))
