#!/bin/bash
# FETCH_SIZE calibration on known byte counts (MI355X_MICROARCH.md §HBM: calibrate in your own
# access pattern): tools/gather_bench issues exactly 32/64/128 B per probe.
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/prof_gather
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE TCC_EA0_RDREQ_sum --output-format csv -d $R/gpurun_out/prof_gather/pmc -o pmc -- $R/tools/gather_bench > $R/gpurun_out/prof_gather/out.txt 2>&1
python - <<PY
import pandas as pd
df = pd.read_csv("$R/gpurun_out/prof_gather/pmc/pmc_counter_collection.csv")
df["k"] = df.Kernel_Name.str.slice(0, 48)
g = df.groupby(["k", "Counter_Name"]).Counter_Value.mean().unstack()
g.to_csv("$R/gpurun_out/prof_gather/calib.csv")
print(g.to_string())
PY
