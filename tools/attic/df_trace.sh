#!/bin/bash
# kernel trace of the GPU BGZF writer:  bash tools/df_trace.sh
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/df
BR_L=200000000 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/df -o t -- python $R/tools/bgzf_rate.py > /dev/null 2>&1
python - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/df/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f))):
    if "deflate" in r['Name'] or "bgzf" in r['Name']:
        print(r['Name'][:40], r['Calls'], r['TotalDurationNs'], r['AverageNs'])
PY
