#!/bin/bash
# table build times only (first genome / the others), from the kernel trace:  bash tools/ab_build.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=/tmp/abb_$1; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o t -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-compare --no-other-shapes --no-sharded-leg --no-rehash > /dev/null 2> $O/err
python - <<PY
import csv,glob
f=glob.glob("$O/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "k_insert" in r["Name"]: print("[$1]", r["Name"][:40], "calls", r["Calls"], "total ms", round(float(r["TotalDurationNs"])/1e6,2), "min", round(float(r["MinNs"])/1e6,2), "max", round(float(r["MaxNs"])/1e6,2))
PY
